// GPU box: where does qkv_attn_kernel (q|k|v projection + attention, one workgroup per (sequence, head)) spend its time?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -DLL_QA_PROBE \
//         tools/qkv_attn_probe.hip -o /tmp/qkv_attn_probe && /tmp/qkv_attn_probe [B=8] [weight sets=64]
// Launches the production kernel (llamole_amd/csrc/dit_kernels.h, compiled with cycle stamps of wave 0) over `copies` distinct
// weight sets back to back (more than the 256 MiB Infinity Cache holds), prints the event-timed average launch and the stamp
// deltas: 0 start | 1 panel + first weight blocks requested, panel in LDS | 2 barrier | 3 K loop done | 4 LayerNorm statistics exchanged,
// this wave's Q column group written | 5 all of Q and K written | 6 QK^T + softmax | 7 V^T written by its waves, PV + store.
#include "../llamole_amd/csrc/dit_kernels.h"
#include <vector>
namespace ll { void set_error(const char *, ...) {} }
using namespace ll;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, kpb = QkvAttnGeom<32, 1024>::KPB;
    const int N = 32, H = 1024, heads = 16, S = 2 * B, copies = argc > 2 ? atoi(argv[2]) : 64;      // copies = 1: L2 / Infinity-Cache-hot weights
    bf16_t *xa, *W, *o;
    float *ln;
    int *nn;
    CK(hipMalloc(&xa, (size_t)S * N * H * 2));
    CK(hipMalloc(&W, (size_t)copies * 3 * H * H * 2));
    CK(hipMalloc(&o, (size_t)S * N * H * 2));
    CK(hipMalloc(&ln, 256 * 4));
    CK(hipMalloc(&nn, B * 4));
    std::vector<bf16_t> h((size_t)3 * H * H);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (bf16_t)(0x3c00 + (rand() & 0xff));      // small positive bf16 values
    for (int c = 0; c < copies; ++c) CK(hipMemcpy(W + (size_t)c * 3 * H * H, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    std::vector<bf16_t> hx((size_t)S * N * H);
    for (auto &v : hx) v = (bf16_t)(0x3c00 + (rand() & 0xff));
    CK(hipMemcpy(xa, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hl(256, 1.f);
    CK(hipMemcpy(ln, hl.data(), 1024, hipMemcpyHostToDevice));
    std::vector<int> hn(B, N);
    CK(hipMemcpy(nn, hn.data(), B * 4, hipMemcpyHostToDevice));
    const size_t lds = QkvAttnGeom<32, 1024>::lds_bytes();
    CK(hipFuncSetAttribute((const void *)qkv_attn_kernel<32, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&](int c) {
        const bf16_t *w = W + (size_t)c * 3 * H * H;
        hipLaunchKernelGGL((qkv_attn_kernel<32, 1024>), dim3(S * heads), dim3(768), lds, 0, xa, w, o, ln, ln + 64, ln + 128, ln + 192, nn, B, N, H, heads);
    };
    for (int c = 0; c < copies; ++c) launch(c);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < copies; ++c) launch(c);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B=%d kpb=%d: %d workgroups, %.2f us per launch (%d weight sets, back to back)\n", B, kpb, S * heads, ms * 1000.f / (4 * copies), copies);
    std::vector<unsigned long long> st(4096 * 8);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_qa_stamps), st.size() * 8));
    const int nw = std::min(S * heads, 4096);
    double d[8] = {0};
    for (int w = 0; w < nw; ++w)
        for (int i = 1; i < 8; ++i) d[i] += (double)(st[w * 8 + i] - st[w * 8 + i - 1]);
    printf("cycles (wave 0, mean over %d workgroups): stage %.0f | barrier %.0f | k loop %.0f | ln + own image %.0f | wait for Q, K images %.0f | qk+softmax %.0f | wait for V^T + pv + store %.0f | total %.0f\n",
           nw, d[1] / nw, d[2] / nw, d[3] / nw, d[4] / nw, d[5] / nw, d[6] / nw, d[7] / nw,
           (d[1] + d[2] + d[3] + d[4] + d[5] + d[6] + d[7]) / nw);
    std::vector<unsigned long long> ws(1024 * 12 * 4);
    CK(hipMemcpyFromSymbol(ws.data(), HIP_SYMBOL(g_qa_wstamps), ws.size() * 8));
    const int nb = std::min(S * heads, 1024);
    printf("per wave, cycles after wave 0 entered the kernel (mean over %d workgroups): entered | panel staged | past the barrier | k loop done\n", nb);
    for (int w = 0; w < 12; ++w) {
        double a[4] = {0};
        for (int b = 0; b < nb; ++b)
            for (int i = 0; i < 4; ++i) a[i] += (double)(long long)(ws[(b * 12 + w) * 4 + i] - ws[(b * 12) * 4]);
        printf("  wave %2d: %6.0f %6.0f %6.0f %6.0f\n", w, a[0] / nb, a[1] / nb, a[2] / nb, a[3] / nb);
    }
    return 0;
}
