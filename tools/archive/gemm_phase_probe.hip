// GPU box: where does a wave of the LDS-DMA GEMM spend its cycles per k-tile?  (round 2)
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_phase_probe.hip -o /tmp/gemm_phase_probe && /tmp/gemm_phase_probe
// A stand-alone copy of gemm_bf16_pipe_kernel's main loop (tile BM x BN, WM x WN waves, BK = 64, STAGES-deep ring filled by
// global_load_lds_dwordx4, counted vmcnt) with s_memtime stamps around its four phases, accumulated per wave and averaged on the
// host: (1) s_waitcnt vmcnt for the oldest tile, (2) the workgroup barrier, (3) issuing the next tile's DMA, (4) fragment reads + MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int BM, int BN, int WM, int WN, int STAGES, bool BUF>
__global__ __launch_bounds__(WM *WN * 64) void probe_kernel(const bf16_t *__restrict__ A, int lda, const bf16_t *__restrict__ W, int ldw,
                                                             float *__restrict__ C, int ldc, int M, int N, int K, unsigned long long *stamps) {
#if defined(__HIP_DEVICE_COMPILE__)      // the buffer-resource builtins do not exist in the host pass
    constexpr int NT = WM * WN * 64, NW = WM * WN, BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 16, NTL = TN / 16;
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT, LPT = LA + LB;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nk = K / BK;
    const bf16_t *pa[LA];
    const bf16_t *pb[LB];
#pragma unroll
    for (int it = 0; it < LA; ++it) {
        const int c = (it * NW + wave) * 64 + lane, row = c >> 3, slot = c & 7;
        pa[it] = A + (int64_t)(m0 + row) * lda + ((slot ^ (row & 7)) << 3);
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
        const int c = (it * NW + wave) * 64 + lane, row = c >> 3, slot = c & 7;
        pb[it] = W + (int64_t)(n0 + row) * ldw + ((slot ^ (row & 7)) << 3);
    }
    // BUF: the same pieces as buffer_load_dwordx4 ... offen lds (SGPR resource + 32-bit per-lane byte offset + SGPR k offset)
    // instead of global_load_lds_dwordx4 with a 64-bit address per lane
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, M * lda * 2, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, N * ldw * 2, 0x00020000);
    int oa[LA], ob[LB];
#pragma unroll
    for (int it = 0; it < LA; ++it) oa[it] = (int)((pa[it] - A) * 2);
#pragma unroll
    for (int it = 0; it < LB; ++it) ob[it] = (int)((pb[it] - W) * 2);
    auto issue = [&](int kt) {
        unsigned char *sa = smem + (kt % STAGES) * STAGE_BYTES;
        unsigned char *sb = sa + BM * 128;
        if (BUF) {
            const int so = kt * BK * 2;
#pragma unroll
            for (int it = 0; it < LA; ++it)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void *)(sa + (it * NW + wave) * 1024), 16, oa[it], so, 0, 0);
#pragma unroll
            for (int it = 0; it < LB; ++it)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(sb + (it * NW + wave) * 1024), 16, ob[it], so, 0, 0);
            return;
        }
#pragma unroll
        for (int it = 0; it < LA; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pa[it] + (int64_t)kt * BK),
                                             (__attribute__((address_space(3))) void *)(sa + (it * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int it = 0; it < LB; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pb[it] + (int64_t)kt * BK),
                                             (__attribute__((address_space(3))) void *)(sb + (it * NW + wave) * 1024), 16, 0, 0);
    };
    f32x4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t_wait = 0, t_bar = 0, t_issue = 0, t_math = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p)
        if (p < nk) issue(p);
    const int frow = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (kt + STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1);
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        const unsigned char *Ab = smem + (kt % STAGES) * STAGE_BYTES;
        const unsigned char *Bb = Ab + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MT], fb[NTL];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = wm * TM + i * 16 + frow, ch = kk * 4 + fk;
                fa[i] = *reinterpret_cast<const bf16x8 *>(Ab + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int row = wn * TN + j * 16 + frow, ch = kk * 4 + fk;
                fb[j] = *reinterpret_cast<const bf16x8 *>(Bb + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_nop 0" ::: "memory");
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        t_wait += t1 - t0; t_bar += t2 - t1; t_issue += t3 - t2; t_math += t4 - t3;
    }
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int col = n0 + wn * TN + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * TM + i * 16 + (lane >> 4) * 4 + r;
                if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[i][j][r];
            }
        }
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        unsigned long long *o = stamps + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
        o[0] = t_wait; o[1] = t_bar; o[2] = t_issue; o[3] = t_math; o[4] = t_loop - t_begin; o[5] = t_end - t_loop; o[6] = t_begin; o[7] = t_end;
    }
#endif
}

template <int BM, int BN, int WM, int WN, int STAGES, bool BUF = false>
static void run(const char *name, int M, int N, int K, const bf16_t *A, const bf16_t *W, size_t wstride, int nw, float *C, unsigned long long *stamps) {
    constexpr size_t lds = (size_t)STAGES * (BM + BN) * 128;
    CK(hipFuncSetAttribute((const void *)probe_kernel<BM, BN, WM, WN, STAGES, BUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(N / BN, M / BM);
    const int nwg = grid.x * grid.y, NW = WM * WN;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < nw; ++i)
        hipLaunchKernelGGL((probe_kernel<BM, BN, WM, WN, STAGES, BUF>), grid, dim3(NW * 64), lds, 0, A, K, W + (size_t)i * wstride, K, C, N, M, N, K, stamps);
    CK(hipEventRecord(e0, 0));
    const int iters = 2 * nw;
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((probe_kernel<BM, BN, WM, WN, STAGES, BUF>), grid, dim3(NW * 64), lds, 0, A, K, W + (size_t)(i % nw) * wstride, K, C, N, M, N, K, stamps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)nwg * NW * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double s[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < nwg * NW; ++w) {
        for (int j = 0; j < 6; ++j) s[j] += (double)h[(size_t)w * 8 + j];
        if (h[(size_t)w * 8 + 6] < tmin) tmin = h[(size_t)w * 8 + 6];
        if (h[(size_t)w * 8 + 7] > tmax) tmax = h[(size_t)w * 8 + 7];
    }
    const double nwv = (double)nwg * NW, nk = K / 64.0;
    printf("%-22s %dx%dx%d: %6.2f us/launch | per wave per k-tile (s_memtime ticks @100 MHz? see span): wait %.0f  barrier %.0f  issue %.0f  read+mfma %.0f | "
           "loop %.0f  epilogue %.0f ticks per wave | first begin -> last end %llu ticks\n",
           name, M, N, K, ms * 1e3 / iters, s[0] / nwv / nk, s[1] / nwv / nk, s[2] / nwv / nk, s[3] / nwv / nk, s[4] / nwv, s[5] / nwv, tmax - tmin);
}

int main() {
    const int M = 512, N = 4096, K = 1024, nw = 40;
    bf16_t *A, *W;
    float *C;
    unsigned long long *stamps;
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&W, (size_t)nw * N * K * 2));
    CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&stamps, (size_t)8192 * 16 * 8 * 8));
    CK(hipMemset(A, 0x11, (size_t)M * K * 2));
    CK(hipMemset(W, 0x11, (size_t)nw * N * K * 2));
    const size_t ws = (size_t)N * K;
    run<64, 64, 4, 2, 4>("64x64 8w 4st", M, N, K, A, W, ws, nw, C, stamps);
    run<64, 64, 2, 2, 4>("64x64 4w 4st", M, N, K, A, W, ws, nw, C, stamps);
    run<128, 64, 4, 2, 4>("128x64 8w 4st", M, N, K, A, W, ws, nw, C, stamps);
    run<128, 64, 4, 2, 6>("128x64 8w 6st", M, N, K, A, W, ws, nw, C, stamps);
    run<128, 128, 4, 2, 4>("128x128 8w 4st", M, N, K, A, W, ws, nw, C, stamps);
    run<64, 64, 4, 2, 4, true>("64x64 8w 4st buf", M, N, K, A, W, ws, nw, C, stamps);
    run<128, 64, 4, 2, 4, true>("128x64 8w 4st buf", M, N, K, A, W, ws, nw, C, stamps);
    run<128, 64, 4, 2, 6, true>("128x64 8w 6st buf", M, N, K, A, W, ws, nw, C, stamps);
    run<128, 128, 4, 2, 4, true>("128x128 8w 4st buf", M, N, K, A, W, ws, nw, C, stamps);
    // the same with ONE weight matrix (Infinity-Cache / L2 warm): is it the HBM stream?
    run<64, 64, 4, 2, 4>("64x64 8w 4st warm", M, N, K, A, W, 0, 1, C, stamps);
    run<128, 64, 4, 2, 4>("128x64 8w 4st warm", M, N, K, A, W, 0, 1, C, stamps);
    return 0;
}
