// GPU box: how long does ONE launch take to stream a fixed total of HBM-cold bytes when it is spread over G workgroups?
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
// Planning measurement for a per-(sequence, head) fused q|k|v + attention + proj kernel of the GraphDiT block: at batch 1 it has
// only 32 workgroups, each of which must pull 576 KB of weights that no other workgroup needs.  Every workgroup reads its own
// `bytes` (16 B per lane per load, U loads in flight per thread, consecutive lanes contiguous); successive launches rotate over a
// 2 GB buffer so nothing is served from the 256 MiB Infinity Cache.  Reports us per launch (captured chain of dependent launches).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(1024) void stream_kernel(const unsigned char *__restrict__ buf, size_t bytes, uint32_t *out) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const unsigned char *base = buf + (size_t)blockIdx.x * bytes;
    u32x4 acc = (u32x4)(0);
    for (size_t off = (size_t)tid * 16; off < bytes; off += (size_t)nthr * 16 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = off + (size_t)u * nthr * 16;
            const u32x4 *p = reinterpret_cast<const u32x4 *>(base + (o < bytes ? o : 0));
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

template <int U, bool NT>
static float run(const unsigned char *buf, size_t total_buf, int wgs, int threads, size_t bytes, uint32_t *out) {
    const int chain = 64;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g;
    hipGraphExec_t ge;
    const size_t per_launch = (size_t)wgs * bytes;
    const int slots = (int)(total_buf / per_launch);
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < chain; ++i)
        hipLaunchKernelGGL((stream_kernel<U, NT>), dim3(wgs), dim3(threads), 0, st, buf + (size_t)(i % slots) * per_launch, bytes, out);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(st));
    return ms * 1e3f / (4 * chain);
}

int main() {
    unsigned char *buf;
    uint32_t *out;
    const size_t total = (size_t)2048 * 1024 * 1024;
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    CK(hipMalloc(&out, 4096 * 4));
    const size_t work = (size_t)18 * 1024 * 1024;   // q|k|v + proj weights of one block read once per sequence at batch 1: 2 x 8 MB (+ panels)
    printf("one launch streams %zu MB of HBM-cold bytes split over G workgroups (us per launch; GB/s total; GB/s per workgroup)\n", work >> 20);
    for (int wgs : {16, 32, 64, 128, 256, 512, 1024}) {
        const size_t bytes = work / wgs;
        for (int threads : {256, 512, 1024}) {
            const float a = run<4, false>(buf, total, wgs, threads, bytes, out), b = run<8, false>(buf, total, wgs, threads, bytes, out),
                        c = run<16, false>(buf, total, wgs, threads, bytes, out), d = run<8, true>(buf, total, wgs, threads, bytes, out);
            printf("G %4d x %4zu KB  threads %4d | U=4 %6.2f  U=8 %6.2f  U=16 %6.2f  U=8 nt %6.2f us | best %6.0f GB/s total, %5.1f per workgroup\n", wgs,
                   bytes >> 10, threads, a, b, c, d, work / (fminf(fminf(a, b), fminf(c, d)) * 1e-6) / 1e9,
                   bytes / (fminf(fminf(a, b), fminf(c, d)) * 1e-6) / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
