// GPU box: how fast can ONE CU pull an L2-resident panel?  ("per-CU ingest")
//   hipcc -O3 --offload-arch=gfx950 tools/ingest_probe.hip -o /tmp/ingest_probe && /tmp/ingest_probe
// Every workgroup reads the same `panel` bytes (an activation panel all workgroups of a skinny GEMM need) `reps` times,
// 16 B per lane per load, U loads in flight per thread.  pattern 0: a wave instruction covers 1 KB contiguous;
// pattern 1: row segments -- a wave instruction reads (1024/seg) rows x seg bytes (row pitch = rowbytes); seg = 64 is the
// MFMA-operand pattern (16 rows x 64 B), seg = 128 a BK=64 bf16 tile row.  Reports us per launch and GB/s per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e));                             \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <int U, int PATTERN, int SEG>
__global__ __launch_bounds__(1024) void probe(const unsigned char *__restrict__ buf, int panel, int reps, int rowbytes, int distinct,
                                              uint32_t *out) {
    constexpr int ROWB = 2048, seg = SEG < 0 ? -SEG : SEG;   // compile-time geometry: no integer divisions in the loop
    (void)rowbytes;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    const unsigned char *base = buf + (distinct ? (size_t)blockIdx.x * panel : 0);
    u32x4 acc = (u32x4)(0);
    for (int r = 0; r < reps; ++r) {
        if (PATTERN == 0) {
            for (int off = tid * 16; off < panel; off += nthr * 16 * U) {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int o = off + u * nthr * 16;
                    v[u] = o < panel ? *reinterpret_cast<const u32x4 *>(base + o) : (u32x4)(0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= v[u];
            }
        } else {
            // panel = rows x rowbytes; a wave instruction reads 16 rows x 64 B at column block cb; waves split the column blocks
            const int rpi = 1024 / seg, lps = seg / 16;        // rows per instruction, lanes per segment
            const int rows = panel / ROWB, ncb = ROWB / seg, ntile = (rows / rpi) * ncb;
            for (int t = wave; t < ntile; t += nwave * U) {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int tt = t + u * nwave;
                    const int rt = tt / ncb, cb = tt % ncb;
                    const int o = SEG < 0 ? (rt * rpi + (lane & (rpi - 1))) * ROWB + cb * seg + (lane / rpi) * 16   // MFMA operand order: lanes 0..15 = rows
                                          : (rt * rpi + lane / lps) * ROWB + cb * seg + (lane % lps) * 16;
                    v[u] = tt < ntile ? *reinterpret_cast<const u32x4 *>(base + o) : (u32x4)(0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= v[u];
            }
        }
        asm volatile("" ::: "memory");
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int U, int P, int SEG = 64>
static float run(const unsigned char *buf, int wgs, int threads, int panel, int reps, int rowbytes, int distinct, uint32_t *out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<U, P, SEG>), dim3(wgs), dim3(threads), 0, 0, buf, panel, reps, rowbytes, distinct, out);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<U, P, SEG>), dim3(wgs), dim3(threads), 0, 0, buf, panel, reps, rowbytes, distinct, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main() {
    unsigned char *buf;
    uint32_t *out;
    const size_t total = (size_t)1024 * 512 * 1024;   // distinct panels for up to 1024 workgroups
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    CK(hipMalloc(&out, 4));
    const int panel = 128 * 1024, rowbytes = 2048;
    printf("panel %d KB; us per launch at reps=1 / reps=33, steady-state GB/s per workgroup = 32 panels / (t33 - t1)\n", panel / 1024);
    for (int wgs : {32, 256, 512})
        for (int threads : {256, 512, 1024}) {
            const float a1 = run<8, 0>(buf, wgs, threads, panel, 1, rowbytes, 0, out, 200), a33 = run<8, 0>(buf, wgs, threads, panel, 33, rowbytes, 0, out, 50);
            printf("shared panel (L2) wgs %4d threads %4d | contiguous 1 KB: %5.2f / %6.2f us %6.1f GB/s |", wgs, threads, a1, a33,
                   32.0 * panel / ((a33 - a1) * 1e-6) / 1e9);
#define SEGRUN(SEG_)                                                                                         \
    {                                                                                                        \
        const float t1 = run<8, 1, SEG_>(buf, wgs, threads, panel, 1, rowbytes, 0, out, 200);                \
        const float t33 = run<8, 1, SEG_>(buf, wgs, threads, panel, 33, rowbytes, 0, out, 50);               \
        printf(" seg %4d: %5.2f / %6.2f us %6.1f GB/s |", SEG_, t1, t33, 32.0 * panel / ((t33 - t1) * 1e-6) / 1e9); \
    }
            SEGRUN(-64) SEGRUN(64) SEGRUN(128) SEGRUN(256) SEGRUN(512)
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
