// GPU box: the floor of one dependent phase = one kernel of a captured hipGraph chain.
//   hipcc -O3 --offload-arch=gfx950 [-mllvm -amdgpu-kernarg-preload-count=16] tools/phase_floor_probe.hip -o /tmp/pf && /tmp/pf
// A chain of 256 launches is captured once and replayed; per-launch time = replay time / 256 (HIP events).  Variants:
//   empty      : the kernel returns at once                                  -> launch boundary alone
//   load       : 64 workgroups x 64 threads read 16 B each of what the PREVIOUS launch wrote (no store)
//   load+store : read, add, write for the next launch                          (a ln_mod_res-like phase without its math)
//   wide       : 32 / 192 workgroups x 256 threads each reading a 16..128 KB panel (a gemm_m64-like ingest): one shared panel or an own
//                slice per workgroup, fresh (rewritten by every launch for the next) or read-only
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(const u4 *in, u4 *out, int n) {}
__global__ void k_load(const u4 *in, u4 *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    u4 v = in[i];
    if (v[0] == 0x12345678u) out[i] = v;
}
__global__ void k_load_store(const u4 *in, u4 *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    u4 v = in[i];
    v[0] += 1;
    out[i] = v;
}
// every workgroup reads `kb` KB (own = its own slice, else one shared panel), 16 B per lane, up to 32 loads in flight per thread;
// fresh = the panel was written by the previous launch (every launch rewrites what the next one reads), else it is read-only
template <int KB, bool OWN, bool FRESH, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void k_wide(const u4 *in, u4 *out, int n) {
    constexpr int LOADS = KB * 1024 / 16 / THREADS;
    const u4 *src = (FRESH ? in : in + (size_t)4 * 1024 * 1024) + (OWN ? (size_t)blockIdx.x * (KB * 64) : 0);
    u4 acc = (u4)(0);
#pragma unroll
    for (int j = 0; j < LOADS; ++j) acc ^= src[threadIdx.x + j * THREADS];
    acc[0] += 1;
    // rewrite the region the next launch reads (OWN: own slice; shared: the first KB/4 workgroups cover the panel)
    u4 *dst = out + (OWN ? (size_t)blockIdx.x * (KB * 64) : 0);
#pragma unroll
    for (int j = 0; j < LOADS; ++j)
        if (OWN || (int)blockIdx.x == j % (int)gridDim.x) dst[threadIdx.x + j * THREADS] = acc;
}

template <typename K> static float chain(K kern, dim3 grid, dim3 block, u4 *a, u4 *b, int n) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 256; ++i) hipLaunchKernelGGL(kern, grid, block, 0, st, (const u4 *)((i & 1) ? b : a), (i & 1) ? a : b, n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(st));
    return ms * 1e3f / (20 * 256);
}

int main() {
    u4 *a, *b;
    const int n = 8192;
    const size_t bytes = (size_t)160 * 1024 * 1024;     // room for 192 own 128 KB slices and the read-only region at +64 MB
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    printf("empty       (64 x 64)  : %.2f us per launch\n", chain(k_empty, dim3(64), dim3(64), a, b, n));
    printf("load        (64 x 64)  : %.2f us per launch\n", chain(k_load, dim3(64), dim3(64), a, b, n));
    printf("load+store  (64 x 64)  : %.2f us per launch\n", chain(k_load_store, dim3(64), dim3(64), a, b, n));
    printf("load+store  (256 x 256): %.2f us per launch\n", chain(k_load_store, dim3(32), dim3(256), a, b, n));
#define WIDE(KB, OWN, FRESH, G) printf("%3d KB %s %s panel, %3d workgroups: %.2f us per launch\n", KB, OWN ? "own   " : "shared", \
                                       FRESH ? "fresh    " : "read-only", G, chain(k_wide<KB, OWN, FRESH>, dim3(G), dim3(256), a, b, n))
    WIDE(128, false, true, 192); WIDE(128, false, true, 32); WIDE(128, false, false, 192); WIDE(128, false, false, 32);
    WIDE(64, false, true, 192); WIDE(32, false, true, 192); WIDE(16, false, true, 192);
    WIDE(128, true, true, 192); WIDE(128, true, false, 192); WIDE(32, true, true, 192);
    printf("128 KB shared fresh panel, 192 workgroups of 512 threads : %.2f us per launch\n", chain(k_wide<128, false, true, 512>, dim3(192), dim3(512), a, b, n));
    printf("128 KB shared fresh panel, 192 workgroups of 1024 threads: %.2f us per launch\n", chain(k_wide<128, false, true, 1024>, dim3(192), dim3(1024), a, b, n));
    printf("128 KB shared fresh panel,  32 workgroups of 1024 threads: %.2f us per launch\n", chain(k_wide<128, false, true, 1024>, dim3(32), dim3(1024), a, b, n));
    return 0;
}
