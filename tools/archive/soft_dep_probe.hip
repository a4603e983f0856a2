// GPU box: can a chain of dependent launches overlap each launch's fixed cost (dispatch ramp, weight prefetch) with its predecessor's
// tail?  Round-4 planning measurement for the GraphDiT launch chain (HISTORY.md R4.5).
//   mode 0  ordinary launches in stream order (the barrier bit of every AQL packet makes launch k+1 wait for launch k to COMPLETE)
//   mode 1  hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch): no barrier bit; the dependency is carried by memory --
//             producer: sc1 (agent-scope, write-through) stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> one relaxed agent atomic add
//             consumer: independent prologue (its "weights") -> one lane polls the counter -> workgroup barrier -> sc1 loads
//           Deadlock-free only if a queue's packets are dispatched in order (all workgroups of launch k are placed before any of k+1);
//           every wait is bounded, a timeout sets an error bit and the chain runs on.
//   mode 2  as 1, but the consumer's payload loads are plain (L2-cached) after ONE buffer_inv sc1 per workgroup
//   two streams: mode 1's kernels, ordinary launches, phases alternating between two streams (launch k+1 waits for k-1, not for k)
// Phase k, workgroup i: reads the slice workgroup perm(i) of phase k-1 wrote, adds 1, writes its own slice (ping-pong buffers); after K
// phases every word must be K.  `wkb` KB of never-reused "weights" per workgroup are read before the wait.
//   hipcc -O3 --offload-arch=gfx950 tools/soft_dep_probe.hip -o /tmp/soft_dep_probe && timeout 120 /tmp/soft_dep_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e = (x);                                       \
        if (e != hipSuccess) {                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e));         \
            exit(1);                                              \
        }                                                         \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;

struct Ctl {
    unsigned int error;
    unsigned int pad[31];
    unsigned int done[4096][32];      // one 128-byte line per phase
    unsigned int ticket[4096][8];     // poll 2: arrivals per (phase, XCC)
    unsigned int seen[4096][8][32];   // poll 2: per (phase, XCC) "the previous phase is complete", own line
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}

__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }

// poll: 0 = every workgroup polls the phase counter (s_sleep 2); 1 = the same with s_sleep 16; 2 = per XCD the first workgroup to arrive polls the
// phase counter and publishes it in the XCD's own word, the others poll that word
template <int MODE>
__global__ __launch_bounds__(THREADS) void phase(Ctl *ctl, uint32_t *buf, const u32x4 *w, size_t wvec, int wkb, int k, int G, int slice_bytes,
                                                 u32x4 *sink, int poll) {
    const int i = blockIdx.x, t = threadIdx.x;
    // ---- independent prologue: wkb KB of weights nobody else reads
    u32x4 acc = (u32x4)(0);
    {
        const int nv = wkb * 1024 / 16 / THREADS;      // 16-byte vectors per thread
        size_t pos = ((size_t)k * G + i) * (size_t)(wkb * 64) % wvec;
        for (int v = 0; v < nv; ++v) {
            const u32x4 x = __builtin_nontemporal_load(w + (pos + (size_t)v * THREADS + t) % wvec);
            acc ^= x;
        }
    }
    const int sv = slice_bytes / 16;      // vectors per slice
    const uint32_t *in = buf + (size_t)((k + 1) & 1) * G * (slice_bytes / 4);
    uint32_t *out = buf + (size_t)(k & 1) * G * (slice_bytes / 4);
    const int src = (int)(((long long)i * 7 + 3) % G);
    if (MODE != 0 && k > 0 && poll != 3) {
        if (t == 0) {
            unsigned int spins = 0;
            const unsigned int xcc = xcc_id();
            const bool leader = poll != 2 || atomicAdd(&ctl->ticket[k][xcc], 1u) == 0;
            if (leader) {
                while (__hip_atomic_load(&ctl->done[k - 1][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
                    if (poll == 1) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(2);
                    if (++spins > 400000u || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        atomicOr(&ctl->error, 1u);
                        break;
                    }
                }
                if (poll == 2) __hip_atomic_store(&ctl->seen[k][xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&ctl->seen[k][xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > 400000u || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        atomicOr(&ctl->error, 1u);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
    }
    for (int v = t; v < sv; v += THREADS) {
        u32x4 x;
        if (MODE == 1) x = __builtin_amdgcn_raw_buffer_load_b128(rsrc(in), (src * sv + v) * 16, 0, 16);      // sc1
        else x = *reinterpret_cast<const u32x4 *>(in + ((size_t)src * sv + v) * 4);
        x += (u32x4)(1);
        if (MODE != 0) __builtin_amdgcn_raw_buffer_store_b128(x, rsrc(out), (i * sv + v) * 16, 0, 16);       // sc1: write-through
        else *reinterpret_cast<u32x4 *>(out + ((size_t)i * sv + v) * 4) = x;
    }
    if (MODE != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_fetch_add(&ctl->done[k][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[i * THREADS + t] = acc;      // never (keeps the prologue alive)
}

static int g_poll = 0;
static hipStream_t g_st2;
static hipEvent_t g_fork, g_join;
static bool g_two_streams = false;      // mode 1 kernels, phases alternating between two streams (two hardware queues, no barrier between them)

template <int MODE>
static double run(Ctl *ctl, uint32_t *buf, const u32x4 *w, size_t wvec, int wkb, int K, int G, int slice_bytes, u32x4 *sink, hipStream_t st,
                  unsigned *err, unsigned *bad) {
    const size_t words = (size_t)2 * G * slice_bytes / 4;
    CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), st));
    CK(hipMemsetAsync(buf, 0, words * 4, st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    if (g_two_streams) {
        CK(hipEventRecord(g_fork, st));
        CK(hipStreamWaitEvent(g_st2, g_fork, 0));
    }
    for (int k = 0; k < K; ++k) {
        if (g_two_streams)
            hipLaunchKernelGGL(phase<MODE>, dim3(G), dim3(THREADS), 0, (k & 1) ? g_st2 : st, ctl, buf, w, wvec, wkb, k, G, slice_bytes, sink, g_poll);
        else if (MODE == 0)
            hipLaunchKernelGGL(phase<MODE>, dim3(G), dim3(THREADS), 0, st, ctl, buf, w, wvec, wkb, k, G, slice_bytes, sink, g_poll);
        else
            hipExtLaunchKernelGGL(phase<MODE>, dim3(G), dim3(THREADS), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ctl, buf, w, wvec, wkb, k, G,
                                  slice_bytes, sink, g_poll);
    }
    if (g_two_streams) {
        CK(hipEventRecord(g_join, g_st2));
        CK(hipStreamWaitEvent(st, g_join, 0));
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint32_t> h(words);
    CK(hipMemcpy(h.data(), buf, words * 4, hipMemcpyDeviceToHost));
    Ctl hc;
    CK(hipMemcpy(&hc, ctl, sizeof(unsigned int) * 32, hipMemcpyDeviceToHost));
    *err = hc.error;
    unsigned nb = 0;
    const uint32_t *last = h.data() + (size_t)((K - 1) & 1) * G * (slice_bytes / 4);
    for (size_t x = 0; x < (size_t)G * slice_bytes / 4; ++x) nb += last[x] != (uint32_t)K;
    *bad = nb;
    return ms * 1000.0 / K;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Ctl *ctl;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    const int K = 2000;
    const size_t wbytes = (size_t)1 << 30;      // 1 GB of "weights": a chunk is not reused for thousands of phases
    u32x4 *w, *sink;
    CK(hipMalloc(&w, wbytes));
    CK(hipMemset(w, 1, wbytes));
    CK(hipMalloc(&sink, (size_t)1024 * THREADS * 16));
    uint32_t *buf;
    CK(hipMalloc(&buf, (size_t)2 * 1024 * 65536));
    printf("chain of %d dependent phases, us per phase (error word, wrong words)\n", K);
    CK(hipStreamCreate(&g_st2));
    CK(hipEventCreateWithFlags(&g_fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&g_join, hipEventDisableTiming));
    printf("%5s %5s %5s | %-18s | %-18s | two streams (phase k+1 waits for k-1 in its stream, for k by flag): sc1 stores and ...\n", "G", "slice", "w KB",
           "stream order", "any-order, sc1 ld");
    printf("%5s %5s %5s | %-18s | %-18s | %-18s | %-18s | %-18s | %-18s\n", "", "KB", "", "", "", "sc1 loads", "plain loads", "plain, sleep 16", "plain, XCD leader");
    printf("(last column: two streams, no wait at all -- the launch path alone, results wrong by construction)\n");
    const int Gs[] = {256, 512};
    const int slices[] = {4096, 16384};
    const int wkbs[] = {0, 32, 128};
    for (int G : Gs)
        for (int sb : slices)
            for (int wkb : wkbs) {
                unsigned e[7], b[7];
                double t[7];
                run<0>(ctl, buf, w, wbytes / 16, wkb, 200, G, sb, sink, st, &e[0], &b[0]);      // warm
                t[0] = run<0>(ctl, buf, w, wbytes / 16, wkb, K, G, sb, sink, st, &e[0], &b[0]);
                t[1] = run<1>(ctl, buf, w, wbytes / 16, wkb, K, G, sb, sink, st, &e[1], &b[1]);
                g_two_streams = true;
                t[2] = run<1>(ctl, buf, w, wbytes / 16, wkb, K, G, sb, sink, st, &e[2], &b[2]);
                t[3] = run<3>(ctl, buf, w, wbytes / 16, wkb, K, G, sb, sink, st, &e[3], &b[3]);
                g_poll = 1;
                t[4] = run<3>(ctl, buf, w, wbytes / 16, wkb, K, G, sb, sink, st, &e[4], &b[4]);
                g_poll = 2;
                t[5] = run<3>(ctl, buf, w, wbytes / 16, wkb, K, G, sb, sink, st, &e[5], &b[5]);
                g_poll = 3;      // no wait at all: the launch path's own rate (results wrong by construction)
                t[6] = run<3>(ctl, buf, w, wbytes / 16, wkb, K, G, sb, sink, st, &e[6], &b[6]);
                g_poll = 0;
                g_two_streams = false;
                printf("%5d %5d %5d |", G, sb / 1024, wkb);
                for (int x = 0; x < 7; ++x) printf(" %6.2f (%u,%7u) |", t[x], e[x], b[x]);
                printf("\n");
                fflush(stdout);
            }
    return 0;
}
