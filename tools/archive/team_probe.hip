// GPU box: price of a dependent phase INSIDE one launch when the workgroups that exchange data share an XCD, with the cheap
// same-L2 protocol (round 4 planning measurement for the per-XCD GraphDiT trajectory kernel):
//   producer: plain 16-byte stores -> every wave s_waitcnt vmcnt(0) -> workgroup barrier -> one relaxed agent-scope atomic add
//   consumer: one lane polls the counter (sc1 loads) -> workgroup barrier -> reads the payload with sc1 (L1-bypassing) 16-byte loads
// No buffer_wbl2 / buffer_inv: the L2 is shared by the CUs of an XCD and the vector L1 is write-through; sc1 loads skip the L1.
// Teams are formed from HW_REG_XCC_ID at run time (census), never assumed from the workgroup id.
//   hipcc -O3 --offload-arch=gfx950 tools/team_probe.hip -o /tmp/team_probe && timeout 120 /tmp/team_probe
// Every word of every hand-off is checked; consumers re-read the same addresses every phase (L1-warm); `uneven` adds a
// pseudo-random delay per workgroup and phase; `stream` keeps 8 x 16 B non-temporal loads per thread in flight across the barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e = (x);                                       \
        if (e != hipSuccess) {                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e));         \
            exit(1);                                              \
        }                                                         \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Ctl {
    unsigned int census[8][32];     // per XCC: arrivals (own 128-byte line each)
    unsigned int bar[8][32];        // per XCC: monotonic barrier counter
    unsigned int error;             // bit 0 timeout, bit 2 stale word
    unsigned int stale_words;
    unsigned long long ticks[8];
};

__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }

__device__ __forceinline__ u32x4 load_sc1(const void *base, uint32_t byte_off, uint32_t bytes) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);      // aux 16 = sc1
}

constexpr int TEAM = 32, THREADS = 512;

// mode bits: 1 = write payload, 2 = read + verify everybody's payload, 4 = weight stream across the barrier, 8 = uneven load,
//            16 = plain (L1-cached) payload loads instead of sc1 (expected stale: shows the hazard is real)
__global__ __launch_bounds__(THREADS) void probe(Ctl *ctl, uint32_t *buf, const u32x4 *wstream, size_t wwords, int iters, int mode,
                                                  int payload_bytes, u32x4 *sink, int nteams) {
    extern __shared__ unsigned char pad_lds[];      // > 80 KB requested: one workgroup per CU
    __shared__ unsigned int s_rank;
    const unsigned int xcc = xcc_id();
    if (threadIdx.x == 0) s_rank = atomicAdd(&ctl->census[xcc][0], 1u);
    __syncthreads();
    const unsigned int rank = s_rank;
    if (rank >= TEAM) return;
    if (xcc >= (unsigned)nteams) return;      // only the first `nteams` XCDs take part
    // wait until the whole team has shown up (bounded)
    if (threadIdx.x == 0) {
        unsigned int spins = 0;
        while (__hip_atomic_load(&ctl->census[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < TEAM) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > 2000000u) { atomicOr(&ctl->error, 1u); break; }
        }
    }
    __syncthreads();
    const int pw = payload_bytes / 4;                      // words per workgroup payload
    uint32_t *team_buf = buf + (size_t)xcc * 2 * TEAM * pw;  // two generations
    unsigned long long t0 = 0;
    u32x4 acc = (u32x4)(0);
    uint32_t lcg = rank * 2654435761u + xcc * 40503u + 12345u;
    size_t wpos = (size_t)(xcc * TEAM + rank) * THREADS * 8 + threadIdx.x;      // wave instructions read 1 KB contiguous
    bool ok = true;
    if (mode & 1024) for (unsigned k = 0; k < xcc * 40; ++k) __builtin_amdgcn_s_sleep(64);      // XCDs staggered by ~1 us each
    for (int it = 0; it <= iters && ok; ++it) {
        if (it == 1 && rank == 0 && threadIdx.x == 0) t0 = wall_clock64();
        uint32_t *gen = team_buf + (size_t)(it & 1) * TEAM * pw;
        if (mode & 8) {
            lcg = lcg * 1664525u + 1013904223u;
            const int d = (lcg >> 24) & 31;
            for (int k = 0; k < d; ++k) __builtin_amdgcn_s_sleep(8);
        }
        if (mode & 1) {
            uint32_t *mine = gen + (size_t)rank * pw;
            for (int i = threadIdx.x * 4; i < pw; i += THREADS * 4) {
                const uint32_t v = (uint32_t)it * 0x10001u + (uint32_t)(rank * pw + i);
                *reinterpret_cast<u32x4 *>(mine + i) = (u32x4){v, v + 1, v + 2, v + 3};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have reached the L2
        __syncthreads();
        u32x4 wreg[8];
        const bool poller = (threadIdx.x >> 6) == 0 && !(mode & 32);      // the polling wave asks for its weights AFTER the poll: loads return in order per wave
        if ((mode & 4) && !poller) {                          // next phase's weights requested before the barrier completes
#pragma unroll
            for (int q = 0; q < 8; ++q) wreg[q] = __builtin_nontemporal_load(wstream + ((wpos + (size_t)q * THREADS) & (wwords - 1)));
            wpos += (size_t)256 * THREADS * 8;
        }
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&ctl->bar[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int target = (unsigned)(it + 1) * TEAM;
            unsigned int spins = 0;
            while (__hip_atomic_load(&ctl->bar[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { atomicOr(&ctl->error, 1u); ok = false; break; }
            }
        }
        if ((mode & 4) && poller) {
#pragma unroll
            for (int q = 0; q < 8; ++q) wreg[q] = __builtin_nontemporal_load(wstream + ((wpos + (size_t)q * THREADS) & (wwords - 1)));
            wpos += (size_t)256 * THREADS * 8;
        }
        __syncthreads();
        if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) ok = false;
        if ((mode & 2) && ok) {
            uint32_t bad = 0;
            const int total = TEAM * pw;
            const int rot = (mode & 2048) ? (int)rank * (total / TEAM) : 0;      // 2048: every CU starts its sweep at its own piece (spreads the L2 channels)
            for (int i0 = threadIdx.x * 4; i0 < total; i0 += THREADS * 4 * 16) {      // 16 loads in flight per thread (128 KB per workgroup at once)
                u32x4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * THREADS * 4;
                    if (i < total) {
                        const int ir = (i + rot) % total;
                        if (mode & 16) v[u] = *reinterpret_cast<const volatile u32x4 *>(gen + ir);
                        else v[u] = load_sc1(gen, (uint32_t)ir * 4u, (uint32_t)total * 4u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * THREADS * 4;
                    if (i < total) {
                        const uint32_t e = (uint32_t)it * 0x10001u + (uint32_t)((i + rot) % total);
                        bad += (v[u][0] != e) + (v[u][1] != e + 1) + (v[u][2] != e + 2) + (v[u][3] != e + 3);
                    }
                }
            }
            if (bad) { atomicOr(&ctl->error, 4u); atomicAdd(&ctl->stale_words, bad); }
        }
        if (mode & 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= wreg[q];
        }
        if (mode & 192) {      // 64: every XCD streams the SAME 16 KB-per-wave pieces (one HBM read, seven Infinity-Cache hits); 128: its own
            const size_t team_off = (mode & 128) ? (size_t)xcc * ((size_t)1 << 23) : 0;      // 128 MB apart
            // software-pipelined: two sets of 8 loads, one always in flight while the other is consumed (6 batches = 384 KB per CU)
            const size_t wmask = ((mode & 256) ? ((size_t)1 << 22) : wwords) - 1;
            const size_t base0 = team_off + ((size_t)(it * TEAM + rank) * THREADS * 48) + threadIdx.x;
            auto ldb = [&](u32x4 (&t)[8], int batch) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const u32x4 *pp = wstream + ((base0 + (size_t)(batch * 8 + q) * THREADS) & wmask);
                    t[q] = (mode & 512) ? *pp : __builtin_nontemporal_load(pp);
                }
            };
            u32x4 ta[8], tb[8];
            ldb(ta, 0);
            ldb(tb, 1);
#pragma unroll
            for (int bt = 0; bt < 6; bt += 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc ^= ta[q];
                if (bt + 2 < 6) ldb(ta, bt + 2);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc ^= tb[q];
                if (bt + 3 < 6) ldb(tb, bt + 3);
            }
        }
    }
    if (rank == 0 && threadIdx.x == 0) ctl->ticks[xcc] = wall_clock64() - t0;
    if (acc[0] == 0x12345678u) sink[threadIdx.x] = acc;
}

int main() {
    Ctl *ctl;
    uint32_t *buf;
    u32x4 *w, *sink;
    const size_t wbytes = (size_t)1 << 30;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&buf, (size_t)8 * 2 * TEAM * 65536));
    CK(hipMalloc(&w, wbytes));
    CK(hipMalloc(&sink, 65536));
    CK(hipMemset(w, 1, wbytes));
    CK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const int iters = 300;
    struct Case { int mode, payload; const char *what; int nteams = 8; };
    const Case cases[] = {
        {0, 4096, "barrier only"},
        {1, 4096, "write 4 KB + barrier"},
        {3, 4096, "write 4 KB + barrier + read 128 KB (sc1), verified"},
        {3, 8192, "write 8 KB + barrier + read 256 KB (sc1), verified"},
        {3, 16384, "write 16 KB + barrier + read 512 KB (sc1), verified"},
        {3 + 2048, 4096, "write 4 KB + barrier + read 128 KB (sc1) starting at the CU's own piece, verified"},
        {3 + 2048, 16384, "write 16 KB + barrier + read 512 KB (sc1) starting at the CU's own piece, verified"},
        {11, 4096, "uneven: write 4 KB + barrier + read 128 KB (sc1), verified"},
        {7, 4096, "weight stream across the barrier + write 4 KB + read 128 KB (sc1), verified"},
        {15, 4096, "uneven + weight stream + write 4 KB + read 128 KB (sc1), verified"},
        {19, 4096, "write 4 KB + barrier + read 128 KB with PLAIN loads (hazard demo)"},
        {4, 4096, "weight stream + barrier only"},
        {64, 4096, "barrier + 384 KB per CU streamed, same bytes in every XCD (3 x 16 loads per thread)"},
        {128, 4096, "barrier + 384 KB per CU streamed, different bytes per XCD"},
        {64 + 512, 4096, "same bytes, default cache policy"},
        {64 + 1024, 4096, "same bytes, nt, XCDs staggered"},
        {64 + 512 + 1024, 4096, "same bytes, default policy, XCDs staggered"},
        {64 + 256, 4096, "same bytes, nt, 64 MB window"},
        {64 + 256 + 512, 4096, "same bytes, default policy, 64 MB window"},
        {128 + 256 + 512, 4096, "different bytes per XCD (8 x 8 MB.. of a 64 MB window), default policy"},
        {64, 4096, "ONE XCD streams 384 KB per CU per phase (nt)", 1},
        {64, 4096, "TWO XCDs stream 384 KB per CU per phase, same bytes (nt)", 2},
        {128, 4096, "TWO XCDs stream 384 KB per CU per phase, different bytes (nt)", 2},
        {64, 4096, "FOUR XCDs stream 384 KB per CU per phase, same bytes (nt)", 4},
        {128, 4096, "FOUR XCDs stream 384 KB per CU per phase, different bytes (nt)", 4},
        {64 + 256, 4096, "ONE XCD, 64 MB window (nt)", 1},
        {64 + 256 + 512, 4096, "ONE XCD, 64 MB window (default policy)", 1},
        {67, 4096, "write 4 KB + barrier + read 128 KB (sc1) + 384 KB per CU streamed (same bytes)"},
        {36, 4096, "weight stream (polling wave prefetches BEFORE its poll) + barrier only"},
        {39, 4096, "weight stream (polling wave prefetches before its poll) + write 4 KB + read 128 KB (sc1), verified"},
    };
    for (const Case &c : cases) {
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        hipLaunchKernelGGL(probe, dim3(256), dim3(THREADS), 100 * 1024, 0, ctl, buf, w, wbytes / 16, iters, c.mode, c.payload, sink, c.nteams);
        CK(hipDeviceSynchronize());
        Ctl r;
        CK(hipMemcpy(&r, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        double mn = 1e9, mx = 0;
        for (int x = 0; x < c.nteams; ++x) {
            const double us = (double)r.ticks[x] / rate_khz * 1e3 / iters;
            mn = us < mn ? us : mn;
            mx = us > mx ? us : mx;
        }
        printf("%-86s %.2f .. %.2f us per phase  census", c.what, mn, mx);
        for (int x = 0; x < 8; ++x) printf(" %u", r.census[x][0]);
        printf("%s", r.error & 1 ? "  ** TIMEOUT **" : "");
        if (r.error & 4) printf("  stale words: %u", r.stale_words);
        printf("\n");
    }
    return 0;
}
