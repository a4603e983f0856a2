// GPU box: what does a dependent phase cost INSIDE one launch when producer and consumer share an XCD (one L2)?
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_probe.hip -o /tmp/xcd_probe && timeout 120 /tmp/xcd_probe
// Planning measurement for a per-XCD persistent GraphDiT trajectory kernel ("what would"): one workgroup
// per CU; the workgroups that find themselves on XCC `x` (HW_REG_XCC_ID) form a team of 32 and run `iters` phases of
//   write 4 KB -> team barrier (one monotonic counter, agent-scope relaxed atomics, s_sleep polling)
//   [-> agent-scope acquire fence] [-> read the 4 KB the left neighbour wrote, checked]
// against the same loop over all 256 workgroups (chip-wide barrier).  Every spin is bounded; a timeout sets an error flag.
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

struct Ctl {
    unsigned int team_size[9];   // arrivals per XCC (8 = whole chip)
    unsigned int barrier[9];     // monotonic barrier counters
    unsigned int error;
    unsigned int census[8];
    unsigned long long ticks;    // wall_clock64 ticks of rank 0 over the timed loop
};

__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]

// mode bit 0: acquire fence after the barrier; bit 1: read the neighbour's 4 KB; team = 0..7 one XCC, 8 = all workgroups
__global__ __launch_bounds__(256) void probe(Ctl *ctl, uint32_t *buf, int iters, int mode, int team, int expect) {
    extern __shared__ unsigned char pad_lds[];      // 100 KB requested: one workgroup per CU
    __shared__ unsigned int s_rank;
    const unsigned int xcc = xcc_id();
    if (threadIdx.x == 0 && team == 8) atomicAdd(&ctl->census[xcc], 1u);
    if (team != 8 && xcc != (unsigned)team) return;
    if (threadIdx.x == 0) s_rank = atomicAdd(&ctl->team_size[team], 1u);
    __syncthreads();
    const unsigned int rank = s_rank;
    if (rank >= (unsigned)expect) return;            // more members than planned: sit out (the barrier counts `expect`)
    uint32_t *mine0 = buf + (size_t)rank * 2048, *left0 = buf + (size_t)((rank + expect - 1) % expect) * 2048;     // two 4 KB buffers each
    unsigned long long t0 = 0;
    bool ok = true;
    for (int it = 0; it <= iters && ok; ++it) {
        if (it == 1 && rank == 0 && threadIdx.x == 0) t0 = wall_clock64();
        uint32_t *mine = mine0 + (it & 1) * 1024, *left = left0 + (it & 1) * 1024;
        // phase body: 4 KB per workgroup, value = iteration tag
        for (int i = threadIdx.x; i < 1024; i += 256) __hip_atomic_store(mine + i, (uint32_t)(it * 4096 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(&ctl->barrier[team], 1u);
            const unsigned int target = (unsigned)(it + 1) * (unsigned)expect;
            unsigned int spins = 0;
            while (__hip_atomic_load(&ctl->barrier[team], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 4000000u) { atomicExch(&ctl->error, 1u); ok = false; break; }
            }
            if (mode & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (__hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) ok = false;
        if ((mode & 2) && ok) {
            uint32_t bad = 0;
            for (int i = threadIdx.x; i < 1024; i += 256) bad |= (left[i] != (uint32_t)(it * 4096 + i));
            if (bad) atomicOr(&ctl->error, 4u);      // stale read (informational without the acquire)
        }
    }
    if (rank == 0 && threadIdx.x == 0) ctl->ticks = wall_clock64() - t0;
}

int main() {
    Ctl *ctl;
    uint32_t *buf;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&buf, 256 * 8192));
    CK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    Ctl h;
    // census: workgroups per XCC for a 256-workgroup launch
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    hipLaunchKernelGGL(probe, dim3(256), dim3(256), 100 * 1024, 0, ctl, buf, 0, 0, 8, 256);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("workgroups per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %u", h.census[x]);
    printf("  (wall clock %d kHz)\n", rate_khz);
    const int iters = 400;
    for (int team : {0, 8}) {
        const int expect = team == 8 ? 256 : (int)h.census[0];
        for (int mode : {0, 1, 2, 3}) {
            CK(hipMemset(ctl, 0, sizeof(Ctl)));
            hipLaunchKernelGGL(probe, dim3(256), dim3(256), 100 * 1024, 0, ctl, buf, iters, mode, team, expect);
            CK(hipDeviceSynchronize());
            Ctl r;
            CK(hipMemcpy(&r, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            printf("%s, %3d workgroups: %.2f us per phase (write 4 KB + barrier%s%s)%s%s\n", team == 8 ? "whole chip" : "one XCD   ", expect,
                   (double)r.ticks / rate_khz * 1e3 / iters, mode & 1 ? " + agent acquire" : "", mode & 2 ? " + read the neighbour's 4 KB" : "",
                   r.error & 1 ? "  ** TIMEOUT **" : "", r.error & 4 ? "  (stale reads seen)" : "");
        }
    }
    return 0;
}
