// Shared helpers for the gfx950 kernels of the Llamole graph hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/llamole_hip.h"
#include "../../include/llamole_hip_tuning.h"

// LL_TUNING=1 (libllamole_hip_tuning.so: tests, tools, bench.py) also compiles the entry points of include/llamole_hip_tuning.h --
// process-global A/B switches, micro-benchmarks, probes; the product library (libllamole_hip.so, LL_TUNING=0) exports none of them.
// Kernels and product entry points are the same translation units in both builds.
#ifndef LL_TUNING
#define LL_TUNING 0
#endif

namespace ll {

typedef uint16_t bf16_t;  // raw bfloat16 bits

void set_error(const char *fmt, ...);

#define LL_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            ll::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return LL_EHIP;                                                                     \
        }                                                                                       \
    } while (0)

#define LL_CHECK(cond, ...)            \
    do {                               \
        if (!(cond)) {                 \
            ll::set_error(__VA_ARGS__); \
            return LL_EINVAL;          \
        }                              \
    } while (0)

#define LL_TRY(expr)            \
    do {                        \
        int rc__ = (expr);      \
        if (rc__ != 0) return rc__; \
    } while (0)

// Debug allocation mode of the engines' device buffers (env LL_DEBUG_POISON=1; off = plain zero-filled allocations): the payload starts as
// 0xFF bytes (NaN / -1) and LL_GUARD_BYTES of 0xA5 follow it; `debug_guard_check` (buffer growth / release) reports any byte a kernel wrote
// past the end on stderr and aborts -- the GPU test suite run in this mode is the out-of-bounds hunt (no GPU AddressSanitizer on this pool).
constexpr size_t LL_GUARD_BYTES = 4096;
inline bool debug_poison() {
    static const bool on = getenv("LL_DEBUG_POISON") != nullptr;
    return on;
}
void debug_registry_add(const void *p, size_t n);      // graphdit.hip: the live guarded buffers (debug mode only)
void debug_registry_remove(const void *p);
inline int debug_alloc(void **p, size_t n) {
    const bool dbg = debug_poison();
    hipError_t e = hipMalloc(p, n + (dbg ? LL_GUARD_BYTES : 0));
    if (e == hipSuccess) e = hipMemset(*p, dbg ? 0xFF : 0, n);
    if (e == hipSuccess && dbg) e = hipMemset((char *)*p + n, 0xA5, LL_GUARD_BYTES);
    if (e != hipSuccess) {
        set_error("device allocation of %zu bytes failed: %s", n, hipGetErrorString(e));
        return LL_EHIP;
    }
    if (dbg) debug_registry_add(*p, n);
    return LL_OK;
}
// number of guard regions a kernel wrote into (0 = intact); the caller has synchronised the device
inline int debug_guard_damage(const void *p, size_t n, const char *when) {
    static unsigned char host[LL_GUARD_BYTES];
    if (hipMemcpy(host, (const char *)p + n, LL_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    for (size_t i = 0; i < LL_GUARD_BYTES; ++i)
        if (host[i] != 0xA5) {
            fprintf(stderr, "LL_GUARD: a kernel wrote past the end of a %zu-byte engine buffer (first damaged byte at +%zu; %s)\n", n, i, when);
            return 1;
        }
    return 0;
}
inline void debug_guard_check(const void *p, size_t n, const char *when) {      // buffer growth / release
    if (!p || !debug_poison()) return;
    debug_registry_remove(p);
    if (hipDeviceSynchronize() == hipSuccess && debug_guard_damage(p, n, when)) abort();
}

#define LL_LAUNCH_CHECK()                                                                   \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess) {                                                            \
            ll::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return LL_EHIP;                                                                 \
        }                                                                                   \
    } while (0)

__host__ __device__ inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device numerics
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the native conversion (v_cvt_pk_bf16_f32 on gfx950): branch-free, NaN-safe
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float softsign(float x) { return x / (1.0f + fabsf(x)); }

// Cross-lane reductions on the VALU data path (DPP), not the LDS crossbar (ds_bpermute, ~100 cycles each):
// quad_perm swaps for xor 1 / xor 2, then row_half_mirror / row_mirror (after the quad steps every quad, then
// every 8-lane half, already holds one value, so the mirrors act as xor 4 / xor 8).
#define LL_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float row16_sum(float v) {
    v += LL_DPP(v, 0xB1);
    v += LL_DPP(v, 0x4E);
    v += LL_DPP(v, 0x141);
    v += LL_DPP(v, 0x140);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, LL_DPP(v, 0xB1));
    v = fmaxf(v, LL_DPP(v, 0x4E));
    v = fmaxf(v, LL_DPP(v, 0x141));
    v = fmaxf(v, LL_DPP(v, 0x140));
    return v;
}
__device__ __forceinline__ float row8_sum(float v) {
    v += LL_DPP(v, 0xB1);
    v += LL_DPP(v, 0x4E);
    v += LL_DPP(v, 0x141);
    return v;
}
__device__ __forceinline__ float row4_sum(float v) {
    v += LL_DPP(v, 0xB1);
    v += LL_DPP(v, 0x4E);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    const int b = __builtin_bit_cast(int, v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
}

// Philox4x32-10 counter-based generator (Salmon et al. 2011); one call = 4 x 32 random bits.
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}
// Exp(1) variate from 32 random bits: -log(u), u uniform in (0,1].
__device__ __forceinline__ float exp1_from_bits(uint32_t r) { return -__logf(((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f)); }

// sum of squared deviations of the four columns [k, k + 4) of a row chunk, counting only columns below `width` (the true row width: engines
// pad rows to a multiple of 64 with zeros, and LayerNorm statistics are over the checkpoint's own width); same order as the plain form
__device__ __forceinline__ float sq_dev4(const float4 v, float mean, int k, int width) {
    if (k + 4 <= width) {
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        return d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    float r = 0.f;
    if (k < width) { const float d = v.x - mean; r += d * d; }
    if (k + 1 < width) { const float d = v.y - mean; r += d * d; }
    if (k + 2 < width) { const float d = v.z - mean; r += d * d; }
    return r;
}


// ---------------------------------------------------------------- zero-padded internal weight layouts (graphdit.hip, gin.hip)
// dst[(r / rg2) * rgp2 + (r % rg2 / rg) * rgp + r % rg][(c / cg) * cgp + c % cg] = src[r][c]: how a checkpoint tensor lands in an engine's arena
struct PadMap {
    int rg2 = 0, rgp2 = 0, rg = 0, rgp = 0, cg = 0, cgp = 0;      // 0 = one group (identity)
};
// A checkpoint tensor [R][C] into its zero-padded internal form (engine creation; the destination was zeroed):
//   dst[(r / rg2) * rgp2 + (r % rg2 / rg) * rgp + r % rg][(c / cg) * cgp + c % cg] = src[r][c],   destination row pitch ldd
// rows: sections of rg2 rows (q | k | v of the qkv weight) made of groups of rg rows (heads, the six chunks of an adaLN output);
// columns: groups of cg columns (heads on the K side of proj).  A group count of one (rg = R, cg = C) is the identity.
static __global__ void pad_copy_kernel(const float *__restrict__ src, float *__restrict__ dst, int R, int C, int ldd, int rg2, int rgp2, int rg,
                                int rgp, int cg, int cgp) {
    const int64_t n = (int64_t)R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i - (int64_t)r * C);
        const int rr = r % rg2;
        const int64_t dr = (int64_t)(r / rg2) * rgp2 + (rr / rg) * rgp + rr % rg;
        const int dc = (c / cg) * cgp + c % cg;
        dst[dr * ldd + dc] = src[i];
    }
}


// ---------------------------------------------------------------- host helpers
int linear_launch(int dtype, const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc,
                  int M, int N, int K, int epi, int out_f32, hipStream_t stream);

// split-K variant: writes `splits` f32 slabs C[z][M][ldc] (no bias / epilogue); consumer sums slabs in order.
int linear_splitk_launch(int dtype, const void *A, int lda, const void *W, int ldw, float *Cslabs, int ldc,
                         int64_t slab_stride, int M, int N, int K, int splits, hipStream_t stream);

// two row groups in one launch: rows [0, m_split) x W1 (+ bias1), rows [m_split, M) x W2 (+ bias2); m_split % 64 == 0 (gemm.hip)
int linear_grouped2_launch(int dtype, const void *A, int lda, const void *W1, const void *W2, int ldw, const float *bias1,
                           const float *bias2, void *C, int ldc, int M, int m_split, int N, int K, int splits, int64_t slab_stride,
                           int epi, int out_f32, hipStream_t stream);

int convert_f32_to_bf16(const float *src, bf16_t *dst, int64_t n, hipStream_t stream);

// [Nout, K] bf16 weight -> MFMA A-operand order (fragment blocks of 16 rows x 32 k, 1 KB contiguous each, row-tile major)
int pack_mfma16(const bf16_t *W, bf16_t *out, int Nout, int K, hipStream_t stream);
// tell the <= 64-row panel GEMM that `packed` is the pack_mfma16 copy of the row-major weight `w` (packed = nullptr: forget it)
void register_packed_weight(const void *w, const void *packed);

// <= 64-row panels: gemm_m64_kernel (a whole CU's LDS per workgroup; fastest alone) or, when off, the LDS-DMA ring (48 KB: leaves room for
// the workgroups of a concurrent stream).  Consulted at launch / capture time by linear_launch and linear_splitk_launch.
void set_panel_gemm(bool on);

// weight-streaming MFMA Linear for 1..16 rows of bf16 x (llm_rows16.hip); out bf16, or f32 with the plain epilogue
int linear_rows16_launch(const void *x, int ldx, const void *W, int ldw, const float *bias, const void *norm_w, float eps,
                         const void *residual, int ldr, void *out, int ldc, int M, int N, int K, int epi, int out_f32, hipStream_t stream);

}  // namespace ll
