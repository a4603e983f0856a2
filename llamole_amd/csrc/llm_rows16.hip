// Weight-streaming Linear for 3..16 token rows (batched decode of the HF LLM, SURVEY.md section 8 f2: 16 lock-step A* searches,
// 8 prompts per GPU): out[M,N] = epilogue(x[M,K] . W[N,K]^T + bias), bf16 operands, f32 accumulation on
// v_mfma_f32_16x16x32_bf16 with the token rows as the 16-wide MFMA column block.
//
// What shapes the kernel (tools/ingest_probe.hip, MI355X): a CU pulls 125-150 GB/s when the 16-byte loads of adjacent lanes
// are contiguous (>= 128 B per row segment) and 38 GB/s in the MFMA operand order (lane & 15 = row, lane >> 4 = 16-B piece:
// every lane group touches 16 different lines).  So nothing is loaded in operand order: each WAVE streams its own 16 weight
// rows in 512-byte (256-byte for the gated-MLP pair) row segments, two (four) rows per wave instruction, passes them
// through a wave-private, padded LDS image (pitch = segment + 16 B: conflict-free ds_read_b128 fragments) and multiplies;
// the matching x segment goes the same way.  The LDS image is private to the wave, so the main loop has no barrier at all;
// the next block's loads are in flight (in registers) while the current block is multiplied.
// A workgroup of `waves` waves (4 or 8) owns waves / ksplit row tiles; `ksplit` consecutive waves split K of one tile and
// their partial tiles are summed through LDS in wave order (deterministic).  What limits a wave is its own latency chain
// (load -> LDS -> fragment read -> MFMA), so the launcher splits K until there are ~14 waves per CU (tools/rows16_sweep.py):
// lm_head and the GIN template head 4 waves x ksplit 2, gate|up and q|k|v 4 x 4 (52 KB of LDS per workgroup: three per CU),
// o_proj / down_proj 8 x 8.
// Epilogues as ll_gemv_fused_bf16 (same intermediate bf16 roundings as PyTorch's op-by-op evaluation).  RMSNorm prologue
// (norm_w != NULL): x is multiplied by the norm weight while it is staged, x' = bf16(x * w), the waves accumulate sum(x^2)
// of their K slice on the side, and rsqrt(mean(x^2) + eps) of the token row scales the accumulator in the epilogue -- the
// same quantity as Qwen2RMSNorm + nn.Linear up to the place of one bf16 rounding (bf16(x * rstd) is not formed).
#include "common.h"

namespace ll {

typedef uint32_t r16_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 r16_bf16x8;
typedef __attribute__((ext_vector_type(4))) float r16_f32x4;

enum { R16_PLAIN = 0, R16_RESIDUAL = 1, R16_SILU_MUL = 2 };

__device__ __forceinline__ float r16_bfr(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <int EPI, int SEG, bool NORM>
__global__ __launch_bounds__(512) void rows16_kernel(const bf16_t *__restrict__ X, int ldx, const bf16_t *__restrict__ W, int ldw,
                                                     const float *__restrict__ bias, const bf16_t *__restrict__ normw, float eps,
                                                     const bf16_t *__restrict__ res, int ldr, void *__restrict__ Cv, int ldc, int M,
                                                     int N, int K, int ksplit, int out_f32) {
    constexpr int NT = EPI == R16_SILU_MUL ? 2 : 1;     // weight sub-tiles per wave (gate rows + the matching up rows)
    // SEG = bytes of a row per block (128 | 256 | 512)
    constexpr int PITCH = SEG + 16;                     // LDS row pitch: (PITCH / 4) % 64 == 4 -> 16 rows cover the 64 banks once
    constexpr int RPI = 1024 / SEG;                     // rows per wave instruction
    constexpr int IPT = 16 / RPI;                       // instructions per 16-row tile
    constexpr int LPR = SEG / 16;                       // lanes per row segment
    constexpr int KSTEPS = SEG / 64;                    // MFMA k-steps per block
    constexpr int WAVE_LDS = (NT + 1) * 16 * PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_r16[];
    const int tid = threadIdx.x, lane = tid & 63, waves = blockDim.x >> 6;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *wl = sm_r16 + wid * WAVE_LDS;        // [NT*16 weight rows | 16 x rows][PITCH]
    unsigned char *xl = wl + NT * 16 * PITCH;
    const int tpw = waves / ksplit;                     // tiles per workgroup
    const int tile = blockIdx.x * tpw + wid / ksplit, slice = wid % ksplit;
    const int ntiles = (N + 15) / 16;
    const bool tile_ok = tile < ntiles;
    // K range of this wave, in bytes of a row (multiples of 64 B = one MFMA k-step)
    const int steps = K / 32, per = (steps + ksplit - 1) / ksplit;
    const int kb = min(slice * per, steps) * 64, ke = min((slice + 1) * per, steps) * 64;
    const int lrow = lane / LPR, lcol = (lane % LPR) * 16;      // this lane's row within an instruction, byte within the segment
    const int n0 = tile * 16;
    r16_f32x4 acc[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) acc[s] = (r16_f32x4)(0.f);
    r16_u32x4 wr[NT][IPT], xr[IPT], nwr = (r16_u32x4)(0);
    float ssq[IPT];                                     // NORM: sum of squares of this lane's x elements, per row of its instructions
#pragma unroll
    for (int q = 0; q < IPT; ++q) ssq[q] = 0.f;
    auto load_block = [&](int k0) {
        const bool kin = k0 + lcol < ke;
        if (NORM) nwr = kin ? *reinterpret_cast<const r16_u32x4 *>(reinterpret_cast<const unsigned char *>(normw) + k0 + lcol) : (r16_u32x4)(0);
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int r = q * RPI + lrow;
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                int row = n0 + r;
                row = row < N ? row : N - 1;
                const unsigned char *p = reinterpret_cast<const unsigned char *>(W + ((int64_t)row + (int64_t)s * N) * ldw) + k0 + lcol;
                wr[s][q] = (kin && tile_ok) ? __builtin_nontemporal_load(reinterpret_cast<const r16_u32x4 *>(p)) : (r16_u32x4)(0);
            }
            const unsigned char *px = reinterpret_cast<const unsigned char *>(X + (int64_t)r * ldx) + k0 + lcol;
            xr[q] = (kin && r < M && tile_ok) ? *reinterpret_cast<const r16_u32x4 *>(px) : (r16_u32x4)(0);
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    if (kb < ke) load_block(kb);
    for (int k0 = kb; k0 < ke; k0 += SEG) {
        // registers -> the wave's LDS image (the previous block's fragment reads have retired: same wave, in order)
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int r = q * RPI + lrow;
#pragma unroll
            for (int s = 0; s < NT; ++s) *reinterpret_cast<r16_u32x4 *>(wl + (s * 16 + r) * PITCH + lcol) = wr[s][q];
            r16_u32x4 xv = xr[q];
            if (NORM) {     // x' = bf16(x * w_norm); the row's rsqrt(mean(x^2) + eps) multiplies the accumulator in the epilogue
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float a = __uint_as_float(xv[t] << 16), b = __uint_as_float(xv[t] & 0xffff0000u);
                    ssq[q] = fmaf(a, a, ssq[q]);
                    ssq[q] = fmaf(b, b, ssq[q]);
                    const float wa = __uint_as_float(nwr[t] << 16), wb = __uint_as_float(nwr[t] & 0xffff0000u);
                    xv[t] = (uint32_t)f32_to_bf16(a * wa) | ((uint32_t)f32_to_bf16(b * wb) << 16);
                }
            }
            *reinterpret_cast<r16_u32x4 *>(xl + r * PITCH + lcol) = xv;
        }
        if (k0 + SEG < ke) load_block(k0 + SEG);       // next block in flight while this one is multiplied
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const r16_bf16x8 b = *reinterpret_cast<const r16_bf16x8 *>(xl + fr * PITCH + (ks * 4 + fq) * 16);
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                const r16_bf16x8 a = *reinterpret_cast<const r16_bf16x8 *>(wl + (s * 16 + fr) * PITCH + (ks * 4 + fq) * 16);
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[s], 0, 0, 0);
            }
        }
    }
    // acc[s][j] = C[weight row n0 + (lane>>4)*4 + j][token row lane & 15]
    if (NORM) {             // lanes of one row segment -> one sum per row of this wave's K slice
#pragma unroll
        for (int q = 0; q < IPT; ++q)
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) ssq[q] += __shfl_xor(ssq[q], off, 64);
    }
    float rstd = 1.f;
    if (ksplit > 1) {
        __syncthreads();                                // every wave is done with its LDS image
        float *part = reinterpret_cast<float *>(sm_r16);               // [waves][NT][64 lanes][4] | [waves][16] sums of squares
        float *psq = part + waves * NT * 256;
#pragma unroll
        for (int s = 0; s < NT; ++s) *reinterpret_cast<r16_f32x4 *>(part + ((wid * NT + s) * 64 + lane) * 4) = acc[s];
        if (NORM && lane % LPR == 0) {
#pragma unroll
            for (int q = 0; q < IPT; ++q) psq[wid * 16 + q * RPI + lrow] = ssq[q];
        }
        __syncthreads();
        if (slice != 0) return;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            r16_f32x4 t = acc[s];
            for (int w = 1; w < ksplit; ++w) t += *reinterpret_cast<const r16_f32x4 *>(part + (((wid + w) * NT + s) * 64 + lane) * 4);
            acc[s] = t;
        }
        if (NORM) {
            float t = psq[wid * 16 + (lane & 15)];
            for (int w = 1; w < ksplit; ++w) t += psq[(wid + w) * 16 + (lane & 15)];
            rstd = rsqrtf(t / (float)K + eps);
        }
    } else if (NORM) {      // the wave owns the whole K: row sums through its own LDS image (same wave: in order)
        float *psq = reinterpret_cast<float *>(wl);
        if (lane % LPR == 0) {
#pragma unroll
            for (int q = 0; q < IPT; ++q) psq[q * RPI + lrow] = ssq[q];
        }
        rstd = rsqrtf(psq[lane & 15] / (float)K + eps);
    }
    if (NORM) {
#pragma unroll
        for (int s = 0; s < NT; ++s) acc[s] *= rstd;
    }
    if (!tile_ok) return;
    const int m = lane & 15, nb = n0 + (lane >> 4) * 4;
    if (m >= M) return;
    if (EPI == R16_PLAIN && out_f32) {      // f32 logits (GIN template head): acc + bias, no rounding
        float *dst = reinterpret_cast<float *>(Cv) + (int64_t)m * ldc + nb;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[0][j] + ((bias && nb + j < N) ? bias[nb + j] : 0.f);
        if (nb + 3 < N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (nb + j < N) dst[j] = v[j];
        }
        return;
    }
    bf16_t *C = reinterpret_cast<bf16_t *>(Cv);
    uint16_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = nb + j;
        const bool ok = n < N;
        if (EPI == R16_SILU_MUL) {
            const float g = r16_bfr(acc[0][j] + ((bias && ok) ? bias[n] : 0.f));
            const float u = r16_bfr(acc[NT - 1][j] + ((bias && ok) ? bias[n + N] : 0.f));
            o[j] = f32_to_bf16(r16_bfr(silu(g)) * u);
        } else {
            float v = acc[0][j] + ((bias && ok) ? bias[n] : 0.f);
            if (EPI == R16_RESIDUAL) v = (ok ? bf16_to_f32(res[(int64_t)m * ldr + n]) : 0.f) + r16_bfr(v);
            o[j] = f32_to_bf16(v);
        }
    }
    bf16_t *dst = C + (int64_t)m * ldc + nb;
    if (nb + 3 < N && ((reinterpret_cast<uintptr_t>(dst) & 7) == 0)) {
        *reinterpret_cast<uint2 *>(dst) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (nb + j < N) dst[j] = o[j];
    }
}

static int g_rows16_geom = 0;     // 0: by tile count; else seg << 16 | waves << 8 | ksplit (tuning)

template <int EPI, int SEG, bool NORM>
static int launch_rows16_seg(int waves, int ksplit, hipStream_t s, const bf16_t *X, int ldx, const bf16_t *W, int ldw, const float *bias,
                             const bf16_t *normw, float eps, const bf16_t *res, int ldr, void *C, int ldc, int M, int N, int K, int out_f32) {
    constexpr int NT = EPI == R16_SILU_MUL ? 2 : 1;
    const int ntiles = (N + 15) / 16;
    const size_t lds = (size_t)waves * (NT + 1) * 16 * (SEG + 16);
    LL_CHECK(lds <= 160 * 1024, "ll_linear_rows16_bf16: %d waves x %d-byte segments need %zu bytes of LDS", waves, SEG, lds);
    static size_t attr_lds = 0;       // > 64 KB of dynamic LDS needs the attribute
    if (lds > attr_lds) {
        LL_HIP(hipFuncSetAttribute((const void *)rows16_kernel<EPI, SEG, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int tpw = waves / ksplit;
    hipLaunchKernelGGL((rows16_kernel<EPI, SEG, NORM>), dim3((ntiles + tpw - 1) / tpw), dim3(waves * 64), lds, s, X, ldx, W, ldw, bias, normw,
                       eps, res, ldr, C, ldc, M, N, K, ksplit, out_f32);
    return LL_OK;
}

template <int EPI, bool NORM>
static int launch_rows16(hipStream_t s, const bf16_t *X, int ldx, const bf16_t *W, int ldw, const float *bias, const bf16_t *normw, float eps,
                         const bf16_t *res, int ldr, void *C, int ldc, int M, int N, int K, int out_f32) {
    const int ntiles = (N + 15) / 16;
    int seg, waves, ksplit;
    if (g_rows16_geom) {
        seg = g_rows16_geom >> 16, waves = (g_rows16_geom >> 8) & 255, ksplit = g_rows16_geom & 255;
    } else {
        // enough waves to hide the load -> LDS -> MFMA latency chain of each one: split K until tiles x ksplit >= ~14 per CU
        // (tools/rows16_sweep.py: gate|up 4 waves x ksplit 4, q|k|v / o_proj / down_proj 8 x 8, lm_head 4 x 1)
        seg = EPI == R16_SILU_MUL ? 256 : 512;
        ksplit = 1;
        while (ksplit < 8 && ntiles * ksplit < 3500) ksplit *= 2;
        waves = ksplit <= 4 ? 4 : 8;
        if (ksplit == 8 && ntiles * 4 >= 1100) seg = 256, waves = 4, ksplit = 4;      // q|k|v (288 tiles): 11.1 us vs 12.5
        if (ksplit == 1) seg = 256, ksplit = 2;     // thousands of tiles (lm_head, GIN template head): 4 x 2 beats 4 x 1 (127 vs 141 us on the head)
    }
    if (seg == 128) return launch_rows16_seg<EPI, 128, NORM>(waves, ksplit, s, X, ldx, W, ldw, bias, normw, eps, res, ldr, C, ldc, M, N, K, out_f32);
    if (seg == 256) return launch_rows16_seg<EPI, 256, NORM>(waves, ksplit, s, X, ldx, W, ldw, bias, normw, eps, res, ldr, C, ldc, M, N, K, out_f32);
    return launch_rows16_seg<EPI, 512, NORM>(waves, ksplit, s, X, ldx, W, ldw, bias, normw, eps, res, ldr, C, ldc, M, N, K, out_f32);
}

int linear_rows16_launch(const void *x, int ldx, const void *W, int ldw, const float *bias, const void *norm_w, float eps,
                         const void *residual, int ldr, void *out, int ldc, int M, int N, int K, int epi, int out_f32, hipStream_t s) {
    LL_CHECK(x && W && out, "ll_linear_rows16_bf16: null argument");
    LL_CHECK(M >= 1 && M <= 16, "ll_linear_rows16_bf16: M=%d rows (1..16)", M);
    LL_CHECK(N >= 1 && K >= 32 && K % 32 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "ll_linear_rows16_bf16: K must be a multiple of 32, ldx / ldw of 8");
    LL_CHECK(epi >= R16_PLAIN && epi <= R16_SILU_MUL, "ll_linear_rows16_bf16: epilogue %d", epi);
    LL_CHECK(epi != R16_RESIDUAL || residual, "ll_linear_rows16_bf16: residual epilogue without a residual");
    LL_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0, "ll_linear_rows16_bf16: operands must be 16-byte aligned");
    const bf16_t *X = (const bf16_t *)x, *Wt = (const bf16_t *)W, *rs = (const bf16_t *)residual;
    LL_CHECK(!out_f32 || epi == R16_PLAIN, "ll_linear_rows16_bf16: f32 output only with the plain epilogue");
    const bf16_t *nw = (const bf16_t *)norm_w;
    LL_CHECK(!nw || ((uintptr_t)nw & 15) == 0, "ll_linear_rows16_bf16: norm weight must be 16-byte aligned");
    if (nw) {
        if (epi == R16_PLAIN) LL_TRY((launch_rows16<R16_PLAIN, true>(s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, out, ldc, M, N, K, out_f32)));
        else if (epi == R16_RESIDUAL) LL_TRY((launch_rows16<R16_RESIDUAL, true>(s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, out, ldc, M, N, K, 0)));
        else LL_TRY((launch_rows16<R16_SILU_MUL, true>(s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, out, ldc, M, N, K, 0)));
    } else {
        if (epi == R16_PLAIN) LL_TRY((launch_rows16<R16_PLAIN, false>(s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, out, ldc, M, N, K, out_f32)));
        else if (epi == R16_RESIDUAL) LL_TRY((launch_rows16<R16_RESIDUAL, false>(s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, out, ldc, M, N, K, 0)));
        else LL_TRY((launch_rows16<R16_SILU_MUL, false>(s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, out, ldc, M, N, K, 0)));
    }
    LL_LAUNCH_CHECK();
    return LL_OK;
}

}  // namespace ll

using namespace ll;

extern "C" {

int ll_linear_rows16_bf16(const void *x, int ldx, const void *W, int ldw, const float *bias, const void *norm_w, float eps,
                          const void *residual, int ldr, void *out, int ldc, int M, int N, int K, int epi, void *stream) {
    return linear_rows16_launch(x, ldx, W, ldw, bias, norm_w, eps, residual, ldr, out, ldc, M, N, K, epi, 0, (hipStream_t)stream);
}

#if LL_TUNING
int ll_set_rows16_geometry(int seg, int waves, int ksplit) {
    const int old = g_rows16_geom;
    const bool ok = (seg == 128 || seg == 256 || seg == 512) && (waves == 4 || waves == 8) && ksplit >= 1 && ksplit <= waves &&
                    waves % ksplit == 0;
    g_rows16_geom = ok ? (seg << 16 | waves << 8 | ksplit) : 0;
    return old;
}
#endif

// Times ll_linear_rows16_bf16 on synthetic operands over `nweights` distinct weight matrices (defeats the Infinity Cache).
#if LL_TUNING
int ll_rows16_bench(int M, int N, int K, int epi, int norm, int iters, int nweights, float *ms) {
    LL_CHECK(ms && iters > 0 && nweights > 0 && M >= 1 && M <= 16, "bad argument");
    const int out_f32 = (epi & 0x100) ? 1 : 0;      // epi | 0x100: f32 output (the GIN template head), plain epilogue only
    epi &= 0xff;
    const int rowsW = epi == R16_SILU_MUL ? 2 * N : N;
    bf16_t *X = nullptr, *W = nullptr, *C = nullptr, *R = nullptr;
    LL_HIP(hipMalloc(&X, (size_t)16 * K * 2));
    LL_HIP(hipMalloc(&W, (size_t)nweights * rowsW * K * 2));
    LL_HIP(hipMalloc(&C, (size_t)16 * N * 4));
    LL_HIP(hipMalloc(&R, (size_t)16 * N * 2));
    LL_HIP(hipMemset(X, 0x11, (size_t)16 * K * 2));
    LL_HIP(hipMemset(R, 0x11, (size_t)16 * N * 2));
    LL_HIP(hipMemset(W, 0x11, (size_t)nweights * rowsW * K * 2));
    hipStream_t st;
    LL_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    LL_HIP(hipEventCreate(&e0));
    LL_HIP(hipEventCreate(&e1));
    int rc = LL_OK;
    for (int pass = 0; pass < 2 && rc == LL_OK; ++pass) {
        if (pass == 1) (void)hipEventRecord(e0, st);
        for (int i = 0; i < (pass ? iters : nweights) && rc == LL_OK; ++i)
            rc = linear_rows16_launch(X, K, W + (size_t)(i % nweights) * rowsW * K, K, nullptr, norm ? X : nullptr, 1e-6f, R, N, C, N, M, N, K, epi, out_f32, st);
    }
    (void)hipEventRecord(e1, st);
    hipError_t he = hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms = t / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    (void)hipFree(X);
    (void)hipFree(W);
    (void)hipFree(C);
    (void)hipFree(R);
    if (rc != LL_OK) return rc;
    LL_HIP(he);
    return LL_OK;
}
#endif

}  // extern "C"
