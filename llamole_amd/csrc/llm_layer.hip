// Decode-step layer kernels for an untouched HuggingFace Llama-family LLM: the twelve launches
// of one decoder layer at batch <= 4 (RMSNorm, q/k/v GEMV, rotary, KV append, attention, o_proj GEMV, add, RMSNorm,
// gate/up GEMV, SiLU*mul, down GEMV, add) become five:
//   ll_gemv_fused_bf16       x W^T with an optional RMSNorm prologue on x and an epilogue: bias | residual add | SiLU(gate)*up
//   ll_decode_attn_rope_bf16 rotary embedding of q and the new k, KV-cache append, GQA attention over the static cache
// The arithmetic -- f32 accumulation order, every intermediate bf16 rounding PyTorch's op-by-op evaluation implies --
// is the one of the unfused kernels (gemv_bf16_kernel, rmsnorm/rope/silu_mul/kv_append/decode_attn in llm_ops.hip),
// so results are bit-identical to them; what goes away is seven launch boundaries and seven HBM/L2 round trips of
// [B, hidden]-sized vectors per layer.
#include "common.h"
#include "attn_decode.h"

namespace ll {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bfr2(float v) { return bf16_to_f32(f32_to_bf16(v)); }   // round through bf16

enum { GEMV_PLAIN = 0, GEMV_RESIDUAL = 1, GEMV_SILU_MUL = 2 };
// 16-byte weight loads per row in flight per lane at one token row (x 2 rows per wave).  Re-swept once the kernels were built with
// kernel-argument preloading: 4 beats the earlier 8 (q|k|v 9.2 -> 8.2 us, LLM part of a molecule 353.6 -> 347.1 ms with the
// plain GEMV of lm_head at 4 as well); the FMA order per output row does not depend on it.
#ifndef LL_GEMV_UNR
#define LL_GEMV_UNR 4
#endif
#ifndef LL_GEMV_STAGE_UNR
#define LL_GEMV_STAGE_UNR 4
#endif

template <bool NT> __device__ __forceinline__ u32x4 ldw16(const bf16_t *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return *reinterpret_cast<const u32x4 *>(p);
}

template <int UNR, bool NT>
__device__ __forceinline__ void gemv_load_w(u32x4 (&wv)[UNR][2], const bf16_t *const (&wr)[2], int c0, int nchunk, bool active) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int c = c0 + u * 64;
#pragma unroll
        for (int r = 0; r < 2; ++r) wv[u][r] = (active && c < nchunk) ? ldw16<NT>(wr[r] + c * 8) : (u32x4)(0);
    }
}

// acc += w . x over UNR 16-byte chunks per lane; x rows at xp + m*ldx (LDS when XLDS)
template <int MROWS, int UNR, bool XLDS>
__device__ __forceinline__ void gemv_fma(float (&acc)[MROWS][2], const u32x4 (&wv)[UNR][2], const bf16_t *xp, int64_t ldx, int c0,
                                         int nchunk) {
    u32x4 xv[UNR][MROWS];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int c = c0 + u * 64;
#pragma unroll
        for (int m = 0; m < MROWS; ++m) xv[u][m] = c < nchunk ? *reinterpret_cast<const u32x4 *>(xp + (int64_t)m * ldx + c * 8) : (u32x4)(0);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
        for (int m = 0; m < MROWS; ++m) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float a = acc[m][r];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    a = fmaf(__uint_as_float(wv[u][r][t] << 16), __uint_as_float(xv[u][m][t] << 16), a);
                    a = fmaf(__uint_as_float(wv[u][r][t] & 0xffff0000u), __uint_as_float(xv[u][m][t] & 0xffff0000u), a);
                }
                acc[m][r] = a;
            }
        }
    }
}

// Each wave owns two weight rows (GEMV_SILU_MUL: gate row n and up row n + N; otherwise rows 2w, 2w+1); a lane reads
// 16 B of each row per step with UNR steps in flight, x comes from LDS (NORM: the workgroup normalises x once, with
// the first weight loads already in flight) or from L1/L2.  f32 FMA chains in the lane/chunk order of gemv_bf16_kernel.
template <int MROWS, bool NORM, int EPI, bool NT, int XC>
__global__ __launch_bounds__(256) void gemv_fused_kernel(const bf16_t *__restrict__ X, int ldx, const bf16_t *__restrict__ W,
                                                         int ldw, const float *__restrict__ bias,
                                                         const bf16_t *__restrict__ normw, float eps,
                                                         const bf16_t *__restrict__ res, int ldr, bf16_t *__restrict__ C,
                                                         int ldc, int N, int K) {
    constexpr int R = 2, UNR = MROWS == 1 ? LL_GEMV_UNR : 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_gemv[];
    bf16_t *xs = reinterpret_cast<bf16_t *>(sm_gemv);   // NORM: [MROWS][K] normalised x
    __shared__ float red[MROWS][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = blockIdx.x * 4 + (tid >> 6);
    const int n0 = EPI == GEMV_SILU_MUL ? wave : wave * R;
    const bool active = n0 < N;     // whole waves; inactive waves still take part in the NORM barriers
    const bf16_t *wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = EPI == GEMV_SILU_MUL ? (int64_t)(n0 < N ? n0 : N - 1) + (int64_t)r * N : (n0 + r < N ? n0 + r : N - 1);
        wr[r] = W + row * ldw;
    }
    const int nchunk = K / 8;
    // XC 16-byte chunks of x per thread (K <= 2048 XC): keeping it at 2 for hidden sizes <= 4096 holds the kernel at
    // <= 128 VGPRs = 4 waves per SIMD
    u32x4 xv0[MROWS][XC], nw0[XC];
    if (NORM) {   // x (and the norm weight) first: vmcnt retires in order, the weight loads below stay in flight
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            const int ch = tid + c * 256;
            const bool ok = ch < nchunk;
#pragma unroll
            for (int m = 0; m < MROWS; ++m) xv0[m][c] = ok ? *reinterpret_cast<const u32x4 *>(X + (int64_t)m * ldx + ch * 8) : (u32x4)(0);
            nw0[c] = ok ? *reinterpret_cast<const u32x4 *>(normw + ch * 8) : (u32x4)(0);
        }
    }
    u32x4 wv[UNR][R];
    if (NORM) gemv_load_w<UNR, NT>(wv, wr, lane, nchunk, active);
    if (NORM) {
        // same summation order as rmsnorm_bf16_kernel: per-thread fmaf chain over its chunks, wave_sum, 4 partials
#pragma unroll
        for (int m = 0; m < MROWS; ++m) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < XC; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float a = __uint_as_float(xv0[m][c][t] << 16), b = __uint_as_float(xv0[m][c][t] & 0xffff0000u);
                    ss = fmaf(a, a, ss);
                    ss = fmaf(b, b, ss);
                }
            ss = wave_sum(ss);
            if (lane == 0) red[m][tid >> 6] = ss;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MROWS; ++m) {
            const float var = (red[m][0] + red[m][1] + red[m][2] + red[m][3]) / (float)K;
            const float rstd = rsqrtf(var + eps);
#pragma unroll
            for (int c = 0; c < XC; ++c) {
                const int ch = tid + c * 256;
                if (ch < nchunk) {
                    u32x4 o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float a = bfr2(__uint_as_float(xv0[m][c][t] << 16) * rstd) * __uint_as_float(nw0[c][t] << 16);
                        const float b = bfr2(__uint_as_float(xv0[m][c][t] & 0xffff0000u) * rstd) * __uint_as_float(nw0[c][t] & 0xffff0000u);
                        o[t] = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
                    }
                    *reinterpret_cast<u32x4 *>(xs + (int64_t)m * K + ch * 8) = o;
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;
    float acc[MROWS][R];
#pragma unroll
    for (int m = 0; m < MROWS; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
    int cbeg = lane;
    if (NORM) {   // first block: its weights were requested before the normalisation
        gemv_fma<MROWS, UNR, true>(acc, wv, xs, K, lane, nchunk);
        cbeg += 64 * UNR;
    }
    for (int c0 = cbeg; c0 < nchunk; c0 += 64 * UNR) {
        if (NORM) {
            gemv_load_w<UNR, NT>(wv, wr, c0, nchunk, true);
            gemv_fma<MROWS, UNR, true>(acc, wv, xs, K, c0, nchunk);
        } else {
            // x chunk u right behind weight chunk u (loads retire in order: the first FMAs need not wait for all of W)
            u32x4 xv[UNR][MROWS];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = c0 + u * 64;
                const bool ok = c < nchunk;
#pragma unroll
                for (int r = 0; r < R; ++r) wv[u][r] = ok ? ldw16<NT>(wr[r] + c * 8) : (u32x4)(0);
#pragma unroll
                for (int m = 0; m < MROWS; ++m) xv[u][m] = ok ? *reinterpret_cast<const u32x4 *>(X + (int64_t)m * ldx + c * 8) : (u32x4)(0);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
#pragma unroll
                for (int m = 0; m < MROWS; ++m) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float a = acc[m][r];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            a = fmaf(__uint_as_float(wv[u][r][t] << 16), __uint_as_float(xv[u][m][t] << 16), a);
                            a = fmaf(__uint_as_float(wv[u][r][t] & 0xffff0000u), __uint_as_float(xv[u][m][t] & 0xffff0000u), a);
                        }
                        acc[m][r] = a;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MROWS; ++m) {
        float v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = wave_sum(acc[m][r]);
        if (lane == 0) {
            if (EPI == GEMV_SILU_MUL) {
                const float g = bfr2(v[0] + (bias ? bias[n0] : 0.f)), up = bfr2(v[1] + (bias ? bias[n0 + N] : 0.f));
                C[(int64_t)m * ldc + n0] = f32_to_bf16(bfr2(silu(g)) * up);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (n0 + r < N) {
                        float o = v[r] + (bias ? bias[n0 + r] : 0.f);
                        if (EPI == GEMV_RESIDUAL) o = bf16_to_f32(res[(int64_t)m * ldr + n0 + r]) + bfr2(o);
                        C[(int64_t)m * ldc + n0 + r] = f32_to_bf16(o);
                    }
                }
            }
        }
    }
}

// gemv_fused_kernel's prologue-free form (down_proj: no RMSNorm, long K) for ONE token row with x staged in LDS: in the
// generic kernel every wave re-reads its x chunks from L1/L2 next to the weight stream (8 more loads per lane and block,
// 24 in flight); here the workgroup copies x into LDS once, with its first weight loads already in flight, and the stream
// keeps 16 loads per lane.  Same FMA order per output row, so the result is bit-identical.
template <int EPI>
__global__ __launch_bounds__(256) void gemv_stage_kernel(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, int ldw,
                                                         const float *__restrict__ bias, const bf16_t *__restrict__ res,
                                                         bf16_t *__restrict__ C, int N, int K) {
    constexpr int R = 2, UNR = LL_GEMV_STAGE_UNR, XCS = 10;       // XCS 16-byte chunks of x per thread: K <= 20480
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_gemvs[];
    bf16_t *xs = reinterpret_cast<bf16_t *>(sm_gemvs);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = blockIdx.x * 4 + (tid >> 6);
    const int n0 = wave * R;
    const bool active = n0 < N;
    const bf16_t *wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = W + (int64_t)(n0 + r < N ? n0 + r : N - 1) * ldw;
    const int nchunk = K / 8;
    u32x4 xv0[XCS];
#pragma unroll
    for (int c = 0; c < XCS; ++c) {
        const int ch = tid + c * 256;
        xv0[c] = ch < nchunk ? *reinterpret_cast<const u32x4 *>(X + ch * 8) : (u32x4)(0);
    }
    u32x4 wv[UNR][R];
    gemv_load_w<UNR, true>(wv, wr, lane, nchunk, active);
#pragma unroll
    for (int c = 0; c < XCS; ++c) {
        const int ch = tid + c * 256;
        if (ch < nchunk) *reinterpret_cast<u32x4 *>(xs + ch * 8) = xv0[c];
    }
    __syncthreads();
    if (!active) return;
    float acc[1][R] = {{0.f, 0.f}};
    gemv_fma<1, UNR, true>(acc, wv, xs, K, lane, nchunk);
    for (int c0 = lane + 64 * UNR; c0 < nchunk; c0 += 64 * UNR) {
        gemv_load_w<UNR, true>(wv, wr, c0, nchunk, true);
        gemv_fma<1, UNR, true>(acc, wv, xs, K, c0, nchunk);
    }
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = wave_sum(acc[0][r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (n0 + r < N) {
                float o = v[r] + (bias ? bias[n0 + r] : 0.f);
                if (EPI == GEMV_RESIDUAL) o = bf16_to_f32(res[n0 + r]) + bfr2(o);
                C[n0 + r] = f32_to_bf16(o);
            }
        }
    }
}

// Rotary embedding + KV append + GQA decode attention for ONE new position per sequence.
// grid (nh, B).  qkv row b = [q: nh*D | k: nkv*D | v: nkv*D] (output of the fused q/k/v GEMV).  The new key / value of
// the head's KV group is rotated in LDS and used from there; the first query head of each group also stores it to the
// cache at *pos.  Nobody reads cache slot *pos in this launch, so there is no cross-workgroup dependency.
template <int D>
__global__ __launch_bounds__(ATTN_THREADS) void decode_attn_rope_kernel(const bf16_t *__restrict__ qkv, int64_t ld_qkv,
                                                               const bf16_t *__restrict__ cs, const bf16_t *__restrict__ sn,
                                                               int64_t cs0, bf16_t *__restrict__ K, bf16_t *__restrict__ V,
                                                               const long long *__restrict__ pos_ptr,
                                                               const unsigned char *__restrict__ mask, int64_t ms0,
                                                               bf16_t *__restrict__ out, int nh, int nkv, int maxlen, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm_attn2[];
    float *qs = sm_attn2;                 // [D]     rotated query, f32 of its bf16 value
    float *part = qs + D;                 // [waves][rows per load][D] partial outputs = ATTN_PART_FLOATS for either D
    bf16_t *kn = reinterpret_cast<bf16_t *>(part + ATTN_PART_FLOATS);    // [D] rotated new key (bf16)
    bf16_t *vn = kn + D;                  // [D] new value
    float *sc = reinterpret_cast<float *>(vn + D);           // [maxlen] scores -> probabilities
    __shared__ float red[2 * ATTN_WAVES];
    const int h = blockIdx.x, b = blockIdx.y;
    const int group = nh / nkv, kvh = h / group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long p = *pos_ptr;
    const bool pvalid = p >= 0 && p < maxlen;
    const bf16_t *row = qkv + b * ld_qkv;
    const bf16_t *c = cs + b * cs0, *sv = sn + b * cs0;
    constexpr int half = D / 2;
    bf16_t *Kb = K + ((int64_t)b * nkv + kvh) * maxlen * D;
    bf16_t *Vb = V + ((int64_t)b * nkv + kvh) * maxlen * D;
    const unsigned char *mrow = mask + b * ms0;
    AttnTile0<D> t0;      // first 256 keys / values / mask bytes: requested before the rotary arithmetic
    attn_prefetch<D>(t0, Kb, Vb, mrow, maxlen, tid, lane, wave);
    if (wave == 0) {
        if (lane < half) {
            const bf16_t *src = row + h * D;
            const float x1 = bf16_to_f32(src[lane]), x2 = bf16_to_f32(src[lane + half]);
            const float c1 = bf16_to_f32(c[lane]), c2 = bf16_to_f32(c[lane + half]);
            const float s1 = bf16_to_f32(sv[lane]), s2 = bf16_to_f32(sv[lane + half]);
            qs[lane] = bfr2(bfr2(x1 * c1) + bfr2(-x2 * s1));
            qs[lane + half] = bfr2(bfr2(x2 * c2) + bfr2(x1 * s2));
        }
    } else if (wave == 1) {
        if (lane < half) {
            const bf16_t *src = row + (nh + kvh) * D;
            const float x1 = bf16_to_f32(src[lane]), x2 = bf16_to_f32(src[lane + half]);
            const float c1 = bf16_to_f32(c[lane]), c2 = bf16_to_f32(c[lane + half]);
            const float s1 = bf16_to_f32(sv[lane]), s2 = bf16_to_f32(sv[lane + half]);
            const bf16_t k1 = f32_to_bf16(bfr2(x1 * c1) + bfr2(-x2 * s1)), k2 = f32_to_bf16(bfr2(x2 * c2) + bfr2(x1 * s2));
            kn[lane] = k1;
            kn[lane + half] = k2;
            if (pvalid && h % group == 0) {
                Kb[p * D + lane] = k1;
                Kb[p * D + lane + half] = k2;
            }
        }
    } else if (wave == 2) {
        if (lane < half) {
            const bf16_t *src = row + (nh + nkv + kvh) * D;
            const uint32_t v2 = *reinterpret_cast<const uint32_t *>(src + lane * 2);
            *reinterpret_cast<uint32_t *>(vn + lane * 2) = v2;
            if (pvalid && h % group == 0) *reinterpret_cast<uint32_t *>(Vb + p * D + lane * 2) = v2;
        }
    }
    __syncthreads();
    attn_finish<D, true>(t0, qs, part, sc, red, Kb, Vb, mrow, maxlen, pvalid ? p : -1, kn, vn, scale,
                         out + ((int64_t)b * nh + h) * D, tid, lane, wave);
}

// The same for S consecutive NEW positions of every sequence (the query-token forward on top of the decode's cache,
// modeling_llamole.py:641-646: <design_start> + 8 body tokens at cache slots *pos .. *pos + S - 1).  grid (nh, B*S), row r = b*S + s.
// Query row s attends to the cache up to slot *pos + s, i.e. also to the keys of rows 0..s of THIS launch: the workgroup rotates and
// stores those rows of its KV head itself before it reads them (every workgroup of the sequence that needs a row writes the same bits,
// so nobody waits for anybody), then runs the plain cache attention of decode_attn_bf16_kernel -- bit for bit what rope_bf16_kernel +
// kv_append_bf16_kernel + decode_attn_bf16_kernel compute in three launches.
template <int D>
__global__ __launch_bounds__(ATTN_THREADS) void suffix_attn_rope_kernel(const bf16_t *__restrict__ qkv, int64_t ld_qkv,
                                                               const bf16_t *__restrict__ cs, const bf16_t *__restrict__ sn, bf16_t *K,
                                                               bf16_t *V, const long long *__restrict__ pos_ptr,
                                                               const unsigned char *__restrict__ mask, bf16_t *__restrict__ out, int nh,
                                                               int nkv, int S, int maxlen, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm_attn3[];
    float *qs = sm_attn3;                 // [D]  rotated query, f32 of its bf16 value
    float *part = qs + D;                 // ATTN_PART_FLOATS
    float *sc = part + ATTN_PART_FLOATS;  // [maxlen]
    __shared__ float red[2 * ATTN_WAVES];
    const int h = blockIdx.x, r = blockIdx.y;
    const int b = r / S, s = r - b * S;
    const int group = nh / nkv, kvh = h / group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long p0 = *pos_ptr;
    constexpr int half = D / 2;
    bf16_t *Kb = K + ((int64_t)b * nkv + kvh) * maxlen * D;
    bf16_t *Vb = V + ((int64_t)b * nkv + kvh) * maxlen * D;
    const unsigned char *mrow = mask + (int64_t)r * maxlen;
    if (wave == ATTN_WAVES - 1) {
        if (lane < half) {
            const bf16_t *src = qkv + (int64_t)r * ld_qkv + h * D;
            const bf16_t *c = cs + (int64_t)r * D, *sv = sn + (int64_t)r * D;
            const float x1 = bf16_to_f32(src[lane]), x2 = bf16_to_f32(src[lane + half]);
            const float c1 = bf16_to_f32(c[lane]), c2 = bf16_to_f32(c[lane + half]);
            const float s1 = bf16_to_f32(sv[lane]), s2 = bf16_to_f32(sv[lane + half]);
            qs[lane] = bfr2(bfr2(x1 * c1) + bfr2(-x2 * s1));
            qs[lane + half] = bfr2(bfr2(x2 * c2) + bfr2(x1 * s2));
        }
    }
    for (int t = wave; t <= s; t += ATTN_WAVES) {
        const long long p = p0 + t;
        if (p < 0 || p >= maxlen || lane >= half) continue;
        const int rt = b * S + t;
        const bf16_t *src = qkv + (int64_t)rt * ld_qkv + (nh + kvh) * D;
        const bf16_t *c = cs + (int64_t)rt * D, *sv = sn + (int64_t)rt * D;
        const float x1 = bf16_to_f32(src[lane]), x2 = bf16_to_f32(src[lane + half]);
        const float c1 = bf16_to_f32(c[lane]), c2 = bf16_to_f32(c[lane + half]);
        const float s1 = bf16_to_f32(sv[lane]), s2 = bf16_to_f32(sv[lane + half]);
        Kb[p * D + lane] = f32_to_bf16(bfr2(x1 * c1) + bfr2(-x2 * s1));
        Kb[p * D + lane + half] = f32_to_bf16(bfr2(x2 * c2) + bfr2(x1 * s2));
        const bf16_t *vsrc = qkv + (int64_t)rt * ld_qkv + (nh + nkv + kvh) * D;
        *reinterpret_cast<uint32_t *>(Vb + p * D + lane * 2) = *reinterpret_cast<const uint32_t *>(vsrc + lane * 2);
    }
    __threadfence_block();
    __syncthreads();            // the rows are in the cache (this CU's L1 is write-through and shared by the workgroup's waves)
    AttnTile0<D> t0;
    attn_prefetch<D>(t0, Kb, Vb, mrow, maxlen, tid, lane, wave);
    attn_finish<D, false>(t0, qs, part, sc, red, Kb, Vb, mrow, maxlen, -1, nullptr, nullptr, scale, out + ((int64_t)r * nh + h) * D, tid,
                          lane, wave);
}

// The same for MANY sequences per step (17..64: BASELINE configs[3] on one GPU): one workgroup per (KV head, sequence) serves all G = nh / nkv
// query heads of the group, so a key / value row is fetched once instead of G times (at 64 sequences x 32 heads the per-head kernel took
// 55 us per layer, most of it G-fold re-reads through L2: profiles/r6_llama64_kernel_stats.csv).  Same access pattern as attn_decode.h --
// a wave instruction covers RPI whole rows, lane (r, c) holds 16 bytes of row r -- with G query pieces in registers per lane: G dot
// products per loaded key piece, G axpys per value piece.  Softmax statistics in f32 per head.  The first 256 keys / values are requested
// before the rotary arithmetic.  Arithmetic per head as decode_attn_rope_kernel up to the f32 summation order of P.V.
template <int D, int G, int W>      // W waves per workgroup (8 | 16): a wave takes 256 / W keys of a tile
__global__ __launch_bounds__(64 * W) void decode_attn_rope_gqa_kernel(const bf16_t *__restrict__ qkv, int64_t ld_qkv,
                                                                   const bf16_t *__restrict__ cs, const bf16_t *__restrict__ sn,
                                                                   int64_t cs0, bf16_t *__restrict__ K, bf16_t *__restrict__ V,
                                                                   const long long *__restrict__ pos_ptr,
                                                                   const unsigned char *__restrict__ mask, int64_t ms0,
                                                                   bf16_t *__restrict__ out, int nh, int nkv, int maxlen, float scale) {
    constexpr int LPR = AttnGeom<D>::LPR, RPI = AttnGeom<D>::RPI, KPW = 256 / W, NI = KPW / RPI, half = D / 2, THREADS = 64 * W;
    auto key_of = [](int j0, int wave, int i, int lane) { return j0 + wave * KPW + i * RPI + lane / LPR; };
    extern __shared__ __attribute__((aligned(16))) float sm_attn3[];
    float *qs = sm_attn3;                         // [G][D] rotated queries, f32 of their bf16 values
    float *part = qs + G * D;                     // [waves][G][D] partial outputs
    bf16_t *kn = reinterpret_cast<bf16_t *>(part + W * G * D);      // [D] rotated new key (bf16)
    bf16_t *vn = kn + D;                          // [D] new value
    float *sc = reinterpret_cast<float *>(vn + D);                           // [G][maxlen] scores -> probabilities
    __shared__ float red[2 * G * W];
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long p = *pos_ptr;
    const bool pvalid = p >= 0 && p < maxlen;
    const bf16_t *row = qkv + b * ld_qkv;
    const bf16_t *c = cs + b * cs0, *sv = sn + b * cs0;
    bf16_t *Kb = K + ((int64_t)b * nkv + kvh) * maxlen * D;
    bf16_t *Vb = V + ((int64_t)b * nkv + kvh) * maxlen * D;
    const unsigned char *mrow = mask + b * ms0;
    // the first 256 keys (and the mask byte of key `tid`) are requested before the rotary arithmetic; the values of that tile are requested
    // right behind the score pass, into the registers the keys leave free -- <= 128 VGPRs for groups of up to four heads, i.e. two
    // workgroups per CU: at 64 sequences x 8 KV heads all 512 workgroups are resident at once (one round instead of two)
    const int cc = lane % LPR, r = lane / LPR;
    // cache slots behind the position being written hold no keys yet (every mask this kernel is handed hides them: causal decode), so the
    // first tile is only requested up to it: a quarter of the key / value bytes at the average length of a 128 + 128 token generation
    const int jlim = pvalid ? (int)min((long long)maxlen, p + 1) : maxlen;
    au32x4 k0[NI], v0[NI];
    const unsigned char mk0 = tid < maxlen ? mrow[tid] : (unsigned char)0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = key_of(0, wave, i, lane);
        k0[i] = j < jlim ? *reinterpret_cast<const au32x4 *>(Kb + (int64_t)j * D + cc * 8) : (au32x4)(0);
    }
    for (int role = wave; role < G + 2; role += W) {        // roles 0..G-1: query heads; G: the new key; G+1: the new value
        if (role < G) {
            if (lane < half) {
                const bf16_t *src = row + (kvh * G + role) * D;
                const float x1 = bf16_to_f32(src[lane]), x2 = bf16_to_f32(src[lane + half]);
                const float c1 = bf16_to_f32(c[lane]), c2 = bf16_to_f32(c[lane + half]);
                const float s1 = bf16_to_f32(sv[lane]), s2 = bf16_to_f32(sv[lane + half]);
                qs[role * D + lane] = bfr2(bfr2(x1 * c1) + bfr2(-x2 * s1));
                qs[role * D + lane + half] = bfr2(bfr2(x2 * c2) + bfr2(x1 * s2));
            }
        } else if (role == G) {
            if (lane < half) {
                const bf16_t *src = row + (nh + kvh) * D;
                const float x1 = bf16_to_f32(src[lane]), x2 = bf16_to_f32(src[lane + half]);
                const float c1 = bf16_to_f32(c[lane]), c2 = bf16_to_f32(c[lane + half]);
                const float s1 = bf16_to_f32(sv[lane]), s2 = bf16_to_f32(sv[lane + half]);
                const bf16_t k1 = f32_to_bf16(bfr2(x1 * c1) + bfr2(-x2 * s1)), k2 = f32_to_bf16(bfr2(x2 * c2) + bfr2(x1 * s2));
                kn[lane] = k1;
                kn[lane + half] = k2;
                if (pvalid) {
                    Kb[p * D + lane] = k1;
                    Kb[p * D + lane + half] = k2;
                }
            }
        } else {
            if (lane < half) {
                const bf16_t *src = row + (nh + nkv + kvh) * D;
                const uint32_t v2 = *reinterpret_cast<const uint32_t *>(src + lane * 2);
                *reinterpret_cast<uint32_t *>(vn + lane * 2) = v2;
                if (pvalid) *reinterpret_cast<uint32_t *>(Vb + p * D + lane * 2) = v2;
            }
        }
    }
    __syncthreads();
    const long long pn = pvalid ? p : -1;
    float q[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) q[g][e] = qs[g * D + cc * 8 + e];
    // ---- scores: the prefetched tile first (its registers are free afterwards), then any further tile of a longer context
    auto score_tile = [&](int j0, au32x4 (&kk)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = key_of(j0, wave, i, lane);
            if (j == pn) kk[i] = *reinterpret_cast<const au32x4 *>(kn + cc * 8);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float d = attn_row_sum<LPR>(attn_dot8(q[g], kk[i]));
                if (cc == 0 && j < maxlen) sc[g * maxlen + j] = d * scale;
            }
        }
    };
    score_tile(0, k0);
    for (int j0 = 256; j0 < maxlen; j0 += 256) {
        au32x4 kk[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = key_of(j0, wave, i, lane);
            kk[i] = (j < maxlen && mrow[j] != 0) ? *reinterpret_cast<const au32x4 *>(Kb + (int64_t)j * D + cc * 8) : (au32x4)(0);
        }
        score_tile(j0, kk);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {          // the values of the first tile, into the registers the keys and queries have left: in flight under the softmax
        const int j = key_of(0, wave, i, lane);
        v0[i] = j < jlim ? *reinterpret_cast<const au32x4 *>(Vb + (int64_t)j * D + cc * 8) : (au32x4)(0);
    }
    __syncthreads();
    // ---- softmax statistics per head over the unmasked keys
    float mx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) mx[g] = -INFINITY;
    for (int j = tid; j < maxlen; j += THREADS) {
        const bool ok = (j == tid ? mk0 : mrow[j]) != 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float s = ok ? sc[g * maxlen + j] : -INFINITY;
            sc[g * maxlen + j] = s;
            mx[g] = fmaxf(mx[g], s);
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float m = wave_max(mx[g]);
        if (lane == 0) red[g * W + wave] = m;
    }
    __syncthreads();
    float sum[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float m = red[g * W];
#pragma unroll
        for (int w = 1; w < W; ++w) m = fmaxf(m, red[g * W + w]);
        mx[g] = m;
        sum[g] = 0.f;
    }
    for (int j = tid; j < maxlen; j += THREADS) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float s = sc[g * maxlen + j];
            const float e = (s == -INFINITY) ? 0.f : expf(s - mx[g]);
            sc[g * maxlen + j] = e;
            sum[g] += e;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float t = wave_sum(sum[g]);
        if (lane == 0) red[(G + g) * W + wave] = t;
    }
    __syncthreads();
    // ---- out = P V: lane (r, c) accumulates columns 8c..8c+7 of every head over the rows it sees
    float acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    auto pv_tile = [&](int j0, au32x4 (&vt)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = key_of(j0, wave, i, lane);
            const bool live = j < maxlen && mrow[j < maxlen ? j : 0] != 0;
            au32x4 vv = vt[i];
            if (j == pn) vv = *reinterpret_cast<const au32x4 *>(vn + cc * 8);
            if (!live) vv = (au32x4)(0);          // masked rows may hold anything (NaN from padded positions)
#pragma unroll
            for (int g = 0; g < G; ++g) attn_axpy8(acc[g], live ? sc[g * maxlen + j] : 0.f, vv);
        }
    };
    pv_tile(0, v0);
    for (int j0 = 256; j0 < maxlen; j0 += 256) {
        au32x4 vt[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = key_of(j0, wave, i, lane);
            vt[i] = (j < maxlen && mrow[j] != 0) ? *reinterpret_cast<const au32x4 *>(Vb + (int64_t)j * D + cc * 8) : (au32x4)(0);
        }
        pv_tile(j0, vt);
    }
    // rows of one wave instruction -> one partial per wave (lanes c, c + LPR, ...), then across waves through LDS
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[g][e];
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            acc[g][e] = v;
        }
    if (r == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) part[(wave * G + g) * D + cc * 8 + e] = acc[g][e];
    }
    __syncthreads();
    for (int i = tid; i < G * D; i += THREADS) {
        const int g = i / D, d = i % D;
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) o += part[(w * G + g) * D + d];
        float den = red[(G + g) * W];
#pragma unroll
        for (int w = 1; w < W; ++w) den += red[(G + g) * W + w];
        const float inv = den > 0.f ? 1.f / den : 0.f;      // a fully masked query row (left padding) yields zeros, not NaN
        out[((int64_t)b * nh + kvh * G + g) * D + d] = f32_to_bf16(o * inv);
    }
}

// Per-token prologue of the decode step: what Qwen2RotaryEmbedding.forward (freqs = inv_freq * position, cos/sin in f32,
// * attention_scaling, cast to bf16) and create_causal_mask (key j visible iff j <= query position and not padding) compute
// with ~12 small ATen launches, as one.  grid B.
__global__ __launch_bounds__(256) void decode_prologue_kernel(const long long *__restrict__ posid, const float *__restrict__ inv_freq,
                                                              float scaling, const long long *__restrict__ mask2d, int64_t ms,
                                                              const long long *__restrict__ pos_ptr, bf16_t *__restrict__ cos_o,
                                                              bf16_t *__restrict__ sin_o, unsigned char *__restrict__ mask_o,
                                                              int D, int maxlen, int S) {
    // row r = b*S + s: the s-th new position of sequence b (S = 1: a decode step)
    const int r = blockIdx.x, tid = threadIdx.x;
    const int b = r / S, s = r - b * S;
    const int half = D / 2;
    if (tid < half) {
        const float f = inv_freq[tid] * (float)posid[r];
        const bf16_t c = f32_to_bf16(cosf(f) * scaling), sn = f32_to_bf16(sinf(f) * scaling);
        cos_o[(int64_t)r * D + tid] = c;
        cos_o[(int64_t)r * D + tid + half] = c;
        sin_o[(int64_t)r * D + tid] = sn;
        sin_o[(int64_t)r * D + tid + half] = sn;
    }
    const long long p = *pos_ptr + s;
    for (int j = tid; j < maxlen; j += 256) mask_o[(int64_t)r * maxlen + j] = (j <= p && mask2d[b * ms + j] != 0) ? 1 : 0;
}

static int g_gemv_nt = 1;   // non-temporal weight loads (tools/gemv_fused_sweep.py)
static int g_gemv_stage = 1;   // one-row GEMVs without RMSNorm stage x in LDS (gemv_stage_kernel)

template <int MROWS, bool NORM, int EPI>
static void launch_fused(bool nt, dim3 grid, size_t lds, hipStream_t s, const bf16_t *X, int ldx, const bf16_t *W, int ldw,
                         const float *bias, const bf16_t *normw, float eps, const bf16_t *res, int ldr, bf16_t *C, int ldc, int N,
                         int K) {
#define LL_GF(NT_, XC_) hipLaunchKernelGGL((gemv_fused_kernel<MROWS, NORM, EPI, NT_, XC_>), grid, dim3(256), lds, s, X, ldx, W, ldw, bias, \
                                           normw, eps, res, ldr, C, ldc, N, K)
    if (!NORM || K <= 4096) {
        if (nt) LL_GF(true, 2); else LL_GF(false, 2);
    } else {
        if (nt) LL_GF(true, 4); else LL_GF(false, 4);
    }
#undef LL_GF
}

template <int MROWS>
static int dispatch_fused(bool nt, hipStream_t s, const bf16_t *X, int ldx, const bf16_t *W, int ldw, const float *bias,
                          const bf16_t *normw, float eps, const bf16_t *res, int ldr, bf16_t *C, int ldc, int N, int K, int epi) {
    const bool norm = normw != nullptr;
    const size_t lds = norm ? (size_t)MROWS * K * 2 : 0;
    const dim3 grid(epi == GEMV_SILU_MUL ? cdiv(N, 4) : cdiv(N, 8));
#define LL_FUSED_CASE(NORM_, EPI_)                                                                                          \
    launch_fused<MROWS, NORM_, EPI_>(nt, grid, lds, s, X, ldx, W, ldw, bias, normw, eps, res, ldr, C, ldc, N, K)
    if (norm) {
        if (epi == GEMV_PLAIN) LL_FUSED_CASE(true, GEMV_PLAIN);
        else if (epi == GEMV_RESIDUAL) LL_FUSED_CASE(true, GEMV_RESIDUAL);
        else LL_FUSED_CASE(true, GEMV_SILU_MUL);
    } else {
        if (epi == GEMV_PLAIN) LL_FUSED_CASE(false, GEMV_PLAIN);
        else if (epi == GEMV_RESIDUAL) LL_FUSED_CASE(false, GEMV_RESIDUAL);
        else LL_FUSED_CASE(false, GEMV_SILU_MUL);
    }
#undef LL_FUSED_CASE
    LL_LAUNCH_CHECK();
    return LL_OK;
}

static int gemv_fused(bool nt, const void *x, int ldx, const void *W, int ldw, const float *bias, const void *norm_w, float eps,
                      const void *residual, int ldr, void *out, int ldc, int M, int N, int K, int epi, hipStream_t s) {
    LL_CHECK(x && W && out, "ll_gemv_fused_bf16: null argument");
    LL_CHECK(M >= 1 && M <= 4, "ll_gemv_fused_bf16: M=%d rows (decode shapes only, 1..4)", M);
    LL_CHECK(N >= 1 && K >= 8 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "ll_gemv_fused_bf16: K, ldx, ldw must be multiples of 8");
    LL_CHECK(epi >= GEMV_PLAIN && epi <= GEMV_SILU_MUL, "ll_gemv_fused_bf16: epilogue %d", epi);
    LL_CHECK(epi != GEMV_RESIDUAL || residual, "ll_gemv_fused_bf16: residual epilogue without a residual");
    LL_CHECK(!norm_w || (K <= 8192 && (size_t)M * K * 2 <= 65536), "ll_gemv_fused_bf16: RMSNorm prologue needs K <= 8192 and M*K <= 32768");
    const bf16_t *X = (const bf16_t *)x, *Wt = (const bf16_t *)W, *nw = (const bf16_t *)norm_w, *rs = (const bf16_t *)residual;
    bf16_t *C = (bf16_t *)out;
    if (g_gemv_stage && nt && M == 1 && !nw && epi != GEMV_SILU_MUL && K >= 8192 && K <= 20480) {
        // one row, no RMSNorm, long K (down_proj): x staged in LDS once per workgroup (down_proj 25.0 -> 23.8 us; for the short K of
        // o_proj the extra barrier costs more than the eight x loads per block it removes: 6.4 -> 7.0 us)
        const dim3 grid(cdiv(N, 8));
        if (epi == GEMV_RESIDUAL)
            hipLaunchKernelGGL((gemv_stage_kernel<GEMV_RESIDUAL>), grid, dim3(256), (size_t)K * 2, s, X, Wt, ldw, bias, rs, C, N, K);
        else
            hipLaunchKernelGGL((gemv_stage_kernel<GEMV_PLAIN>), grid, dim3(256), (size_t)K * 2, s, X, Wt, ldw, bias, rs, C, N, K);
        LL_LAUNCH_CHECK();
        return LL_OK;
    }
    switch (M) {
        case 1: return dispatch_fused<1>(nt, s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, C, ldc, N, K, epi);
        case 2: return dispatch_fused<2>(nt, s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, C, ldc, N, K, epi);
        case 3: return dispatch_fused<3>(nt, s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, C, ldc, N, K, epi);
        default: return dispatch_fused<4>(nt, s, X, ldx, Wt, ldw, bias, nw, eps, rs, ldr, C, ldc, N, K, epi);
    }
}

}  // namespace ll

using namespace ll;

extern "C" {

int ll_gemv_fused_bf16(const void *x, int ldx, const void *W, int ldw, const float *bias, const void *norm_w, float eps,
                       const void *residual, int ldr, void *out, int ldc, int M, int N, int K, int epi, void *stream) {
    return gemv_fused(g_gemv_nt != 0, x, ldx, W, ldw, bias, norm_w, eps, residual, ldr, out, ldc, M, N, K, epi, (hipStream_t)stream);
}

int ll_decode_attn_rope_bf16(const void *qkv, int64_t ld_qkv, const void *cos, const void *sin, int64_t cs_stride, void *Kc,
                             void *Vc, const int64_t *pos, const void *mask, int64_t mask_stride, void *out, int B, int nh,
                             int nkv, int maxlen, int D, float scale, void *stream) {
    LL_CHECK(qkv && cos && sin && Kc && Vc && pos && mask && out, "ll_decode_attn_rope_bf16: null argument");
    LL_CHECK((D == 64 || D == 128) && B >= 1 && nkv >= 1 && nh % nkv == 0 && maxlen >= 1 && maxlen <= 16384 && ld_qkv % 8 == 0,
             "ll_decode_attn_rope_bf16: unsupported shape");
    // many sequences: one workgroup per (KV head, sequence) for the group sizes of the supported models (Llama-3.1-8B / Mistral-7B 4,
    // Qwen2-7B 7, the toy models 2); up to 16 sequences keep the per-head kernel (bit-identical to the op-by-op path)
    const int G = nh / nkv;
    if (B > 16 && (G == 2 || G == 4 || G == 7)) {
        const int gw = G <= 4 ? 16 : 8;
        const size_t l3 = (size_t)(G * D + gw * G * D) * 4 + (size_t)2 * D * 2 + (size_t)G * maxlen * 4;
        if (l3 <= 150 * 1024) {
            dim3 g3(nkv, B);
#define LL_GQA(D_, G_, W_)                                                                                                               \
    do {                                                                                                                                 \
        static size_t attr = 0;                                                                                                          \
        if (l3 > attr) {                                                                                                                 \
            LL_HIP(hipFuncSetAttribute((const void *)decode_attn_rope_gqa_kernel<D_, G_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3)); \
            attr = l3;                                                                                                                   \
        }                                                                                                                                \
        hipLaunchKernelGGL((decode_attn_rope_gqa_kernel<D_, G_, W_>), g3, dim3(64 * W_), l3, (hipStream_t)stream, (const bf16_t *)qkv, ld_qkv, \
                           (const bf16_t *)cos, (const bf16_t *)sin, cs_stride, (bf16_t *)Kc, (bf16_t *)Vc, (const long long *)pos,      \
                           (const unsigned char *)mask, mask_stride, (bf16_t *)out, nh, nkv, maxlen, scale);                            \
    } while (0)
            // sixteen waves for groups of up to four heads (16 keys per wave and tile: <= 128 VGPRs), eight for seven (Qwen2-7B)
            if (D == 128) {
                if (G == 2) LL_GQA(128, 2, 16); else if (G == 4) LL_GQA(128, 4, 16); else LL_GQA(128, 7, 8);
            } else {
                if (G == 2) LL_GQA(64, 2, 16); else if (G == 4) LL_GQA(64, 4, 16); else LL_GQA(64, 7, 8);
            }
#undef LL_GQA
            LL_LAUNCH_CHECK();
            return LL_OK;
        }
    }
    const size_t lds = (size_t)(D + ATTN_PART_FLOATS) * 4 + (size_t)2 * D * 2 + (size_t)maxlen * 4;
    dim3 grid(nh, B);
    if (D == 128)
        hipLaunchKernelGGL((decode_attn_rope_kernel<128>), grid, dim3(ATTN_THREADS), lds, (hipStream_t)stream, (const bf16_t *)qkv, ld_qkv,
                           (const bf16_t *)cos, (const bf16_t *)sin, cs_stride, (bf16_t *)Kc, (bf16_t *)Vc, (const long long *)pos,
                           (const unsigned char *)mask, mask_stride, (bf16_t *)out, nh, nkv, maxlen, scale);
    else
        hipLaunchKernelGGL((decode_attn_rope_kernel<64>), grid, dim3(ATTN_THREADS), lds, (hipStream_t)stream, (const bf16_t *)qkv, ld_qkv,
                           (const bf16_t *)cos, (const bf16_t *)sin, cs_stride, (bf16_t *)Kc, (bf16_t *)Vc, (const long long *)pos,
                           (const unsigned char *)mask, mask_stride, (bf16_t *)out, nh, nkv, maxlen, scale);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_decode_prologue(const int64_t *position_ids, const float *inv_freq, float attention_scaling, const int64_t *mask2d,
                       int64_t mask_stride, const int64_t *pos, void *cos, void *sin, void *mask_out, int B, int D, int maxlen,
                       void *stream) {
    LL_CHECK(position_ids && inv_freq && mask2d && pos && cos && sin && mask_out, "ll_decode_prologue: null argument");
    LL_CHECK(B >= 1 && D >= 2 && D % 2 == 0 && D <= 512 && maxlen >= 1, "ll_decode_prologue: unsupported shape");
    hipLaunchKernelGGL(decode_prologue_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long *)position_ids, inv_freq,
                       attention_scaling, (const long long *)mask2d, mask_stride, (const long long *)pos, (bf16_t *)cos, (bf16_t *)sin,
                       (unsigned char *)mask_out, D, maxlen, 1);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_suffix_prologue(const int64_t *position_ids, const float *inv_freq, float attention_scaling, const int64_t *mask2d,
                       int64_t mask_stride, const int64_t *pos, void *cos, void *sin, void *mask_out, int B, int S, int D, int maxlen,
                       void *stream) {
    LL_CHECK(position_ids && inv_freq && mask2d && pos && cos && sin && mask_out, "ll_suffix_prologue: null argument");
    LL_CHECK(B >= 1 && S >= 1 && D >= 2 && D % 2 == 0 && D <= 512 && maxlen >= 1, "ll_suffix_prologue: unsupported shape");
    hipLaunchKernelGGL(decode_prologue_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, (const long long *)position_ids, inv_freq,
                       attention_scaling, (const long long *)mask2d, mask_stride, (const long long *)pos, (bf16_t *)cos, (bf16_t *)sin,
                       (unsigned char *)mask_out, D, maxlen, S);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_suffix_attn_rope_bf16(const void *qkv, int64_t ld_qkv, const void *cos, const void *sin, void *Kc, void *Vc, const int64_t *pos,
                             const void *mask, void *out, int B, int S, int nh, int nkv, int maxlen, int D, float scale, void *stream) {
    LL_CHECK(qkv && cos && sin && Kc && Vc && pos && mask && out, "ll_suffix_attn_rope_bf16: null argument");
    LL_CHECK((D == 64 || D == 128) && B >= 1 && S >= 1 && S <= 16 && nkv >= 1 && nh % nkv == 0 && maxlen >= 1 && maxlen <= 16384 &&
                 ld_qkv % 8 == 0,
             "ll_suffix_attn_rope_bf16: unsupported shape");
    const size_t lds = ((size_t)maxlen + D + ATTN_PART_FLOATS) * 4;
    dim3 grid(nh, B * S);
    if (D == 128)
        hipLaunchKernelGGL((suffix_attn_rope_kernel<128>), grid, dim3(ATTN_THREADS), lds, (hipStream_t)stream, (const bf16_t *)qkv, ld_qkv,
                           (const bf16_t *)cos, (const bf16_t *)sin, (bf16_t *)Kc, (bf16_t *)Vc, (const long long *)pos,
                           (const unsigned char *)mask, (bf16_t *)out, nh, nkv, S, maxlen, scale);
    else
        hipLaunchKernelGGL((suffix_attn_rope_kernel<64>), grid, dim3(ATTN_THREADS), lds, (hipStream_t)stream, (const bf16_t *)qkv, ld_qkv,
                           (const bf16_t *)cos, (const bf16_t *)sin, (bf16_t *)Kc, (bf16_t *)Vc, (const long long *)pos,
                           (const unsigned char *)mask, (bf16_t *)out, nh, nkv, S, maxlen, scale);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

// Times ll_gemv_fused_bf16 on synthetic operands, cycling through `nweights` distinct weight matrices (defeats the
// 256 MiB Infinity Cache).  norm != 0 adds the RMSNorm prologue; nt selects non-temporal weight loads.
#if LL_TUNING
int ll_gemv_fused_bench(int M, int N, int K, int epi, int norm, int nt, int iters, int nweights, float *ms) {
    LL_CHECK(ms && iters > 0 && nweights > 0 && M >= 1 && M <= 4, "bad argument");
    const int rowsW = epi == GEMV_SILU_MUL ? 2 * N : N;
    bf16_t *X = nullptr, *W = nullptr, *C = nullptr, *R = nullptr, *NW = nullptr;
    LL_HIP(hipMalloc(&X, (size_t)4 * K * 2));
    LL_HIP(hipMalloc(&NW, (size_t)K * 2));
    LL_HIP(hipMalloc(&W, (size_t)nweights * rowsW * K * 2));
    LL_HIP(hipMalloc(&C, (size_t)4 * N * 2));
    LL_HIP(hipMalloc(&R, (size_t)4 * N * 2));
    LL_HIP(hipMemset(X, 0x11, (size_t)4 * K * 2));
    LL_HIP(hipMemset(NW, 0x11, (size_t)K * 2));
    LL_HIP(hipMemset(R, 0x11, (size_t)4 * N * 2));
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
            rc = gemv_fused(nt != 0, X, K, W + (size_t)(i % nweights) * rowsW * K, K, nullptr, norm ? NW : nullptr, 1e-6f, R, N, C, N, M,
                            N, K, epi, st);
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
    (void)hipFree(NW);
    (void)hipFree(W);
    (void)hipFree(C);
    (void)hipFree(R);
    if (rc != LL_OK) return rc;
    LL_HIP(he);
    return LL_OK;
}
#endif

#if LL_TUNING
int ll_set_gemv_stage(int on) {
    const int old = g_gemv_stage;
    g_gemv_stage = on ? 1 : 0;
    return old;
}
#endif

#if LL_TUNING
int ll_set_gemv_nt(int on) {
    const int old = g_gemv_nt;
    g_gemv_nt = on ? 1 : 0;
    return old;
}
#endif

}  // extern "C"
