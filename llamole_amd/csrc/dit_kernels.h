// Device kernels of the GraphDiT reverse-diffusion step (gfx950).  Included by graphdit.hip only.
//
// State representation (MI355X-first: the dense one-hot float tensors of the reference,
// X [B,N,16] / E [B,N,N,5], are never materialised): X int8 [B,N], E int8 [B,N,N];
// -1 encodes the all-zero one-hot vector (masked node / masked pair / the z_T diagonal).
#pragma once
#include <type_traits>

#include "common.h"

namespace ll {

// ---------------------------------------------------------------------------------------------- on-device sampling noise
// ONE definition of which Philox4x32-10 counter feeds which Exp(1) variate of a reverse step (s in [0, T)) or of z_T (s = T), used by the
// posterior kernels, init_state_kernel and the test probe (ll_dit_noise_probe) alike.  key = (seed lo, seed hi).
//   atom noise  q[b, i, c],      c in [0,16): counter (b * N + i,           s, c >> 2, 0x58), word c & 3
//   bond noise  q[b, i, j, k],   k in [0, 5): counter ((b * N + i) * N + j, s, k >> 2, 0x45), word k & 3   (pairs j > i only)
// The last counter word separates the two families, the second the steps, the key the batches: no (family, step, element) pair shares a
// counter (tests/test_noise_gpu.py checks the dumps pairwise).
__device__ __forceinline__ uint4 dit_noise_x4(uint2 key, int s, int node, int g) {
    return philox4x32(make_uint4((uint32_t)node, (uint32_t)s, (uint32_t)g, 0x58u), key);
}
__device__ __forceinline__ uint4 dit_noise_e4(uint2 key, int s, int pair, int g) {
    return philox4x32(make_uint4((uint32_t)pair, (uint32_t)s, (uint32_t)g, 0x45u), key);
}


constexpr int XD = LL_XDIM;
constexpr int ED = LL_EDIM;

__device__ __forceinline__ float block_sum_256(float v, float *red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename T> __device__ __forceinline__ void store4(T *p, float4 o);
template <> __device__ __forceinline__ void store4<float>(float *p, float4 o) { *reinterpret_cast<float4 *>(p) = o; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t *p, float4 o) {
    uint2 u;
    u.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
    u.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
    *reinterpret_cast<uint2 *>(p) = u;
}

// ------------------------------------------------------------------------------------------ x_embedder
// h = LayerNorm_affine(W_x . [onehot(x_i) | onehot(e_i0) .. onehot(e_i,N-1)])   (transformer.py:41-44, 95-96)
// The input row has at most N+1 non-zeros, so the Linear is a gather-sum of rows of W_x^T [F,H].
// One workgroup per (graph, node); the result is identical for the conditional and the
// unconditional pass and is written to both halves of the residual stream.
template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const int8_t *__restrict__ X, const int8_t *__restrict__ E,
                                                     const float *__restrict__ WxT, const float *__restrict__ lnw,
                                                     const float *__restrict__ lnb, float *__restrict__ x32,
                                                     T *__restrict__ xa, const int *__restrict__ step_ptr, int B,
                                                     int N, int H, int Ht) {
    // H = row pitch (the engine's padded width, a multiple of 64), Ht = the checkpoint's hidden_size: columns [Ht, H) of W_x^T and of
    // the LayerNorm weight / bias are zero, the statistics run over the Ht true columns, the padded columns come out as exact zeros
    __shared__ float red[4];
    __shared__ int gidx[136];               // the atom class + up to 128 bond partners
    __shared__ int ng;
    const int row = blockIdx.x;  // b*N + i
    const int b = row / N;
    const int i = row - b * N;
    if (threadIdx.x < 64) {  // wave 0 compacts the non-zero input columns with a ballot (order-preserving), 64 partners per pass
        const int lane = threadIdx.x;
        const int half = (*step_ptr + 1) & 1;   // z_{s+1} lives in half (s+1)&1 of the double-buffered state
        const int8_t *er = E + (((int64_t)half * B + b) * N + i) * N;
        const int xi = X[(int64_t)half * B * N + row];
        int base = (xi >= 0) ? 1 : 0;
        if (lane == 0 && xi >= 0) gidx[0] = xi;
        for (int j0 = 0; j0 < N; j0 += 64) {
            const int j = j0 + lane;
            const int e = (j < N) ? (int)er[j] : -1;
            const unsigned long long m = __ballot(e >= 0);
            if (e >= 0) gidx[base + __popcll(m & ((1ull << lane) - 1ull))] = XD + ED * j + e;
            base += __popcll(m);
        }
        if (lane == 0) ng = base;
    }
    __syncthreads();
    constexpr int MAXE = 2;  // float4 chunks per thread: H <= 2048
    float4 v[MAXE];
    const int n = ng;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = (threadIdx.x + e * 256) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h < H) {
            for (int g0 = 0; g0 < n; g0 += 8) {  // 8 independent 16-B gathers in flight per thread
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int g = g0 + u;
                    t[u] = (g < n) ? *reinterpret_cast<const float4 *>(WxT + (int64_t)gidx[g] * H + h)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w;
                }
            }
        }
        v[e] = s;
    }
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) ls += v[e].x + v[e].y + v[e].z + v[e].w;  // zero beyond Ht
    const float mean = block_sum_256(ls, red) / (float)Ht;
    float lv = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) lv += sq_dev4(v[e], mean, (threadIdx.x + e * 256) * 4, Ht);
    const float rstd = rsqrtf(block_sum_256(lv, red) / (float)Ht + 1e-5f);
    const int64_t M = (int64_t)B * N;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = (threadIdx.x + e * 256) * 4;
        if (h < H) {
            const float4 w = *reinterpret_cast<const float4 *>(lnw + h);
            const float4 bb = *reinterpret_cast<const float4 *>(lnb + h);
            float4 o;
            o.x = (v[e].x - mean) * rstd * w.x + bb.x;
            o.y = (v[e].y - mean) * rstd * w.y + bb.y;
            o.z = (v[e].z - mean) * rstd * w.z + bb.z;
            o.w = (v[e].w - mean) * rstd * w.w + bb.w;
            *reinterpret_cast<float4 *>(x32 + (int64_t)row * H + h) = o;
            *reinterpret_cast<float4 *>(x32 + (M + row) * H + h) = o;
            store4<T>(xa + (int64_t)row * H + h, o);
            store4<T>(xa + (M + row) * H + h, o);
        }
    }
}

// ------------------------------------------------------------------------------------------ attention (generic)
// Per (sequence, head): LayerNorm(hd, affine) on q and k rows, key mask valid_i & valid_j with padded
// query rows opened to all keys, softmax(q k^T / sqrt(hd)) v     (layers.py:56-87).
// Generic f32 LDS kernel for any N <= 128, hd <= 128; the MFMA variant below covers the bf16 engine.  K and V of the sequence stay in LDS,
// the queries go through in chunks of QC rows (QC = N whenever that fits: one chunk, the arithmetic of every element is the same either way).
template <typename T>
__global__ __launch_bounds__(256) void attn_generic_kernel(const T *__restrict__ qkv, T *__restrict__ o,
                                                            const float *__restrict__ qw, const float *__restrict__ qb,
                                                            const float *__restrict__ kw, const float *__restrict__ kb,
                                                            const int *__restrict__ n_nodes, int B, int N, int H,
                                                            int hd, int hdp, int QC) {
    // H = width of each of the q | k | v sections of a qkv row (and of an output row), hdp = column pitch of a head inside a section,
    // hd = the checkpoint's head dimension (hd <= hdp; columns [hd, hdp) of a head are the engine's zero padding and are not touched)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int head = blockIdx.x;
    const int seq = blockIdx.y;  // pass*B + b
    const int b = seq % B;
    const int nv = n_nodes[b];
    const int ld = hd + 1;
    float *k = sm;
    float *v = k + N * ld;
    float *q = v + N * ld;   // [QC][ld]
    float *S = q + QC * ld;  // [QC][N+1]
    const int tid = threadIdx.x;
    const int64_t rbase = (int64_t)seq * N;
    auto layer_norm_row = [&](float *r, const float *w, const float *bb) {
        float m = 0.f;
        for (int d = 0; d < hd; ++d) m += r[d];
        m /= (float)hd;
        float var = 0.f;
        for (int d = 0; d < hd; ++d) {
            const float t = r[d] - m;
            var += t * t;
        }
        const float rstd = rsqrtf(var / (float)hd + 1e-5f);
        for (int d = 0; d < hd; ++d) r[d] = (r[d] - m) * rstd * w[d] + bb[d];
    };
    for (int idx = tid; idx < N * hd; idx += 256) {
        const int i = idx / hd, d = idx - i * hd;
        const T *r = qkv + (rbase + i) * (3 * (int64_t)H) + head * hdp + d;
        k[i * ld + d] = to_f32<T>(r[H]);
        v[i * ld + d] = to_f32<T>(r[2 * H]);
    }
    __syncthreads();
    if (tid < N) layer_norm_row(k + tid * ld, kw, kb);      // LayerNorm over hd, one thread per key row
    const float scale = rsqrtf((float)hd);
    for (int i0 = 0; i0 < N; i0 += QC) {
        const int nq = min(QC, N - i0);
        __syncthreads();                                    // the previous chunk is done with q and S (and k is normalised)
        for (int idx = tid; idx < nq * hd; idx += 256) {
            const int i = idx / hd, d = idx - i * hd;
            q[i * ld + d] = to_f32<T>(qkv[(rbase + i0 + i) * (3 * (int64_t)H) + head * hdp + d]);
        }
        __syncthreads();
        if (tid < nq) layer_norm_row(q + tid * ld, qw, qb);
        __syncthreads();
        for (int idx = tid; idx < nq * N; idx += 256) {
            const int i = idx / N, j = idx - i * N;
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s = fmaf(q[i * ld + d], k[j * ld + d], s);
            const bool allow = (i0 + i >= nv) || (j < nv);  // padded query rows attend to every key
            S[i * (N + 1) + j] = allow ? s * scale : -INFINITY;
        }
        __syncthreads();
        if (tid < nq) {
            float *r = S + tid * (N + 1);
            float mx = -INFINITY;
            for (int j = 0; j < N; ++j) mx = fmaxf(mx, r[j]);
            float sum = 0.f;
            for (int j = 0; j < N; ++j) {
                const float e = expf(r[j] - mx);
                r[j] = e;
                sum += e;
            }
            const float inv = 1.f / sum;
            for (int j = 0; j < N; ++j) r[j] *= inv;
        }
        __syncthreads();
        for (int idx = tid; idx < nq * hd; idx += 256) {
            const int i = idx / hd, d = idx - i * hd;
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf(S[i * (N + 1) + j], v[j * ld + d], acc);
            o[(rbase + i0 + i) * (int64_t)H + head * hdp + d] = from_f32<T>(acc);
        }
    }
}

#ifdef LL_QA_PROBE      // tools/qkv_attn_probe.hip: cycle stamps of wave 0 of every workgroup
__device__ unsigned long long g_qa_stamps[4096 * 8];
__device__ unsigned long long g_qa_wstamps[1024 * 12 * 4];     // per wave: stamps 0..3
#define LL_QA_STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter();                                   \
        if (threadIdx.x == 0) g_qa_stamps[(blockIdx.x & 4095) * 8 + (i)] = t_;                                          \
        if ((i) < 4 && (threadIdx.x & 63) == 0) g_qa_wstamps[((blockIdx.x & 1023) * 12 + (threadIdx.x >> 6)) * 4 + (i)] = t_; } while (0)
#else
#define LL_QA_STAMP(i) do { } while (0)
#endif
// ------------------------------------------------------------------------------------------ attention (MFMA, bf16)
// One workgroup of one or two 64-lane waves per (sequence, head).  The whole graph (N <= 64 nodes) is one tile:
//   q,k rows -> f32 LayerNorm(hd) in registers -> bf16 in LDS;  V stored transposed in LDS;
//   S = Q K^T on v_mfma_f32_16x16x32_bf16, mask + softmax in the MFMA C layout (row reductions are
//   4-step xor-shuffles inside 16-lane groups), P -> bf16 via LDS, O = P V on MFMA.   (layers.py:56-87)
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(4))) float af32x4;

// S = Q K^T, mask + softmax, O = P V for the query tiles of wave `wave` (of WPB), operands in LDS: Qs / Ks [NP][HD + 8] bf16
// (LayerNorm applied), Vt [HD][NP + 8] (V transposed, zero columns for padded keys), Ps [NP][NP + 8] scratch for P.  PSEP =
// Ps is its own region; otherwise it aliases Qs and the WPB waves synchronise before P is written.  exp and the reciprocal of
// the row sum are the hardware v_exp_f32 / v_rcp_f32 (1 ulp; P is rounded to bf16 right after).  PV is evaluated as
// O^T = V^T P^T, so a lane ends up with four consecutive head columns of one query row: 8-byte stores.
struct AttnNoWait { __device__ __forceinline__ void operator()() const {} };
// `before_pv()` runs between the softmax and the first read of Vt (qkv_attn_kernel waits there for the waves that write V^T).
// PAIR (NP = 64): rows 0..31 and 32..63 of the images are the tokens of TWO sequences (nv / nv1 valid nodes; `ohead` = first row of the
// first one, the second follows N rows later): a query attends only to the keys of its own half.
template <int NP, int HD, int WPB, bool PSEP, typename BeforePV = AttnNoWait, bool PAIR = false>
__device__ __forceinline__ void attn_core(const bf16_t *Qs, const bf16_t *Ks, const bf16_t *Vt, bf16_t *Ps,
                                          bf16_t *__restrict__ ohead, int N, int nv, int H, int wave, int lane, float scale2,
                                          BeforePV before_pv = BeforePV(), int nv1 = 0) {
    static_assert(!PAIR || NP == 64, "two sequences per workgroup: 2 x 32 token rows");
    constexpr int QLD = HD + 8, PLD = NP + 8;
    constexpr int MT = NP / 16, KS = HD / 32;
    constexpr int MQ = MT / WPB;
    static_assert(MQ >= 1 && MQ * WPB == MT, "query tiles must divide among the waves");
    LL_QA_STAMP(5);
    const int q0 = wave * MQ;
    af32x4 acc[MQ][MT];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = af32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        abf16x8 fa[MQ], fb[MT];
#pragma unroll
        for (int i = 0; i < MQ; ++i) fa[i] = *reinterpret_cast<const abf16x8 *>(Qs + ((q0 + i) * 16 + fr) * QLD + ks * 32 + fk * 8);
#pragma unroll
        for (int i = 0; i < MT; ++i) fb[i] = *reinterpret_cast<const abf16x8 *>(Ks + (i * 16 + fr) * QLD + ks * 32 + fk * 8);
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (!PSEP) {
        if (WPB == 1) {
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // all Q reads done before P overwrites region 0
        } else {
            __syncthreads();
        }
    }

    // ---- mask + softmax in the C layout: element r of tile (mt,nt): i = mt*16 + (lane>>4)*4 + r, j = nt*16 + (lane&15)
    // scale2 = log2(e) / sqrt(hd): softmax(s / sqrt(hd)) through exp2, hd = the checkpoint's head dimension (HD is the padded tile)
#pragma unroll
    for (int mt = 0; mt < MQ; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = (q0 + mt) * 16 + fk * 4 + r;
            float mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < MT; ++nt) {
                const int j = nt * 16 + fr;
                bool allow;
                if (PAIR) {
                    const int il = i & 31, jl = j & 31, nvh = (i >> 5) ? nv1 : nv;
                    allow = ((i >> 5) == (j >> 5)) && (jl < N) && ((il >= nvh) || (jl < nvh));
                } else {
                    allow = (j < N) && ((i >= nv) || (j < nv));
                }
                const float sv = allow ? acc[mt][nt][r] * scale2 : -INFINITY;
                acc[mt][nt][r] = sv;
                mx = fmaxf(mx, sv);
            }
            mx = row16_max(mx);
            float sm = 0.f;
#pragma unroll
            for (int nt = 0; nt < MT; ++nt) {
                const float ev = __builtin_amdgcn_exp2f(acc[mt][nt][r] - mx);
                acc[mt][nt][r] = ev;
                sm += ev;
            }
            sm = row16_sum(sm);
            const float inv = __builtin_amdgcn_rcpf(sm);
#pragma unroll
            for (int nt = 0; nt < MT; ++nt) Ps[i * PLD + nt * 16 + fr] = f32_to_bf16(acc[mt][nt][r] * inv);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    LL_QA_STAMP(6);
    before_pv();

    // ---- O^T = V^T P^T (this wave's rows of P only: written and read by the same wave)
    constexpr int NT2 = HD / 16, KS2 = NP / 32;
    af32x4 oc[MQ][NT2];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < NT2; ++j) oc[i][j] = af32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
        abf16x8 fp[MQ], fv[NT2];
#pragma unroll
        for (int i = 0; i < MQ; ++i) fp[i] = *reinterpret_cast<const abf16x8 *>(Ps + ((q0 + i) * 16 + fr) * PLD + ks * 32 + fk * 8);
#pragma unroll
        for (int j = 0; j < NT2; ++j) fv[j] = *reinterpret_cast<const abf16x8 *>(Vt + (j * 16 + fr) * PLD + ks * 32 + fk * 8);
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < NT2; ++j) oc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[j], fp[i], oc[i][j], 0, 0, 0);
    }
    // oc[mt][nt][r] = O[query (q0 + mt) * 16 + fr][head column nt * 16 + fk * 4 + r]
#pragma unroll
    for (int mt = 0; mt < MQ; ++mt) {
        const int i = (q0 + mt) * 16 + fr;
        const int orow = PAIR ? (i >> 5) * N + (i & 31) : i;          // output row relative to `ohead`
        if (PAIR ? (i & 31) < N : i < N) {
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                const uint32_t lo = (uint32_t)f32_to_bf16(oc[mt][nt][0]) | ((uint32_t)f32_to_bf16(oc[mt][nt][1]) << 16);
                const uint32_t hi = (uint32_t)f32_to_bf16(oc[mt][nt][2]) | ((uint32_t)f32_to_bf16(oc[mt][nt][3]) << 16);
                *reinterpret_cast<uint2 *>(ohead + (int64_t)orow * H + nt * 16 + fk * 4) = make_uint2(lo, hi);
            }
        }
    }
}

// WPB waves per (sequence, head): with two, each wave loads / normalises half the rows (14 instead of 20 loads in flight per
// lane) and owns half the query rows of S, the softmax and O -- the serial MFMA / softmax chain per wave halves; K and V^T are
// shared through LDS (two workgroup barriers).  Row-wise arithmetic is unchanged, so the result is bit-identical to WPB = 1.
// `ld(row, which, d0)` returns the 8 bf16 at columns [d0, d0 + 8) of this head's q (which = 0), k (1) or v (2) row: from
// the qkv activation in global memory (attn_mfma_kernel) or from the LDS image the fused q|k|v GEMM left (qkv_attn_kernel);
// `ohead` = o + first row of the sequence * H + head * HD.  Waves >= WPB of a larger workgroup must not enter.
struct AttnBlockSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };
// `sync()` = barrier of the WPB waves that run the body (the whole workgroup in attn_mfma_kernel)
template <int NP, int HD, int WPB, typename LD, typename SYNC = AttnBlockSync>
__device__ __forceinline__ void attn_mfma_body(LD ld, bf16_t *__restrict__ ohead, const float *__restrict__ qw,
                                               const float *__restrict__ qb, const float *__restrict__ kw,
                                               const float *__restrict__ kb, int N, int nv, int H, int hd, unsigned char *smraw_attn,
                                               int wave, int lane, SYNC sync = SYNC()) {
    // HD = the head's column pitch (32 | 64 | 96 | 128), hd <= HD = the checkpoint's head dimension: columns [hd, HD) of q | k | v are the
    // engine's zero padding (zero weight rows, zero LayerNorm weight / bias) -- they take no part in the LayerNorm statistics and
    // contribute exact zeros to QK^T and to the output
    constexpr int QLD = HD + 8;           // padded row strides (elements)
    constexpr int PLD = NP + 8;
    constexpr int QK_ELEMS = NP * QLD;
    constexpr int P_ELEMS = NP * PLD;
    constexpr int R0 = QK_ELEMS > P_ELEMS ? QK_ELEMS : P_ELEMS;   // region 0: Q, later P
    static_assert(WPB == 1 || WPB == 2 || WPB == 4, "one, two or four waves per (sequence, head)");
    static_assert(HD % 32 == 0 && HD >= 32 && HD <= 128, "head pitch: a multiple of the MFMA k-step up to 128");
    constexpr int CW = WPB > NP / 16 ? NP / 16 : WPB;             // waves of the QK^T / softmax / PV part: at most one per query tile
    constexpr bool PSEP = CW != WPB;                              // then P gets its own region (no barrier inside the core)
    bf16_t *Qs = reinterpret_cast<bf16_t *>(smraw_attn);
    bf16_t *Ks = Qs + R0;
    bf16_t *Vt = Ks + QK_ELEMS;
    bf16_t *Ps = PSEP ? Vt + HD * PLD : Qs;

    // ---- load + LayerNorm (q, k), transpose (v).  LPR lanes cover one row with 16-B loads (full 128-B
    //      lines for hd = 64); every global load of the wave (3 tensors x PASSES + LN parameters) is issued
    //      before the first use, so the phase costs one memory round trip.
    {
        constexpr int LPRA = HD / 8;            // lanes per row that hold data
        constexpr int LPR = HD <= 32 ? 4 : HD <= 64 ? 8 : 16;   // lanes per row (a power of two: HD = 96 leaves four of sixteen idle)
        constexpr int RPP = 64 / LPR;           // rows per pass
        constexpr int PASSES = NP / RPP / WPB;   // passes of THIS wave: wave w takes rows [w * NP / WPB, (w + 1) * NP / WPB)
        static_assert(PASSES >= 1, "too few rows for this many waves");
        const int sub = lane % LPR, rin = lane / LPR + wave * (NP / WPB);
        const bool act = LPRA == LPR || sub < LPRA;
        const int d0 = sub * 8;
        uint4 rq[PASSES], rk[PASSES], rv[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = p * RPP + rin, srow = row < N ? row : 0;
            rq[p] = act ? ld(srow, 0, d0) : make_uint4(0u, 0u, 0u, 0u);
            rk[p] = act ? ld(srow, 1, d0) : make_uint4(0u, 0u, 0u, 0u);
            rv[p] = act ? ld(srow, 2, d0) : make_uint4(0u, 0u, 0u, 0u);
        }
        float wq[8], bq[8], wk[8], bk[8];
        {
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 a0 = act ? *reinterpret_cast<const float4 *>(qw + d0) : z4, a1 = act ? *reinterpret_cast<const float4 *>(qw + d0 + 4) : z4;
            const float4 b0 = act ? *reinterpret_cast<const float4 *>(qb + d0) : z4, b1 = act ? *reinterpret_cast<const float4 *>(qb + d0 + 4) : z4;
            const float4 c0 = act ? *reinterpret_cast<const float4 *>(kw + d0) : z4, c1 = act ? *reinterpret_cast<const float4 *>(kw + d0 + 4) : z4;
            const float4 e0 = act ? *reinterpret_cast<const float4 *>(kb + d0) : z4, e1 = act ? *reinterpret_cast<const float4 *>(kb + d0 + 4) : z4;
            wq[0] = a0.x; wq[1] = a0.y; wq[2] = a0.z; wq[3] = a0.w; wq[4] = a1.x; wq[5] = a1.y; wq[6] = a1.z; wq[7] = a1.w;
            bq[0] = b0.x; bq[1] = b0.y; bq[2] = b0.z; bq[3] = b0.w; bq[4] = b1.x; bq[5] = b1.y; bq[6] = b1.z; bq[7] = b1.w;
            wk[0] = c0.x; wk[1] = c0.y; wk[2] = c0.z; wk[3] = c0.w; wk[4] = c1.x; wk[5] = c1.y; wk[6] = c1.z; wk[7] = c1.w;
            bk[0] = e0.x; bk[1] = e0.y; bk[2] = e0.z; bk[3] = e0.w; bk[4] = e1.x; bk[5] = e1.y; bk[6] = e1.z; bk[7] = e1.w;
        }
        const float inv_hd = 1.f / (float)hd;
        auto norm_store = [&](const uint4 r, const float *w, const float *bvec, bf16_t *dst, int row) {
            const bool live = row < N;
            float f[8];
            const uint32_t u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f[2 * t] = __uint_as_float(u[t] << 16);
                f[2 * t + 1] = __uint_as_float(u[t] & 0xffff0000u);
            }
            float sm = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += f[e];
            sm = (LPR == 16) ? row16_sum(sm) : (LPR == 8) ? row8_sum(sm) : row4_sum(sm);
            const float mean = sm * inv_hd;
            float vr = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = f[e] - mean;
                vr += (d0 + e < hd) ? d * d : 0.f;
            }
            vr = (LPR == 16) ? row16_sum(vr) : (LPR == 8) ? row8_sum(vr) : row4_sum(vr);
            const float rstd = rsqrtf(vr * inv_hd + 1e-5f);
            uint32_t pk[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float a0 = live ? (f[2 * t] - mean) * rstd * w[2 * t] + bvec[2 * t] : 0.f;
                const float a1 = live ? (f[2 * t + 1] - mean) * rstd * w[2 * t + 1] + bvec[2 * t + 1] : 0.f;
                pk[t] = (uint32_t)f32_to_bf16(a0) | ((uint32_t)f32_to_bf16(a1) << 16);
            }
            if (act) *reinterpret_cast<uint4 *>(dst + row * QLD + d0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        };
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = p * RPP + rin;
            norm_store(rq[p], wq, bq, Qs, row);
            norm_store(rk[p], wk, bk, Ks, row);
            const bool live = row < N;
            const uint32_t u[4] = {rv[p].x, rv[p].y, rv[p].z, rv[p].w};
            if (act) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    Vt[(d0 + 2 * t) * PLD + row] = live ? (bf16_t)(u[t] & 0xffffu) : (bf16_t)0;
                    Vt[(d0 + 2 * t + 1) * PLD + row] = live ? (bf16_t)(u[t] >> 16) : (bf16_t)0;
                }
            }
        }
    }
    if (WPB == 1) {
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else {
        sync();
    }
    if (PSEP && wave >= CW) return;
    static_assert(PSEP || WPB == 1 || std::is_same<SYNC, AttnBlockSync>::value, "attn_core synchronises the whole workgroup when P aliases Q");
    attn_core<NP, HD, CW, PSEP>(Qs, Ks, Vt, Ps, ohead, N, nv, H, wave, lane, rsqrtf((float)hd) * 1.44269504088896340736f);
}

template <int NP, int HD, int WPB>
__global__ __launch_bounds__(64 * WPB) void attn_mfma_kernel(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ o,
                                                         const float *__restrict__ qw, const float *__restrict__ qb,
                                                         const float *__restrict__ kw, const float *__restrict__ kb,
                                                         const int *__restrict__ n_nodes, int B, int N, int H,
                                                         int heads, int hd) {
    // H = width of each of the q | k | v sections of a qkv row (heads * HD, padded to a multiple of 64), hd = the true head dimension
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw_attn[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int head = blockIdx.x;
    if (head >= heads) return;
    const int seq = blockIdx.y;
    const int nv = n_nodes[seq % B];
    const bf16_t *base = qkv + (int64_t)seq * N * (3 * (int64_t)H) + head * HD;
    attn_mfma_body<NP, HD, WPB>(
        [&](int row, int which, int d0) {
            return *reinterpret_cast<const uint4 *>(base + (int64_t)row * (3 * (int64_t)H) + (int64_t)which * H + d0);
        },
        o + (int64_t)seq * N * H + head * HD, qw, qb, kw, kb, N, nv, H, hd, smraw_attn, wave, lane);
}

template <int NP, int HD, int WPB = 2> static constexpr size_t attn_mfma_lds_bytes() {
    constexpr int QLD = HD + 8, PLD = NP + 8;
    constexpr int QK = NP * QLD, P = NP * PLD;
    if (WPB > NP / 16) return (size_t)((QK > P ? QK : P) + QK + HD * PLD + P) * 2;
    constexpr int R0 = QK > P ? QK : P;
    return (size_t)(R0 + QK + HD * PLD) * 2;
}

// ------------------------------------------------------------------------------------------ q|k|v GEMM + attention, one launch
// One workgroup per (sequence, head): the head's 3 * HD rows of the q|k|v weight (3 * HD * H bf16 = 384 KB at H = 1024) times
// the sequence's token panel, then the attention above on the result -- the q|k|v activation never leaves the CU and the block
// loses one dependent launch (layers.py:56-87; there is no full-row LayerNorm between the projection and the attention, q/k
// LayerNorm is per head).  Twelve waves; wave w owns output columns [16 w, 16 w + 16) of the head's q|k|v for all token rows.
// The weight comes from a copy the engine packs once at creation in MFMA A-operand order (gemm.hip pack_mfma16: the 16 rows x
// 32 k block of a fragment is 1 KB contiguous, lane l's 16 bytes at offset 16 l), so a wave's stream is one contiguous 32 KB
// run read with full-line wave instructions STRAIGHT INTO the operand registers -- no LDS round trip, and not the 16-lines-per-
// instruction pattern fragment loads from the row-major weight have (38 GB/s per CU, tools/ingest_probe.hip); two blocks of
// four k-steps stay in flight per lane.  The token panel is staged in LDS once per K chunk for all waves (XOR-swizzled 16-byte
// pieces: conflict-free fragment reads).  bf16(acc) goes to an LDS image of the qkv rows -- the same rounding the separate GEMM
// applies when it stores its output -- and two waves run attn_mfma_body on it.  Pays when the launch has >= ~128 workgroups
// (batch >= 4): a workgroup pulls its 384 KB at the per-CU rate, which 32 workgroups cannot hide.
// Workgroup barrier that makes LDS writes visible but leaves global loads in flight (__syncthreads() drains vmcnt, which
// would stall every wave on the weight blocks it has prefetched).
__device__ __forceinline__ void qa_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Hand-off between the waves of a workgroup without a workgroup barrier: a wave publishes its LDS writes and bumps a counter;
// a consumer spins (s_sleep) until the counter reaches the number of producers.  All waves of the workgroup are resident.
__device__ __forceinline__ void qa_signal(int *ctr, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // this wave's LDS writes have landed
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void qa_wait(int *ctr, int target) {
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int *>(ctr)) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// x[l] + x[l ^ 16] + x[l ^ 32] + x[l ^ 48] in every lane: v_permlane16_swap exchanges the odd 16-lane rows of its first operand
// with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second; with both
// operands holding x the two results add up to the pairwise sums.  (Inline asm: with identical inputs the builtin's second result
// came back as the first on ROCm 7.2.)
__device__ __forceinline__ float qa_sum_lane_groups(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

template <int NP, int KC> struct QkvAttnGeom {                     // KC = K chunk of the token panel staged in LDS (elements; 256 | 512 | 1024)
    static constexpr int HD = 64, WAVES = 12;
    static constexpr int KPB = KC >= 512 ? 8 : 4;                     // k-steps (of 32) per prefetch block; two blocks in flight per lane
    static constexpr int XBYTES = NP * 2 * KC;
    // behind the panel: the Q | K | V^T | P images (< 37 KB), the LayerNorm partials at + 37 KB (<= 4 KB) and the hand-off counters
    // at + 41 KB -- their own region, because the waves that finish their K loop first write them while the others still read the panel
    static constexpr int TAIL_BYTES = 42 * 1024;
    static constexpr size_t lds_bytes() { return (size_t)XBYTES + TAIL_BYTES; }
};

template <int NP, int KC, bool PAIR = false>
__global__ __launch_bounds__(768) void qkv_attn_kernel(const bf16_t *__restrict__ xa, const bf16_t *__restrict__ Wp,
                                                       bf16_t *__restrict__ o, const float *__restrict__ qw,
                                                       const float *__restrict__ qb, const float *__restrict__ kw,
                                                       const float *__restrict__ kb, const int *__restrict__ n_nodes, int B,
                                                       int N, int H, int heads) {
    using G = QkvAttnGeom<NP, KC>;
    constexpr int HD = G::HD, MT = NP / 16, KPB = G::KPB, XPITCH = 2 * KC;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_qa[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PAIR (NP = 64, N <= 32): the workgroup takes TWO sequences (rows 0..31 / 32..63 of its panel) through the head's weights -- half the
    // weight intake per sequence; the engine's choice when one sequence per workgroup would need more than two rounds of workgroups
    static_assert(!PAIR || NP == 64, "two sequences per workgroup: 2 x 32 token rows");
    const int head = blockIdx.x % heads, seq = (blockIdx.x / heads) * (PAIR ? 2 : 1);    // consecutive workgroups = the heads of one sequence (pair):
    const int nv = n_nodes[seq % B];                                  // head h of every sequence lands on XCD h % 8 (one L2 copy of its weights)
    const int nv1 = PAIR ? n_nodes[(seq + 1) % B] : 0;
    auto tok_of = [&](int row) { return PAIR ? (row & 31) : row; };              // token index of a panel row inside its sequence
    auto grow_of = [&](int row) { return PAIR ? (row >> 5) * N + (row & 31) : row; };   // its row relative to the first sequence's first row
    unsigned char *xs = sm_qa;                                        // [NP][XPITCH] token panel chunk, 16-byte piece p of row r at p ^ (r & 15)
    int *ctr = reinterpret_cast<int *>(sm_qa + G::XBYTES + 41 * 1024);    // hand-off counters of the tail (zeroed before the first barrier)
    if (tid < 8) ctr[tid] = 0;
    LL_QA_STAMP(0);
    const int part = wid >> 2, sub = wid & 3;                         // q | k | v, 16-column group inside the head
    const int fr = lane & 15, fq = lane >> 4;
    const int kts = H / 32;
    // LayerNorm weight / bias of this lane's four output columns (q and k waves); requested first, used after the K loop
    const float4 lnw = *reinterpret_cast<const float4 *>((part == 1 ? kw : qw) + sub * 16 + fq * 4);
    const float4 lnb = *reinterpret_cast<const float4 *>((part == 1 ? kb : qb) + sub * 16 + fq * 4);
    const unsigned char *wtile = reinterpret_cast<const unsigned char *>(Wp) + (int64_t)((part * H + head * HD) / 16 + sub) * kts * 1024;   // uniform
    af32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = af32x4{0.f, 0.f, 0.f, 0.f};
    abf16x8 wa[KPB], wb[KPB];
    const int nblk = kts / KPB;
    auto loadw = [&](abf16x8 (&r)[KPB], int blk) {
        const unsigned char *pb = wtile + (int64_t)blk * (KPB * 1024);
#pragma unroll
        for (int q = 0; q < KPB; ++q) r[q] = *reinterpret_cast<const abf16x8 *>(pb + lane * 16 + q * 1024);
    };
    // fragment (token row mt * 16 + fr, 16-byte piece A + fq) with A a multiple of 4 sits at piece (A ^ (fr & 12)) + (fq ^ (fr & 3))
    const int flo = fr * XPITCH + ((fq ^ (fr & 3)) << 4), fhi = fr & 12;
    auto mulblk = [&](const abf16x8 (&r)[KPB], int pin) {             // pin = first 16-byte piece of the block inside the staged chunk
        const unsigned char *xb = xs + flo + (pin << 4);
#if defined(LL_QA_MODE) && LL_QA_MODE == 5      // probe: pure ingest, the blocks are only folded into the accumulator
#pragma unroll
        for (int ks = 0; ks < KPB; ++ks) acc[0][0] += __builtin_bit_cast(af32x4, r[ks])[ks & 3];
        __builtin_amdgcn_sched_barrier(0);
        return;
#endif
#pragma unroll
        for (int ks = 0; ks < KPB; ++ks) {
            const unsigned char *xk = xb + (((ks * 4) ^ fhi) << 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r[ks], *reinterpret_cast<const abf16x8 *>(xk + mt * 16 * XPITCH), acc[mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                            // the loads that follow stay behind these MFMAs, and ahead of the next block's
    };
    // Block order: the workgroups of the 2B sequences that share this head's weights (same XCD) start at different places of the
    // K loop, so between them the whole weight slice is requested within the first memory round trip and everything after it is
    // an L2 hit -- in lock step they would all wait for the same HBM line four times over.
    constexpr int BPC = KC / (32 * KPB);                              // blocks per staged chunk
    const int nch = H / KC;                                           // chunks
#if defined(LL_QA_MODE) && LL_QA_MODE == 4      // probe: the waves of a workgroup start at different blocks too
    const int rot = (seq + (wid >> 1)) % BPC, cs = (seq / BPC) % nch;
#else
    const int rot = seq % BPC, cs = (seq / BPC) % nch;
#endif
    auto chunk_of = [&](int i) { return nch == 1 ? 0 : (cs + i / BPC) % nch; };
    auto blk_in = [&](int i) { return (rot + i) % BPC; };
    auto wblock = [&](int i) { return chunk_of(i) * BPC + blk_in(i); };
    constexpr int PPR = KC / 8, PIECES = NP * PPR;                    // 16-byte pieces per panel row / per chunk (powers of two)
    constexpr int XPT = (PIECES + 767) / 768;
    const bf16_t *xseq = xa + (int64_t)seq * N * H;
    auto stage = [&](int c0, bool first) {
        u32x4 xr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int pc = (tid + i * 768) & (PIECES - 1);            // wraps onto pieces another thread stages too (same bytes)
            const int row = pc / PPR, col = pc % PPR;
#if defined(LL_QA_MODE) && LL_QA_MODE == 2      // probe: no panel loads
            xr[i] = (u32x4)(0x3c003c00u);
#else
            xr[i] = *reinterpret_cast<const u32x4 *>(xseq + (tok_of(row) < N ? grow_of(row) : 0) * H + c0 + col * 8);
#endif
        }
#if !defined(LL_QA_MODE) || LL_QA_MODE != 1
        if (first) {
#if defined(LL_QA_MODE) && LL_QA_MODE == 3      // probe: every wave's panel pieces are requested before anybody's weight blocks
            asm volatile("s_barrier" ::: "memory");
#endif
            loadw(wa, wblock(0));
            loadw(wb, wblock(1));
        }
#endif
        __builtin_amdgcn_sched_barrier(0);                            // every load above is issued before the first one is waited for
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int pc = (tid + i * 768) & (PIECES - 1);
            const int row = pc / PPR, col = pc % PPR;
            *reinterpret_cast<u32x4 *>(xs + row * XPITCH + ((col ^ (row & 15)) << 4)) = tok_of(row) < N ? xr[i] : (u32x4)(0);
        }
    };
    stage(cs * KC, true);
    LL_QA_STAMP(1);
    qa_barrier();
    LL_QA_STAMP(2);
#if defined(LL_QA_MODE) && LL_QA_MODE == 1      // probe: weight blocks requested only after the panel is staged
    loadw(wa, wblock(0));
    loadw(wb, wblock(1));
#endif
    auto pair = [&](int i, bool more) {
        if (i && i % BPC == 0) {                                      // next K chunk of the token panel (H > KC only)
            qa_barrier();
            stage(chunk_of(i) * KC, false);
            qa_barrier();
        }
        mulblk(wa, blk_in(i) * 4 * KPB);
        if (more) loadw(wa, wblock(i + 2));
        __builtin_amdgcn_sched_barrier(0);
        mulblk(wb, blk_in(i + 1) * 4 * KPB);
        if (more) loadw(wb, wblock(i + 3));
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = 0; i < nblk - 2; i += 2) pair(i, true);
    pair(nblk - 2, false);
    LL_QA_STAMP(3);
    // ---- per-head LayerNorm of q and k, V^T, all in place: wave `wid` holds output columns [16 wid, 16 wid + 16) of q|k|v for
    //      every token, acc[mt][j] = C[column wid * 16 + fq * 4 + j][token mt * 16 + fr].  The row statistics over the head's
    //      64 columns (four waves x four lane groups) go through LDS as (sum, sum of squares) partials of the bf16-rounded
    //      projection -- the value the separate GEMM would have stored; var = E[x^2] - mean^2 in f32 (64 terms).
    //      No workgroup barrier from here on: data arrives in issue order, so the q waves (0-3) leave the K loop ~700 cycles before
    //      the k waves and ~1 400 before the v waves; each group hands its part over through an LDS counter as soon as it has it.
    constexpr int QLD = HD + 8, PLD = NP + 8;
    unsigned char *tail = sm_qa + G::XBYTES;
    bf16_t *Qs = reinterpret_cast<bf16_t *>(tail);
    bf16_t *Ks = Qs + NP * QLD;
    bf16_t *Vt = Ks + NP * QLD;
    bf16_t *Ps = Vt + HD * PLD;
    float2 *st = reinterpret_cast<float2 *>(tail + 37 * 1024);        // [2][4][NP] (sum, sum of squares) per 16-column group
    if (part < 2) {
        float v[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[mt][j] = bf16_to_f32(f32_to_bf16(acc[mt][j]));
        // the four lane groups of a token are lanes fr, fr + 16, fr + 32, fr + 48: summed with the gfx950 row / half swaps on the
        // VALU (the LDS crossbar is busy with the fragment reads of the waves that are still in their K loop)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float s1 = (v[mt][0] + v[mt][1]) + (v[mt][2] + v[mt][3]);
            float s2 = (v[mt][0] * v[mt][0] + v[mt][1] * v[mt][1]) + (v[mt][2] * v[mt][2] + v[mt][3] * v[mt][3]);
            s1 = qa_sum_lane_groups(s1);
            s2 = qa_sum_lane_groups(s2);
            if (fq == 0) st[(part * 4 + sub) * NP + mt * 16 + fr] = make_float2(s1, s2);
        }
        qa_signal(ctr + part, lane);                                  // ctr[0] / ctr[1]: statistics of the q / k column groups
        qa_wait(ctr + part, 4);
        bf16_t *dst = part == 0 ? Qs : Ks;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int tok = mt * 16 + fr;
            const float2 t0 = st[(part * 4 + 0) * NP + tok], t1 = st[(part * 4 + 1) * NP + tok];
            const float2 t2 = st[(part * 4 + 2) * NP + tok], t3 = st[(part * 4 + 3) * NP + tok];
            const float s1 = (t0.x + t1.x) + (t2.x + t3.x), s2 = (t0.y + t1.y) + (t2.y + t3.y);
            const float mean = s1 * (1.f / HD);
            const float rstd = rsqrtf(fmaxf(s2 * (1.f / HD) - mean * mean, 0.f) + 1e-5f);
            const bool live = tok_of(tok) < N;
            const float o0 = live ? (v[mt][0] - mean) * rstd * lnw.x + lnb.x : 0.f, o1 = live ? (v[mt][1] - mean) * rstd * lnw.y + lnb.y : 0.f;
            const float o2 = live ? (v[mt][2] - mean) * rstd * lnw.z + lnb.z : 0.f, o3 = live ? (v[mt][3] - mean) * rstd * lnw.w + lnb.w : 0.f;
            *reinterpret_cast<uint2 *>(dst + tok * QLD + sub * 16 + fq * 4) =
                make_uint2((uint32_t)f32_to_bf16(o0) | ((uint32_t)f32_to_bf16(o1) << 16), (uint32_t)f32_to_bf16(o2) | ((uint32_t)f32_to_bf16(o3) << 16));
        }
        qa_signal(ctr + 2 + part, lane);                              // ctr[2] / ctr[3]: Q / K image column groups written
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int tok = mt * 16 + fr;
            const bool live = tok_of(tok) < N;
#pragma unroll
            for (int j = 0; j < 4; ++j) Vt[(sub * 16 + fq * 4 + j) * PLD + tok] = live ? f32_to_bf16(acc[mt][j]) : (bf16_t)0;
        }
        qa_signal(ctr + 4, lane);                                     // ctr[4]: V^T column groups written
    }
    LL_QA_STAMP(4);
    if (wid >= MT) return;                                            // one wave per query tile (they are q waves) runs the attention
    qa_wait(ctr + 2, 4);
    qa_wait(ctr + 3, 4);
    auto wait_v = [&]() { qa_wait(ctr + 4, 4); };
    attn_core<NP, HD, MT, true, decltype(wait_v), PAIR>(Qs, Ks, Vt, Ps, o + (int64_t)seq * N * H + head * HD, N, nv, H, wid, lane,
                                                        rsqrtf((float)HD) * 1.44269504088896340736f, wait_v, nv1);
    LL_QA_STAMP(7);
}

// ------------------------------------------------------------------------------------------ AdaLN epilogue
// x += gate * (LN0(y) * (1 + scale) + shift)       (transformer.py:142-143; LN0 = no affine, eps 1e-5)
// y = sum of `nslab` split-K partial slabs (+ bias), summed in slab order (deterministic).
// One wave per token row; modulation rows come from the hoisted table mod[T][B+1][L][6H].
template <typename T, int NS, int MAXE>
__global__ __launch_bounds__(64) void ln_mod_res_kernel(const float *__restrict__ y, int64_t slab_stride,
                                                          const float *__restrict__ bias, float *__restrict__ x32,
                                                          T *__restrict__ xa, const float *__restrict__ modtab,
                                                          const int *__restrict__ step_ptr,
                                                          const int *__restrict__ rowvec /*[B] table rows or null*/,
                                                          const float *__restrict__ modcur /*[B+1][L][6H] rows of the current step or null*/,
                                                          int layer, int sel, int B, int N, int H, int L, int M2, int Ht) {
    // one 64-lane wave = one token row = one workgroup: rows spread over as many CUs as possible, because the
    // per-CU load path (~25-40 GB/s), not HBM, bounds these small row kernels
    const int row = blockIdx.x;
    if (row >= M2) return;
    const int lane = threadIdx.x & 63;
    const int seq = row / N;
    // table row: the shared reverse step, or (training forward) a per-graph row -- t differs from graph to graph there
    const int ci = (seq < B) ? seq : B;  // unconditional pass shares one row
    // modcur: this step's rows staged at a fixed address by stage_mod_kernel -- the modulation loads then do not wait for the
    // step index (one memory round trip per launch instead of two); the table walk stays for per-graph rows (training forward)
    const float *mod;
    if (modcur) {
        mod = modcur + ((int64_t)ci * L + layer) * (6 * (int64_t)H) + (int64_t)sel * 3 * H;
    } else {
        const int s = rowvec ? rowvec[seq < B ? seq : seq - B] : *step_ptr;
        mod = modtab + (((int64_t)s * (B + 1) + ci) * L + layer) * (6 * (int64_t)H) + (int64_t)sel * 3 * H;
    }
    // MAXE float4 chunks per lane (H <= 256*MAXE).  Every load of the row -- split-K slabs, bias, residual
    // and the three modulation vectors -- is issued up front: one memory round trip after the step index.
    float4 v[MAXE], xr[MAXE], sh[MAXE], sc[MAXE], ga[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = (lane + e * 64) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h < H) {
            float4 t[NS];
#pragma unroll
            for (int z = 0; z < NS; ++z) t[z] = *reinterpret_cast<const float4 *>(y + z * slab_stride + (int64_t)row * H + h);
            const float4 bb = *reinterpret_cast<const float4 *>(bias + h);
            xr[e] = *reinterpret_cast<const float4 *>(x32 + (int64_t)row * H + h);
            sh[e] = *reinterpret_cast<const float4 *>(mod + h);
            sc[e] = *reinterpret_cast<const float4 *>(mod + H + h);
            ga[e] = *reinterpret_cast<const float4 *>(mod + 2 * H + h);
#pragma unroll
            for (int z = 0; z < NS; ++z) { a.x += t[z].x; a.y += t[z].y; a.z += t[z].z; a.w += t[z].w; }
            a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w;
        }
        v[e] = a;
    }
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) sum += v[e].x + v[e].y + v[e].z + v[e].w;
    // H = row pitch (padded width), Ht = the checkpoint's hidden_size: the padded columns of y are exact zeros (zero weight rows and bias)
    // and have zero gates, so they stay zero in the residual stream; the statistics run over the Ht true columns
    const float mean = wave_sum(sum) / (float)Ht;
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) var += sq_dev4(v[e], mean, (lane + e * 64) * 4, Ht);
    const float rstd = rsqrtf(wave_sum(var) / (float)Ht + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = (lane + e * 64) * 4;
        if (h < H) {
            float4 o;
            o.x = xr[e].x + ga[e].x * ((v[e].x - mean) * rstd * (1.f + sc[e].x) + sh[e].x);
            o.y = xr[e].y + ga[e].y * ((v[e].y - mean) * rstd * (1.f + sc[e].y) + sh[e].y);
            o.z = xr[e].z + ga[e].z * ((v[e].z - mean) * rstd * (1.f + sc[e].z) + sh[e].z);
            o.w = xr[e].w + ga[e].w * ((v[e].w - mean) * rstd * (1.f + sc[e].w) + sh[e].w);
            *reinterpret_cast<float4 *>(x32 + (int64_t)row * H + h) = o;
            store4<T>(xa + (int64_t)row * H + h, o);
        }
    }
}

// The same row operation with MAXE waves per row: wave e owns the e-th 256-column chunk, so a lane holds NS + 5 loads in flight
// instead of MAXE * (NS + 5) (tools/phase_floor_probe.hip: beyond ~16 outstanding loads per thread the ingest of a launch gets
// slower).  The reductions keep the single-wave kernel's order -- per lane the chunk partials are added in chunk order, then the
// wave reduction -- so the result is bit-identical to ln_mod_res_kernel.
template <typename T, int NS, int MAXE>
__global__ __launch_bounds__(64 * MAXE) void ln_mod_res_mw_kernel(const float *__restrict__ y, int64_t slab_stride,
                                                                   const float *__restrict__ bias, float *__restrict__ x32,
                                                                   T *__restrict__ xa, const float *__restrict__ modtab,
                                                                   const int *__restrict__ step_ptr, const int *__restrict__ rowvec,
                                                                   const float *__restrict__ modcur,
                                                                   int layer, int sel, int B, int N, int H, int L, int M2, int Ht) {
    __shared__ float part[2][MAXE][64];
    const int row = blockIdx.x;
    if (row >= M2) return;
    const int lane = threadIdx.x & 63, e = threadIdx.x >> 6;
    const int seq = row / N;
    const int ci = (seq < B) ? seq : B;
    const float *mod;
    if (modcur) {
        mod = modcur + ((int64_t)ci * L + layer) * (6 * (int64_t)H) + (int64_t)sel * 3 * H;
    } else {
        const int s = rowvec ? rowvec[seq < B ? seq : seq - B] : *step_ptr;
        mod = modtab + (((int64_t)s * (B + 1) + ci) * L + layer) * (6 * (int64_t)H) + (int64_t)sel * 3 * H;
    }
    const int h = (lane + e * 64) * 4;
    const bool ok = h < H;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), xr = v, sh = v, sc = v, ga = v;
    if (ok) {
        float4 t[NS];
#pragma unroll
        for (int z = 0; z < NS; ++z) t[z] = *reinterpret_cast<const float4 *>(y + z * slab_stride + (int64_t)row * H + h);
        const float4 bb = *reinterpret_cast<const float4 *>(bias + h);
        xr = *reinterpret_cast<const float4 *>(x32 + (int64_t)row * H + h);
        sh = *reinterpret_cast<const float4 *>(mod + h);
        sc = *reinterpret_cast<const float4 *>(mod + H + h);
        ga = *reinterpret_cast<const float4 *>(mod + 2 * H + h);
#pragma unroll
        for (int z = 0; z < NS; ++z) { v.x += t[z].x; v.y += t[z].y; v.z += t[z].z; v.w += t[z].w; }
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    }
    float mean, rstd;
    if (sizeof(T) == 2) {
        // bf16 engine: sum and sum of squares in ONE exchange (one barrier and one LDS round trip less per launch; the f32 engine keeps
        // the two-pass form the golden-vector tests pin): var = E[y^2] - mean^2 in f32 over H terms, clamped at 0
        part[0][e][lane] = v.x + v.y + v.z + v.w;
        part[1][e][lane] = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        __syncthreads();
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
            sum += part[0][k][lane];
            sq += part[1][k][lane];
        }
        mean = wave_sum(sum) / (float)Ht;       // (padded columns are exact zeros: they add nothing to either sum)
        rstd = rsqrtf(fmaxf(wave_sum(sq) / (float)Ht - mean * mean, 0.f) + 1e-5f);
    } else {
        part[0][e][lane] = v.x + v.y + v.z + v.w;
        __syncthreads();
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < MAXE; ++k) sum += part[0][k][lane];
        mean = wave_sum(sum) / (float)Ht;
        part[1][e][lane] = sq_dev4(v, mean, h, Ht);
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int k = 0; k < MAXE; ++k) var += part[1][k][lane];
        rstd = rsqrtf(wave_sum(var) / (float)Ht + 1e-5f);
    }
    if (ok) {
        float4 o;
        o.x = xr.x + ga.x * ((v.x - mean) * rstd * (1.f + sc.x) + sh.x);
        o.y = xr.y + ga.y * ((v.y - mean) * rstd * (1.f + sc.y) + sh.y);
        o.z = xr.z + ga.z * ((v.z - mean) * rstd * (1.f + sc.z) + sh.z);
        o.w = xr.w + ga.w * ((v.w - mean) * rstd * (1.f + sc.w) + sh.w);
        *reinterpret_cast<float4 *>(x32 + (int64_t)row * H + h) = o;
        store4<T>(xa + (int64_t)row * H + h, o);
    }
}

// ------------------------------------------------------------------------------------------ posterior + CFG + sampling
// Two launches replace ~40 ATen launches, six dense [B,F,F] builds and six bmm per step of the reference
// (diffusion_model.py:328-399, diffusion_utils.py:316-349, 376-413, 476-492) with O(N*F) work per graph:
//   post_rows_kernel : one wave per decoder row (pass, graph, node): LN0 + modulate in place, atom softmax, pred_X.u_xe
//   post_pairs_kernel: one wave per (graph, node i): bond softmax of row i, structured posterior of node i and of the
//                      pairs (i, j>i), classifier-free guidance, clamp/renorm, Exp(1)-race sampling, state write.
// The graph state is double-buffered (z_{s+1} in half (s+1)&1, z_s written to half s&1) so rows can be processed by
// independent workgroups without read/write hazards on E.
struct PostArgs {
    float *out;          // [2][B][N][F] decoder output (fc2 + bias); normalised + modulated in place by post_rows
    const float *modo;   // [T][B+1][2F] output-layer modulation (shift | scale)
    float *predX;        // [2][B][N][16] scratch
    float *pxe;          // [2][B][N][8]  scratch: sum_a predX[a] u_xe[a][k]
    int8_t *X;           // [2][B][N]      double-buffered state
    int8_t *E;           // [2][B][N][N]
    const int *n_nodes;  // [B]
    const float *x_marg, *e_marg, *u_xe, *u_ex, *betas, *alphas_bar;
    const float *qx, *qe;          // injected Exp(1) noise or null
    const unsigned long long *seed_ptr;
    const int *step_ptr;
    const int *rowvec;   // [B] per-graph table rows (training forward) or null
    int B, N, F, T;
    float guide;
    float *pX_out, *pE_out;  // optional taps
    float *logX, *logE;      // optional taps [2][B][N][16], [2][B][N][N][5]
    int update_state;
};

// one wave per decoder row `row` = (p*B + b)*N + i; s_known >= 0: the reverse step (the caller knows it), else read from the arguments
template <int MAXF>      // 64-column chunks of a decoder row: F = 16 + 5 N <= 64 MAXF (6 up to 64 nodes, 11 up to 128)
__device__ __forceinline__ void post_rows_body(const PostArgs &a, int row, int lane, int s_known) {
    const int N = a.N, F = a.F, B = a.B;
    if (row < 0 || row >= 2 * B * N) return;
    const int p = row / (B * N);
    const int bi = row - p * B * N;
    const int b = bi / N, i = bi - b * N;
    const int s = s_known >= 0 ? s_known : (a.rowvec ? a.rowvec[b] : *a.step_ptr);
    const int ci = p == 0 ? b : B;
    const float *ss = a.modo + ((int64_t)s * (B + 1) + ci) * (2 * F);
    float *r = a.out + (int64_t)row * F;
    float v[MAXF], sh[MAXF], sc[MAXF];
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < MAXF; ++e) {
        const int f = lane + e * 64;
        v[e] = f < F ? *(r + f) : 0.f;
        sh[e] = f < F ? ss[f] : 0.f;
        sc[e] = f < F ? ss[F + f] : 0.f;
        sm += v[e];
    }
    const float mean = wave_sum(sm) / (float)F;
    float vr = 0.f;
#pragma unroll
    for (int e = 0; e < MAXF; ++e) {
        const float d = v[e] - mean;
        vr += (lane + e * 64 < F) ? d * d : 0.f;
    }
    const float rstd = rsqrtf(wave_sum(vr) / (float)F + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXF; ++e) {
        const int f = lane + e * 64;
        v[e] = (v[e] - mean) * rstd * (1.f + sc[e]) + sh[e];
        if (f < F) r[f] = v[e];
    }
    // atom classes live in lanes 0..15 of chunk 0
    const int st_in = (s + 1) & 1;
    const int xi = *(a.X + ((int64_t)st_in * B + b) * N + i);
    const bool valid = i < a.n_nodes[b];
    if (lane < 16) {
        const float l = valid ? ((xi == lane ? 1.f : 0.f) + v[0]) : 0.f;
        if (a.logX) a.logX[(int64_t)row * XD + lane] = l;
        const float mx = row16_max(l);
        const float ex = expf(l - mx);
        const float pv = ex / row16_sum(ex);
        a.predX[(int64_t)row * XD + lane] = pv;
        float t[ED];
#pragma unroll
        for (int k = 0; k < ED; ++k) t[k] = row16_sum(pv * a.u_xe[lane * ED + k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < ED; ++k) a.pxe[(int64_t)row * 8 + k] = t[k];
        }
    }
}

template <int MAXF>
__global__ __launch_bounds__(256) void post_rows_kernel(PostArgs a) {
    post_rows_body<MAXF>(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, -1);
}

// NW waves per (graph b, node i); thread j = partner node (NW = 1 up to 64 nodes, 2 up to 128: the sums and counts over the partners of a
// row are then added up across the two waves through LDS, wave 0 first -- with one wave the arithmetic is the historical one)
template <int NW>
__device__ __forceinline__ void post_pairs_body(const PostArgs &a, int i, int b, int j, int s_known) {
    __shared__ float xch[NW > 1 ? NW * 16 : 1];
    const int lane = j & 63, wv = j >> 6;
    auto row_sum = [&](float x, int slot) -> float {          // sum over all partners j of the row
        const float w = wave_sum(x);
        if (NW == 1) return w;
        __syncthreads();                                      // the slot's previous use has been read
        if (lane == 0) xch[wv * 16 + slot] = w;
        __syncthreads();
        float t = xch[slot];
#pragma unroll
        for (int u = 1; u < NW; ++u) t += xch[u * 16 + slot];
        return t;
    };
    const int N = a.N, F = a.F, B = a.B;
    const int s = s_known >= 0 ? s_known : *a.step_ptr;
    const int nv = a.n_nodes[b];
    const int st_in = (s + 1) & 1, st_out = s & 1;
    const int8_t *Xin = a.X + ((int64_t)st_in * B + b) * N;
    const int8_t *Ein = a.E + ((int64_t)st_in * B + b) * N * N;
    const bool vi = i < nv, vj = j < nv && j < N;
    const int xi = *(Xin + i);
    const int eij = j < N ? (int)*(Ein + i * N + j) : -1;
    const int eji = j < N ? (int)*(Ein + j * N + i) : -1;
    const float beta = a.betas[s + 1], ab_s = a.alphas_bar[s], ab_t = a.alphas_bar[s + 1];
    const bool guided = (a.guide != 1.0f);
    const unsigned long long seed = a.seed_ptr ? *a.seed_ptr : 0ull;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));

    // ---- pred_E[i][j][:] for both passes (final masked, symmetrised logits -> softmax), SE = sum_j pred_E
    float e5[2][ED], SE[2][ED], SEt[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float l[ED];
        if (vi && vj && i != j) {
            const float *zi = a.out + ((((int64_t)p * B + b) * N + i) * F) + XD + ED * j;
            const float *zj = a.out + ((((int64_t)p * B + b) * N + j) * F) + XD + ED * i;
#pragma unroll
            for (int k = 0; k < ED; ++k) l[k] = 0.5f * (((eij == k ? 1.f : 0.f) + *(zi + k)) + ((eji == k ? 1.f : 0.f) + *(zj + k)));
        } else {
#pragma unroll
            for (int k = 0; k < ED; ++k) l[k] = 0.f;
        }
        if (a.logE && j < N) {
#pragma unroll
            for (int k = 0; k < ED; ++k) a.logE[((((int64_t)p * B + b) * N + i) * N + j) * ED + k] = l[k];
        }
        float mx = l[0];
#pragma unroll
        for (int k = 1; k < ED; ++k) mx = fmaxf(mx, l[k]);
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < ED; ++k) {
            l[k] = expf(l[k] - mx);
            sm += l[k];
        }
        const float inv = 1.f / sm;
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < ED; ++k) {
            e5[p][k] = l[k] * inv;
            SE[p][k] = row_sum(j < N ? e5[p][k] : 0.f, p * ED + k);
            tot += SE[p][k];
        }
        SEt[p] = tot;
    }
    // ---- S[i,f] = sum_g X_t[i,g] u[f,g] in structured form (diffusion_utils.py:296-305)
    float cnt[ED];
#pragma unroll
    for (int k = 0; k < ED; ++k) cnt[k] = NW == 1 ? (float)__popcll(__ballot(eij == k)) : row_sum(eij == k ? 1.f : 0.f, 10 + k);
    float emsum = 0.f;
#pragma unroll
    for (int k = 0; k < ED; ++k) emsum = fmaf(cnt[k], a.e_marg[k], emsum);

    // ---- node posterior: lanes 0..15 = atom classes
    int newX = -1;
    {
        const int c = j & 15;
        float pf = 0.f;
        if (vi) {
            float sx = (xi >= 0) ? a.x_marg[xi] : 0.f;
#pragma unroll
            for (int k = 0; k < ED; ++k) sx = fmaf(cnt[k], a.u_xe[c * ED + k], sx);
            const float xt = (xi == c) ? 1.f : 0.f;
            const float left = (1.f - beta) * xt + beta * sx;
            const float den = fmaxf(ab_t * xt + (1.f - ab_t) * sx, 1e-5f);
            float pc[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float px = *(a.predX + ((((int64_t)p * B + b) * N + i) * XD) + c);
                const float spx = row16_sum(px);
                float r = a.x_marg[c] * spx;
#pragma unroll
                for (int k = 0; k < ED; ++k) r = fmaf(SE[p][k], a.u_ex[k * XD + c], r);
                const float right = ab_s * px + (1.f - ab_s) * r;
                float un = left * right / den;
                float sum = row16_sum(un);
                if (sum == 0.f) {
                    un = 1e-5f;
                    sum = XD * 1e-5f;
                }
                pc[p] = un / sum;
            }
            if (guided) {
                const float u = pc[1];
                pf = u * powf(pc[0] / fmaxf(u, 1e-5f), a.guide);
                pf /= fmaxf(row16_sum(pf), 1e-5f);
            } else {
                pf = pc[0];
            }
        }
        if (a.pX_out && j < 16) a.pX_out[((int64_t)b * N + i) * XD + c] = pf;      // (every wave computes the classes in its lanes 0..15)
        if (vi) {   // sample_discrete_features (diffusion_utils.py:386-395): clamp, renormalise, race
            const float pcl = fmaxf(pf, 1e-5f);
            const float sum = row16_sum(pcl);
            float q;
            if (a.qx) {
                q = a.qx[((int64_t)b * N + i) * XD + c];
            } else {
                const uint4 r = dit_noise_x4(key, s, b * N + i, c >> 2);
                const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
                q = exp1_from_bits(rr[c & 3]);
            }
            const float v = (pcl / sum) / q;
            const float best = row16_max(v);
            const unsigned long long m = __ballot(v == best && lane < 16);
            newX = __ffsll((long long)m) - 1;      // first maximum wins, like argmax
        }
    }
    // ---- bond posterior of the pairs (i, j > i)   (strict upper triangle: diffusion_utils.py:409-411)
    int val = -1;
    if (j > i && j < N && vi && vj) {
        float se = ((xi >= 0) ? 0.f : 0.f);
        float pc[2][ED];
        for (int p = 0; p < (guided ? 2 : 1); ++p) {
            const float *px = a.pxe + ((((int64_t)p * B + b) * N + i) * 8);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < ED; ++k) {
                const float sek = ((xi >= 0) ? a.u_ex[k * XD + xi] : 0.f) + emsum;
                const float r = *(px + k) + a.e_marg[k] * SEt[p];
                const float right = ab_s * e5[p][k] + (1.f - ab_s) * r;
                const float et = (eij == k) ? 1.f : 0.f;
                const float left = (1.f - beta) * et + beta * sek;
                const float den = fmaxf(ab_t * et + (1.f - ab_t) * sek, 1e-5f);
                const float un = left * right / den;
                pc[p][k] = un;
                sum += un;
            }
            if (sum == 0.f) {
#pragma unroll
                for (int k = 0; k < ED; ++k) pc[p][k] = 1e-5f;
                sum = ED * 1e-5f;
            }
#pragma unroll
            for (int k = 0; k < ED; ++k) pc[p][k] /= sum;
        }
        (void)se;
        float pf[ED];
        if (guided) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < ED; ++k) {
                const float u = pc[1][k];
                pf[k] = u * powf(pc[0][k] / fmaxf(u, 1e-5f), a.guide);
                sum += pf[k];
            }
            sum = fmaxf(sum, 1e-5f);
#pragma unroll
            for (int k = 0; k < ED; ++k) pf[k] /= sum;
        } else {
#pragma unroll
            for (int k = 0; k < ED; ++k) pf[k] = pc[0][k];
        }
        if (a.pE_out) {
#pragma unroll
            for (int k = 0; k < ED; ++k) a.pE_out[(((int64_t)b * N + i) * N + j) * ED + k] = pf[k];
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < ED; ++k) {
            pf[k] = fmaxf(pf[k], 1e-5f);
            sum += pf[k];
        }
        float q[8];
        if (a.qe) {
#pragma unroll
            for (int k = 0; k < ED; ++k) q[k] = a.qe[(((int64_t)b * N + i) * N + j) * ED + k];
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint4 r = dit_noise_e4(key, s, (b * N + i) * N + j, g);
                q[g * 4 + 0] = exp1_from_bits(r.x);
                q[g * 4 + 1] = exp1_from_bits(r.y);
                q[g * 4 + 2] = exp1_from_bits(r.z);
                q[g * 4 + 3] = exp1_from_bits(r.w);
            }
        }
        float best = -1.f;
        int arg = 0;
#pragma unroll
        for (int k = 0; k < ED; ++k) {
            const float v = (pf[k] / sum) / q[k];
            if (v > best) {
                best = v;
                arg = k;
            }
        }
        val = arg;
    }
    if (a.update_state) {
        int8_t *Xout = a.X + ((int64_t)st_out * B + b) * N;
        int8_t *Eout = a.E + ((int64_t)st_out * B + b) * N * N;
        if (j > i && j < N) {
            Eout[i * N + j] = (int8_t)val;
            Eout[j * N + i] = (int8_t)val;
        }
        if (j == i) {
            Eout[i * N + i] = vi ? (int8_t)0 : (int8_t)-1;
            Xout[i] = (int8_t)newX;
        }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void post_pairs_kernel(PostArgs a) { post_pairs_body<NW>(a, blockIdx.x, blockIdx.y, threadIdx.x, -1); }

// ------------------------------------------------------------------------------------------ z_T
// sample_discrete_feature_noise (diffusion_utils.py:495-518): limit marginals, strict upper triangle kept,
// symmetrised, masked; the diagonal stays the all-zero vector (-1).
__global__ __launch_bounds__(256) void init_state_kernel(int8_t *X, int8_t *E, const int *n_nodes, const float *x_marg,
                                                          const float *e_marg, const float *qx, const float *qe,
                                                          const unsigned long long *seed_ptr, int B, int N, int T) {
    const int b = blockIdx.x;
    const int nv = n_nodes[b];
    const unsigned long long seed = seed_ptr ? *seed_ptr : 0ull;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    for (int i = threadIdx.x; i < N; i += 256) {
        int arg = -1;
        if (i < nv) {
            float best = -1.f;
            for (int g = 0; g < 4; ++g) {
                float q[4];
                if (qx) {
                    for (int c = 0; c < 4; ++c) q[c] = qx[((int64_t)b * N + i) * XD + g * 4 + c];
                } else {
                    const uint4 r = dit_noise_x4(key, T, b * N + i, g);
                    q[0] = exp1_from_bits(r.x); q[1] = exp1_from_bits(r.y);
                    q[2] = exp1_from_bits(r.z); q[3] = exp1_from_bits(r.w);
                }
                for (int c = 0; c < 4; ++c) {
                    const float v = x_marg[g * 4 + c] / q[c];
                    if (v > best) { best = v; arg = g * 4 + c; }
                }
            }
        }
        X[(int64_t)b * N + i] = (int8_t)arg;
        E[((int64_t)b * N + i) * N + i] = -1;
    }
    const int npairs = N * (N - 1) / 2;
    for (int pi = threadIdx.x; pi < npairs; pi += 256) {
        int i = 0, rem = pi;
        while (rem >= N - 1 - i) { rem -= N - 1 - i; ++i; }
        const int j = i + 1 + rem;
        int arg = -1;
        if (i < nv && j < nv) {
            float q[8];
            if (qe) {
                for (int k = 0; k < ED; ++k) q[k] = qe[(((int64_t)b * N + i) * N + j) * ED + k];
            } else {
                for (int g = 0; g < 2; ++g) {
                    const uint4 r = dit_noise_e4(key, T, (b * N + i) * N + j, g);
                    q[g * 4 + 0] = exp1_from_bits(r.x); q[g * 4 + 1] = exp1_from_bits(r.y);
                    q[g * 4 + 2] = exp1_from_bits(r.z); q[g * 4 + 3] = exp1_from_bits(r.w);
                }
            }
            float best = -1.f;
            for (int k = 0; k < ED; ++k) {
                const float v = e_marg[k] / q[k];
                if (v > best) { best = v; arg = k; }
            }
        }
        E[((int64_t)b * N + i) * N + j] = (int8_t)arg;
        E[((int64_t)b * N + j) * N + i] = (int8_t)arg;
    }
}

// ------------------------------------------------------------------------------------------ cold path (per batch)
// Sinusoidal timestep features for every step s: t = (s+1)/T fractional (conditions.py:32-51).  Row Tsteps holds t = 0,
// which only the training forward draws (diffusion_model.py:201-206).
template <typename T>
__global__ void tfreq_kernel(T *out, int Tsteps) {
    const int s = blockIdx.x;
    const int j = threadIdx.x;  // 0..127
    const float t = s < Tsteps ? (float)(s + 1) / (float)Tsteps : 0.f;
    const float f = expf(-logf(10000.f) * (float)j / 128.f);
    const float arg = t * f;
    out[(int64_t)s * 256 + j] = from_f32<T>(cosf(arg));
    out[(int64_t)s * 256 + 128 + j] = from_f32<T>(sinf(arg));
}

// Property features: Z[b][d*H + h] = softmax_h(y[b,d] * w0_d[h] + b0_d[h]), zero row when y is NaN
// (conditions.py:66-70, 76-93).  One workgroup per (b, d).
template <typename T>
__global__ __launch_bounds__(256) void yfeat_kernel(const float *__restrict__ props, const float *__restrict__ w0,
                                                     const float *__restrict__ b0, T *__restrict__ Z,
                                                     int8_t *__restrict__ ynan, int H, int Ht) {
    // H = row pitch of Z / w0 / b0 (padded width), Ht = the checkpoint's hidden_size: the softmax runs over the Ht true columns
    // (conditions.py:68, Softmax(dim=1) of a [n, hidden_size] tensor); the padded columns of Z are zeros
    __shared__ float red[4];
    const int b = blockIdx.x, d = blockIdx.y;
    const float y = props[b * LL_YDIM + d];
    const bool drop = (y != y);
    if (threadIdx.x == 0) ynan[b * LL_YDIM + d] = drop ? 1 : 0;
    T *z = Z + ((int64_t)b * LL_YDIM + d) * H;
    if (drop) {
        for (int h = threadIdx.x; h < H; h += 256) z[h] = from_f32<T>(0.f);
        return;
    }
    const float *w = w0 + (int64_t)d * H;
    const float *bb = b0 + (int64_t)d * H;
    float mx = -INFINITY;
    for (int h = threadIdx.x; h < Ht; h += 256) mx = fmaxf(mx, fmaf(y, w[h], bb[h]));
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sm = 0.f;
    for (int h = threadIdx.x; h < Ht; h += 256) sm += expf(fmaf(y, w[h], bb[h]) - mx);
    sm = block_sum_256(sm, red);
    const float inv = 1.f / sm;
    for (int h = threadIdx.x; h < H; h += 256) z[h] = from_f32<T>(h < Ht ? expf(fmaf(y, w[h], bb[h]) - mx) * inv : 0.f);
}

// text rows -> operand dtype, flagging rows that contain a NaN (conditions.py:112)
template <typename T>
__global__ __launch_bounds__(256) void txt_prep_kernel(const float *__restrict__ txt, T *__restrict__ out,
                                                        int8_t *__restrict__ tnan, int D) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float bad = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float v = txt[(int64_t)b * D + d];
        bad += (v != v) ? 1.f : 0.f;
        out[(int64_t)b * D + d] = from_f32<T>((v != v) ? 0.f : v);
    }
    bad = block_sum_256(bad, red);
    if (threadIdx.x == 0) tnan[b] = bad > 0.f ? 1 : 0;
}

// c[s][ci] = c_t[s] + c_y[ci] + c_txt[ci]; row ci == B is the unconditional embedding
// (all property slots and the text dropped: conditions.py:78-79, 109-110).
template <typename T>
__global__ void combine_c_kernel(const float *__restrict__ ct, const float *__restrict__ cy_lin,
                                 const float *__restrict__ ctxt, const float *__restrict__ drop_y,
                                 const float *__restrict__ drop_txt, const int8_t *__restrict__ ynan,
                                 const int8_t *__restrict__ tnan, float *__restrict__ c32, T *__restrict__ ca, int B,
                                 int H) {
    const int s = blockIdx.x, ci = blockIdx.y;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        float cy = 0.f, cx;
        if (ci < B) {
            cy = cy_lin[(int64_t)ci * H + h];
            for (int d = 0; d < LL_YDIM; ++d)
                if (ynan[ci * LL_YDIM + d]) cy += drop_y[(int64_t)d * H + h];
            cx = tnan[ci] ? drop_txt[h] : ctxt[(int64_t)ci * H + h];
        } else {
            for (int d = 0; d < LL_YDIM; ++d) cy += drop_y[(int64_t)d * H + h];
            cx = drop_txt[h];
        }
        const float v = ct[(int64_t)s * H + h] + cy + cx;
        const int64_t o = ((int64_t)s * (B + 1) + ci) * H + h;
        c32[o] = v;
        ca[o] = from_f32<T>(v);
    }
}

// training forward: timestep t in 0..T -> row of the hoisted tables (t = 0 lives in row T); out-of-range t clamps
__global__ void t_to_row_kernel(const int *__restrict__ t_int, int *__restrict__ rows, int B, int T) {
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        int t = t_int[b];
        t = t < 0 ? 0 : (t > T ? T : t);
        rows[b] = t == 0 ? T : t - 1;
    }
}
__global__ void set_scalars_kernel(int *step_ptr, int s, unsigned long long *seed_ptr, unsigned long long seed) {
    *step_ptr = s;
    *seed_ptr = seed;
}
__global__ void advance_step_kernel(int *step_ptr) { *step_ptr = *step_ptr - 1; }

// modcur[ci][l][6H] = modtab[s][ci][l][6H] for the step s in device memory: the first node of every step
__global__ __launch_bounds__(256) void stage_mod_kernel(const float *__restrict__ modtab, float *__restrict__ modcur,
                                                         const int *__restrict__ step_ptr, int64_t row_floats) {
    const int s = *step_ptr;
    const float4 *src = reinterpret_cast<const float4 *>(modtab + (int64_t)s * row_floats);
    float4 *dst = reinterpret_cast<float4 *>(modcur);
    const int64_t n4 = row_floats / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// out[c][r] = in[r][c]  (weight re-layout at create time)
__global__ void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int R, int C) {
    const int64_t n = (int64_t)R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i - (int64_t)r * C);
        out[(int64_t)c * R + r] = in[i];
    }
}

// Wcat[h][d*H + k] = W2_d[h][k]   (10 property MLP output weights concatenated along K)
__global__ void ycat_kernel(const float *const *__restrict__ w2, float *__restrict__ out, int H) {
    const int d = blockIdx.y;
    const float *src = w2[d];
    const int64_t n = (int64_t)H * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(i / H), k = (int)(i - (int64_t)h * H);
        out[(int64_t)h * (LL_YDIM * H) + (int64_t)d * H + k] = src[i];
    }
}

}  // namespace ll
