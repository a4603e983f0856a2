// Device kernels of the GraphDiT reverse-diffusion step (gfx950).  Included by graphdit.hip only.
//
// State representation (MI355X-first: the dense one-hot float tensors of the reference,
// X [B,N,16] / E [B,N,N,5], are never materialised): X int8 [B,N], E int8 [B,N,N];
// -1 encodes the all-zero one-hot vector (masked node / masked pair / the z_T diagonal).
#pragma once
#include "common.h"

namespace ll {

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

// ------------------------------------------------------------------------------------------ x_embedder
// h = LayerNorm_affine(W_x . [onehot(x_i) | onehot(e_i0) .. onehot(e_i,N-1)])   (transformer.py:41-44, 95-96)
// The input row has at most N+1 non-zeros, so the Linear is a gather-sum of rows of W_x^T [F,H].
// One workgroup per (graph, node); the result is identical for the conditional and the
// unconditional pass and is written to both halves of the residual stream.
template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const int8_t *__restrict__ X, const int8_t *__restrict__ E,
                                                     const float *__restrict__ WxT, const float *__restrict__ lnw,
                                                     const float *__restrict__ lnb, float *__restrict__ x32,
                                                     T *__restrict__ xa, int B, int N, int H) {
    __shared__ float red[4];
    __shared__ int gidx[72];
    __shared__ int ng;
    const int row = blockIdx.x;  // b*N + i
    const int b = row / N;
    const int i = row - b * N;
    if (threadIdx.x == 0) {
        int n = 0;
        const int xi = X[row];
        if (xi >= 0) gidx[n++] = xi;
        const int8_t *er = E + ((int64_t)b * N + i) * N;
        for (int j = 0; j < N; ++j) {
            const int e = er[j];
            if (e >= 0) gidx[n++] = XD + ED * j + e;
        }
        ng = n;
    }
    __syncthreads();
    constexpr int MAXE = 8;  // H <= 2048
    float v[MAXE];
    const int n = ng;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = threadIdx.x + e * 256;
        float s = 0.f;
        if (h < H)
            for (int g = 0; g < n; ++g) s += WxT[(int64_t)gidx[g] * H + h];
        v[e] = s;
    }
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) ls += (threadIdx.x + e * 256 < H) ? v[e] : 0.f;
    const float mean = block_sum_256(ls, red) / (float)H;
    float lv = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const float d = v[e] - mean;
        lv += (threadIdx.x + e * 256 < H) ? d * d : 0.f;
    }
    const float rstd = rsqrtf(block_sum_256(lv, red) / (float)H + 1e-5f);
    const int64_t M = (int64_t)B * N;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = threadIdx.x + e * 256;
        if (h < H) {
            const float o = (v[e] - mean) * rstd * lnw[h] + lnb[h];
            x32[(int64_t)row * H + h] = o;
            x32[(M + row) * H + h] = o;
            xa[(int64_t)row * H + h] = from_f32<T>(o);
            xa[(M + row) * H + h] = from_f32<T>(o);
        }
    }
}

// ------------------------------------------------------------------------------------------ attention (generic)
// Per (sequence, head): LayerNorm(hd, affine) on q and k rows, key mask valid_i & valid_j with padded
// query rows opened to all keys, softmax(q k^T / sqrt(hd)) v     (layers.py:56-87).
// Generic f32 LDS kernel for any N <= 64, hd <= 128; the MFMA variant below covers the production shapes.
template <typename T>
__global__ __launch_bounds__(256) void attn_generic_kernel(const T *__restrict__ qkv, T *__restrict__ o,
                                                            const float *__restrict__ qw, const float *__restrict__ qb,
                                                            const float *__restrict__ kw, const float *__restrict__ kb,
                                                            const int *__restrict__ n_nodes, int B, int N, int H,
                                                            int hd) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int head = blockIdx.x;
    const int seq = blockIdx.y;  // pass*B + b
    const int b = seq % B;
    const int nv = n_nodes[b];
    const int ld = hd + 1;
    float *q = sm;
    float *k = q + N * ld;
    float *v = k + N * ld;
    float *S = v + N * ld;  // [N][N+1]
    const int tid = threadIdx.x;
    const int64_t rbase = (int64_t)seq * N;
    for (int idx = tid; idx < N * hd; idx += 256) {
        const int i = idx / hd, d = idx - i * hd;
        const T *r = qkv + (rbase + i) * (3 * (int64_t)H) + head * hd + d;
        q[i * ld + d] = to_f32<T>(r[0]);
        k[i * ld + d] = to_f32<T>(r[H]);
        v[i * ld + d] = to_f32<T>(r[2 * H]);
    }
    __syncthreads();
    // LayerNorm over hd for the 2N rows of q and k: 2 threads per row would be enough; use one.
    if (tid < 2 * N) {
        float *r = (tid < N) ? (q + tid * ld) : (k + (tid - N) * ld);
        const float *w = (tid < N) ? qw : kw;
        const float *bb = (tid < N) ? qb : kb;
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
    }
    __syncthreads();
    const float scale = rsqrtf((float)hd);
    for (int idx = tid; idx < N * N; idx += 256) {
        const int i = idx / N, j = idx - i * N;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(q[i * ld + d], k[j * ld + d], s);
        const bool allow = (i >= nv) || (j < nv);  // padded query rows attend to every key
        S[i * (N + 1) + j] = allow ? s * scale : -INFINITY;
    }
    __syncthreads();
    if (tid < N) {
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
    for (int idx = tid; idx < N * hd; idx += 256) {
        const int i = idx / hd, d = idx - i * hd;
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc = fmaf(S[i * (N + 1) + j], v[j * ld + d], acc);
        o[(rbase + i) * (int64_t)H + head * hd + d] = from_f32<T>(acc);
    }
}

// ------------------------------------------------------------------------------------------ AdaLN epilogue
// x += gate * (LN0(y) * (1 + scale) + shift)       (transformer.py:142-143; LN0 = no affine, eps 1e-5)
// y = sum of `nslab` split-K partial slabs (+ bias), summed in slab order (deterministic).
// One wave per token row; modulation rows come from the hoisted table mod[T][B+1][L][6H].
template <typename T>
__global__ __launch_bounds__(256) void ln_mod_res_kernel(const float *__restrict__ y, int nslab, int64_t slab_stride,
                                                          const float *__restrict__ bias, float *__restrict__ x32,
                                                          T *__restrict__ xa, const float *__restrict__ modtab,
                                                          const int *__restrict__ step_ptr, int layer, int sel, int B,
                                                          int N, int H, int L, int M2) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M2) return;
    const int lane = threadIdx.x & 63;
    const int s = *step_ptr;
    const int seq = row / N;
    const int ci = (seq < B) ? seq : B;  // unconditional pass shares one row
    const float *mod = modtab + (((int64_t)s * (B + 1) + ci) * L + layer) * (6 * (int64_t)H) + (int64_t)sel * 3 * H;
    constexpr int MAXE = 32;  // H <= 2048
    float v[MAXE];
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = lane + e * 64;
        float a = 0.f;
        if (h < H) {
            for (int z = 0; z < nslab; ++z) a += y[z * slab_stride + (int64_t)row * H + h];
            if (bias) a += bias[h];
            sum += a;
        }
        v[e] = a;
    }
    const float mean = wave_sum(sum) / (float)H;
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const float d = v[e] - mean;
        var += (lane + e * 64 < H) ? d * d : 0.f;
    }
    const float rstd = rsqrtf(wave_sum(var) / (float)H + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = lane + e * 64;
        if (h < H) {
            const float shift = mod[h], scale = mod[H + h], gate = mod[2 * H + h];
            const int64_t o = (int64_t)row * H + h;
            const float nx = x32[o] + gate * ((v[e] - mean) * rstd * (1.f + scale) + shift);
            x32[o] = nx;
            xa[o] = from_f32<T>(nx);
        }
    }
}

// ------------------------------------------------------------------------------------------ posterior + CFG + sampling
struct PostArgs {
    const float *out;    // [2][B][N][F] decoder output (fc2 + bias), before LN0/modulate
    const float *modo;   // [T][B+1][2F] output-layer modulation (shift | scale)
    int8_t *X;           // [B][N]      in/out
    int8_t *E;           // [B][N][N]   in/out
    const int *n_nodes;  // [B]
    const float *x_marg, *e_marg, *u_xe, *u_ex, *betas, *alphas_bar;
    const float *qx, *qe;          // injected Exp(1) noise or null
    const unsigned long long *seed_ptr;
    const int *step_ptr;
    int B, N, F, T;
    float guide;
    float *pX_out, *pE_out;  // optional taps
    float *logX, *logE;      // optional taps [2][B][N][16], [2][B][N][N][5]
    int update_state;
};

// LN0 + modulate of decoder output element f of row (p,b,i)
struct RowNorm {
    float mean, rstd;
};

__global__ __launch_bounds__(256) void posterior_sample_kernel(PostArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int b = blockIdx.x;
    const int N = a.N, F = a.F, B = a.B;
    const int tid = threadIdx.x;
    const int s = *a.step_ptr;
    const int nv = a.n_nodes[b];

    // ---- LDS carve
    float *shsc = reinterpret_cast<float *>(smraw);  // [2][2F]  (shift | scale) per pass
    float *stat = shsc + 4 * F;                      // [2][N][2]
    float *predX = stat + 4 * N;                     // [2][N][16]
    float *SE = predX + 2 * N * XD;                  // [2][N][8]  (5 class sums, [5] = total)
    float *PXE = SE + 2 * N * 8;                     // [2][N][8]  sum_a predX[a] u_xe[a][k]
    float *Sx = PXE + 2 * N * 8;                     // [N][16]
    float *Se = Sx + N * XD;                         // [N][8]
    float *cst = Se + N * 8;                         // x_marg[16] e_marg[5->8] u_xe[80] u_ex[80]
    int *sX = reinterpret_cast<int *>(cst + 16 + 8 + 80 + 80);  // [N]
    int8_t *sE = reinterpret_cast<int8_t *>(sX + N);            // [N][N]

    float *c_xm = cst, *c_em = cst + 16, *c_uxe = cst + 24, *c_uex = cst + 104;
    if (tid < 16) c_xm[tid] = a.x_marg[tid];
    if (tid < 5) c_em[tid] = a.e_marg[tid];
    if (tid < 80) {
        c_uxe[tid] = a.u_xe[tid];
        c_uex[tid] = a.u_ex[tid];
    }
    for (int i = tid; i < N; i += 256) sX[i] = a.X[(int64_t)b * N + i];
    for (int i = tid; i < N * N; i += 256) sE[i] = a.E[(int64_t)b * N * N + i];
    for (int i = tid; i < 2 * F; i += 256) {
        shsc[i] = a.modo[((int64_t)s * (B + 1) + b) * (2 * F) + i];
        shsc[2 * F + i] = a.modo[((int64_t)s * (B + 1) + B) * (2 * F) + i];
    }
    // ---- phase A: LayerNorm statistics of every decoder row (wave per row)
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int r = wave; r < 2 * N; r += 4) {
            const int p = r / N, i = r - p * N;
            const float *row = a.out + (((int64_t)p * B + b) * N + i) * F;
            float sm = 0.f;
            for (int f = lane; f < F; f += 64) sm += row[f];
            const float mean = wave_sum(sm) / (float)F;
            float vr = 0.f;
            for (int f = lane; f < F; f += 64) {
                const float d = row[f] - mean;
                vr += d * d;
            }
            vr = wave_sum(vr) / (float)F;
            if (lane == 0) {
                stat[r * 2] = mean;
                stat[r * 2 + 1] = rsqrtf(vr + 1e-5f);
            }
        }
    }
    __syncthreads();
    auto lnmod = [&](int p, int i, int f) -> float {
        const float v = a.out[(((int64_t)p * B + b) * N + i) * F + f];
        const float *ss = shsc + p * 2 * F;
        return (v - stat[(p * N + i) * 2]) * stat[(p * N + i) * 2 + 1] * (1.f + ss[F + f]) + ss[f];
    };
    // final (masked, symmetrised) bond logits of pair (i,j) for pass p   (transformer.py:170-185 + mask)
    auto edge_logits = [&](int p, int i, int j, float *l) {
        if (i >= nv || j >= nv || i == j) {
#pragma unroll
            for (int k = 0; k < ED; ++k) l[k] = 0.f;
            return;
        }
        const int eij = sE[i * N + j], eji = sE[j * N + i];
#pragma unroll
        for (int k = 0; k < ED; ++k) {
            const float bij = (eij == k ? 1.f : 0.f) + lnmod(p, i, XD + ED * j + k);
            const float bji = (eji == k ? 1.f : 0.f) + lnmod(p, j, XD + ED * i + k);
            l[k] = 0.5f * (bij + bji);
        }
    };
    auto softmax5 = [](float *l) {
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
#pragma unroll
        for (int k = 0; k < ED; ++k) l[k] *= inv;
    };

    // ---- phase B: pred_X = softmax(atom logits); SE = sum_j pred_E[i][j][:]; thread per (pass,node)
    if (tid < 2 * N) {
        const int p = tid / N, i = tid - p * N;
        float l[XD];
        const int xi = sX[i];
#pragma unroll
        for (int c = 0; c < XD; ++c) l[c] = (i < nv) ? ((xi == c ? 1.f : 0.f) + lnmod(p, i, c)) : 0.f;
        if (a.logX) {
#pragma unroll
            for (int c = 0; c < XD; ++c) a.logX[(((int64_t)p * B + b) * N + i) * XD + c] = l[c];
        }
        float mx = l[0];
#pragma unroll
        for (int c = 1; c < XD; ++c) mx = fmaxf(mx, l[c]);
        float sm = 0.f;
#pragma unroll
        for (int c = 0; c < XD; ++c) {
            l[c] = expf(l[c] - mx);
            sm += l[c];
        }
        const float inv = 1.f / sm;
        float pxe[ED] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < XD; ++c) {
            const float pv = l[c] * inv;
            predX[(p * N + i) * XD + c] = pv;
#pragma unroll
            for (int k = 0; k < ED; ++k) pxe[k] = fmaf(pv, c_uxe[c * ED + k], pxe[k]);
        }
#pragma unroll
        for (int k = 0; k < ED; ++k) PXE[(p * N + i) * 8 + k] = pxe[k];
        float se[ED] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < N; ++j) {
            float e5[ED];
            edge_logits(p, i, j, e5);
            if (a.logE) {
#pragma unroll
                for (int k = 0; k < ED; ++k) a.logE[((((int64_t)p * B + b) * N + i) * N + j) * ED + k] = e5[k];
            }
            softmax5(e5);
#pragma unroll
            for (int k = 0; k < ED; ++k) se[k] += e5[k];
        }
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < ED; ++k) {
            SE[(p * N + i) * 8 + k] = se[k];
            tot += se[k];
        }
        SE[(p * N + i) * 8 + 5] = tot;
    }
    // ---- phase C0: S[i,f] = sum_g X_t[i,g] u[f,g] in its structured form (diffusion_utils.py:296-305)
    if (tid >= 128 && tid < 128 + N) {
        const int i = tid - 128;
        const int xi = sX[i];
        float cnt[ED] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < N; ++j) {
            const int e = sE[i * N + j];
#pragma unroll
            for (int k = 0; k < ED; ++k) cnt[k] += (e == k) ? 1.f : 0.f;
        }
        const float xm = (xi >= 0) ? c_xm[xi] : 0.f;
#pragma unroll
        for (int c = 0; c < XD; ++c) {
            float t = xm;
#pragma unroll
            for (int k = 0; k < ED; ++k) t = fmaf(cnt[k], c_uxe[c * ED + k], t);
            Sx[i * XD + c] = t;
        }
        float em = 0.f;
#pragma unroll
        for (int k = 0; k < ED; ++k) em = fmaf(cnt[k], c_em[k], em);
#pragma unroll
        for (int k = 0; k < ED; ++k) Se[i * 8 + k] = ((xi >= 0) ? c_uex[k * XD + xi] : 0.f) + em;
    }
    __syncthreads();

    const float beta = a.betas[s + 1];
    const float ab_s = a.alphas_bar[s];
    const float ab_t = a.alphas_bar[s + 1];
    const bool guided = (a.guide != 1.0f);
    const unsigned long long seed = a.seed_ptr ? *a.seed_ptr : 0ull;

    // ---- phase C1: node posterior, guidance, sampling (thread per node)
    int newX = -1;
    if (tid < N) {
        const int i = tid;
        const int xi = sX[i];
        float pf[XD];
        if (i < nv) {
            float pc[2][XD];
            for (int p = 0; p < (guided ? 2 : 1); ++p) {
                float spx = 0.f;
#pragma unroll
                for (int c = 0; c < XD; ++c) spx += predX[(p * N + i) * XD + c];
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < XD; ++c) {
                    float r = c_xm[c] * spx;
#pragma unroll
                    for (int k = 0; k < ED; ++k) r = fmaf(SE[(p * N + i) * 8 + k], c_uex[k * XD + c], r);
                    const float right = ab_s * predX[(p * N + i) * XD + c] + (1.f - ab_s) * r;
                    const float xt = (xi == c) ? 1.f : 0.f;
                    const float left = (1.f - beta) * xt + beta * Sx[i * XD + c];
                    const float den = fmaxf(ab_t * xt + (1.f - ab_t) * Sx[i * XD + c], 1e-5f);
                    const float un = left * right / den;
                    pc[p][c] = un;
                    sum += un;
                }
                if (sum == 0.f) {
#pragma unroll
                    for (int c = 0; c < XD; ++c) pc[p][c] = 1e-5f;
                    sum = XD * 1e-5f;
                }
#pragma unroll
                for (int c = 0; c < XD; ++c) pc[p][c] /= sum;
            }
            if (guided) {
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < XD; ++c) {
                    const float u = pc[1][c];
                    pf[c] = u * powf(pc[0][c] / fmaxf(u, 1e-5f), a.guide);
                    sum += pf[c];
                }
                sum = fmaxf(sum, 1e-5f);
#pragma unroll
                for (int c = 0; c < XD; ++c) pf[c] /= sum;
            } else {
#pragma unroll
                for (int c = 0; c < XD; ++c) pf[c] = pc[0][c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < XD; ++c) pf[c] = 0.f;
        }
        if (a.pX_out) {
#pragma unroll
            for (int c = 0; c < XD; ++c) a.pX_out[((int64_t)b * N + i) * XD + c] = pf[c];
        }
        // sample_discrete_features (diffusion_utils.py:386-395)
        if (i < nv) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < XD; ++c) {
                pf[c] = fmaxf(pf[c], 1e-5f);
                sum += pf[c];
            }
            float best = -1.f;
            int arg = 0;
            float q[XD];
            if (a.qx) {
#pragma unroll
                for (int c = 0; c < XD; ++c) q[c] = a.qx[((int64_t)b * N + i) * XD + c];
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint4 r = philox4x32(make_uint4((uint32_t)(b * N + i), (uint32_t)s, (uint32_t)g, 0x58u),
                                               make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
                    q[g * 4 + 0] = exp1_from_bits(r.x);
                    q[g * 4 + 1] = exp1_from_bits(r.y);
                    q[g * 4 + 2] = exp1_from_bits(r.z);
                    q[g * 4 + 3] = exp1_from_bits(r.w);
                }
            }
#pragma unroll
            for (int c = 0; c < XD; ++c) {
                const float v = (pf[c] / sum) / q[c];
                if (v > best) {
                    best = v;
                    arg = c;
                }
            }
            newX = arg;
        }
    }

    // ---- phase C2: bond posterior, guidance, sampling for pairs i<j (strict upper triangle is what
    //      the reference keeps: diffusion_utils.py:409-411)
    if (a.pE_out) {
        for (int idx = tid; idx < N * N * ED; idx += 256) a.pE_out[(int64_t)b * N * N * ED + idx] = 0.f;
    }
    __syncthreads();  // all reads of sE for logits done before anybody overwrites global state; pE_out zeroed
    const int npairs = N * (N - 1) / 2;
    for (int pi = tid; pi < npairs; pi += 256) {
        // unrank pair index -> (i,j), i<j
        int i = 0, rem = pi;
        while (rem >= N - 1 - i) {
            rem -= N - 1 - i;
            ++i;
        }
        const int j = i + 1 + rem;
        int val = -1;
        if (i < nv && j < nv) {
            const int eij = sE[i * N + j];
            float pc[2][ED];
            for (int p = 0; p < (guided ? 2 : 1); ++p) {
                float e5[ED];
                edge_logits(p, i, j, e5);
                softmax5(e5);
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < ED; ++k) {
                    const float r = PXE[(p * N + i) * 8 + k] + c_em[k] * SE[(p * N + i) * 8 + 5];
                    const float right = ab_s * e5[k] + (1.f - ab_s) * r;
                    const float et = (eij == k) ? 1.f : 0.f;
                    const float left = (1.f - beta) * et + beta * Se[i * 8 + k];
                    const float den = fmaxf(ab_t * et + (1.f - ab_t) * Se[i * 8 + k], 1e-5f);
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
                    const uint4 r = philox4x32(make_uint4((uint32_t)((b * N + i) * N + j), (uint32_t)s, (uint32_t)g, 0x45u),
                                               make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
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
            a.E[((int64_t)b * N + i) * N + j] = (int8_t)val;
            a.E[((int64_t)b * N + j) * N + i] = (int8_t)val;
        }
    }
    if (a.update_state && tid < N) {
        a.X[(int64_t)b * N + tid] = (int8_t)newX;
        a.E[((int64_t)b * N + tid) * N + tid] = (tid < nv) ? (int8_t)0 : (int8_t)-1;
    }
}

static inline size_t posterior_lds_bytes(int N, int F) {
    size_t fl = 4 * (size_t)F + 4 * N + 2 * N * XD + 2 * N * 8 + 2 * N * 8 + N * XD + N * 8 + (16 + 8 + 80 + 80);
    return fl * 4 + (size_t)N * 4 + (size_t)N * N + 16;
}

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
                    const uint4 r = philox4x32(make_uint4((uint32_t)(b * N + i), (uint32_t)T, (uint32_t)g, 0x58u), key);
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
                    const uint4 r = philox4x32(make_uint4((uint32_t)((b * N + i) * N + j), (uint32_t)T, (uint32_t)g, 0x45u), key);
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
// Sinusoidal timestep features for every step s: t = (s+1)/T fractional (conditions.py:32-51).
template <typename T>
__global__ void tfreq_kernel(T *out, int Tsteps) {
    const int s = blockIdx.x;
    const int j = threadIdx.x;  // 0..127
    const float t = (float)(s + 1) / (float)Tsteps;
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
                                                     int8_t *__restrict__ ynan, int H) {
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
    for (int h = threadIdx.x; h < H; h += 256) mx = fmaxf(mx, fmaf(y, w[h], bb[h]));
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sm = 0.f;
    for (int h = threadIdx.x; h < H; h += 256) sm += expf(fmaf(y, w[h], bb[h]) - mx);
    sm = block_sum_256(sm, red);
    const float inv = 1.f / sm;
    for (int h = threadIdx.x; h < H; h += 256) z[h] = from_f32<T>(expf(fmaf(y, w[h], bb[h]) - mx) * inv);
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

__global__ void set_scalars_kernel(int *step_ptr, int s, unsigned long long *seed_ptr, unsigned long long seed) {
    *step_ptr = s;
    *seed_ptr = seed;
}
__global__ void advance_step_kernel(int *step_ptr) { *step_ptr = *step_ptr - 1; }

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
