// GIN encoder (GraphCLIP) and GIN predictor (template classifier) for MI355X (gfx950).
//
// Replaces reference GNNEncoder.forward + ProjectionHead (src/model/graph_encoder/model.py:124-205),
// GNNRetrosynthsizer.forward (src/model/graph_predictor/model.py:306-353), the softmax/top-k of
// GraphPredictor.sample_templates (:174-179) and CostMLP.forward (:385-391).
//
// Structure: the PyG scatter-add of the reference (gather x[src] -> +bond embedding -> GELU -> index_add
// into dst) becomes a CSR-by-destination pull: one 64-lane wave owns one destination node, walks its
// (degree <= ~6) incoming edges and accumulates in registers -- no atomics, deterministic summation order,
// feature rows read as coalesced 256-B segments.  Virtual-node max-pool / add-pool are segment reductions
// over the sorted `batch` vector (one workgroup per graph).  All Linears go through the shared MFMA GEMM.
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace ll {

// The reference classes take any sizes (graph_encoder/model.py:87-112, graph_predictor/model.py:231-278: num_layer, hidden_size, text_input_size,
// out_dim come from the checkpoint's config); the kernels want row pitches that are multiples of 64.  As in the GraphDiT engine
// (graphdit.hip: DitDims) the engine keeps a zero-padded INTERNAL copy of the weights when the checkpoint's widths are not of that form:
//   Hp = hidden rounded up to 64 (the MLPs' inner width is 4 Hp), Dp = text_dim rounded up to 64.
// Zero weight rows / columns, biases, LayerNorm affine parameters and adapter rows (shift = scale = gate = 0) keep every padded column of
// every activation an exact zero (GELU(0) = 0; max / sum pooling of zeros is zero); the LayerNorm statistics take the true width.
struct GinDims {
    int Ht, Hp, Dt, Dp;
    bool padded;
};
static GinDims gin_dims(const LLGinConfig &c) {
    GinDims d;
    d.Ht = c.hidden; d.Hp = round_up(c.hidden, 64);
    d.Dt = c.kind == 1 ? c.text_dim : 0; d.Dp = round_up(d.Dt, 64);
    d.padded = d.Hp != d.Ht || d.Dp != d.Dt;
    return d;
}

struct GinParam {
    std::string name;
    int64_t numel, offset;          // checkpoint tensor in the caller's arena (f32 elements)
    int rows, cols;                 // its shape as [rows][cols]
    int64_t inumel, ioffset;        // internal (padded) tensor; == numel / offset when nothing is padded
    int irows, icols;
    PadMap map;
};

static std::vector<GinParam> gin_layout(const LLGinConfig &c) {
    std::vector<GinParam> v;
    int64_t off = 0, ioff = 0;
    const GinDims d = gin_dims(c);
    auto add = [&](const std::string &n, int r, int cc, int ir, int ic, PadMap m = PadMap()) {
        const int64_t ne = (int64_t)r * cc, ine = (int64_t)ir * ic;
        v.push_back({n, ne, off, r, cc, ine, ioff, ir, ic, m});
        off += (ne + 63) / 64 * 64;
        ioff += (ine + 63) / 64 * 64;
    };
    const int H = d.Ht, Hp = d.Hp, D = d.Dt, Dp = d.Dp;
    PadMap rows3;  rows3.rg = H; rows3.rgp = Hp;       // (shift | scale | gate) chunks of H rows -> chunks of Hp rows
    auto mlp = [&](const std::string &p0, const std::string &p1, const std::string &p4) {
        add(p0 + "weight", 4 * H, H, 4 * Hp, Hp);
        add(p0 + "bias", 4 * H, 1, 4 * Hp, 1);
        add(p1 + "weight", 4 * H, 1, 4 * Hp, 1);
        add(p1 + "bias", 4 * H, 1, 4 * Hp, 1);
        add(p4 + "weight", H, 4 * H, Hp, 4 * Hp);
        add(p4 + "bias", H, 1, Hp, 1);
    };
    add("atom_encoder.weight", 118, H, 118, Hp);
    add("virtualnode_embedding.weight", 1, H, 1, Hp);
    if (c.kind == 1) add("text_dropping.weight", 1, D, 1, Dp);
    for (int i = 0; i < c.num_layer; ++i) {
        const std::string p = "convs." + std::to_string(i) + ".";
        add(p + "eps", 1, 1, 1, 1);
        mlp(p + "mlp.0.", p + "mlp.1.", p + "mlp.4.");
        add(p + "bond_encoder.weight", 5, H, 5, Hp);
        if (c.kind == 0) {
            add("norms." + std::to_string(i) + ".weight", H, 1, Hp, 1);
            add("norms." + std::to_string(i) + ".bias", H, 1, Hp, 1);
        } else {
            add("adapters." + std::to_string(i) + ".1.weight", 3 * H, D, 3 * Hp, Dp, rows3);
            add("adapters." + std::to_string(i) + ".1.bias", 3 * H, 1, 3 * Hp, 1, rows3);
        }
        if (i < c.num_layer - 1) {
            const std::string q = "mlp_virtualnode_list." + std::to_string(i) + ".";
            mlp(q + "0.", q + "1.", q + "4.");
        }
    }
    if (c.kind == 0) {  // ProjectionHead, keys prefixed "proj."
        add("proj.fc1.weight", H, H, Hp, Hp);
        add("proj.fc1.bias", H, 1, Hp, 1);
        add("proj.norm1.weight", H, 1, Hp, 1);
        add("proj.norm1.bias", H, 1, Hp, 1);
        add("proj.fc2.weight", H, H, Hp, Hp);
        add("proj.fc2.bias", H, 1, Hp, 1);
    } else {
        add("decoder.0.weight", 4 * H, H, 4 * Hp, Hp);
        add("decoder.0.bias", 4 * H, 1, 4 * Hp, 1);
        add("decoder.1.weight", 4 * H, 1, 4 * Hp, 1);
        add("decoder.1.bias", 4 * H, 1, 4 * Hp, 1);
        add("decoder.4.weight", c.out_dim, 4 * H, c.out_dim, 4 * Hp);
        add("decoder.4.bias", c.out_dim, 1, c.out_dim, 1);
    }
    return v;
}

// hidden <= 2048 bounds the per-row register footprint of the row kernels (the Python wrappers report it as a ValueError naming the limit)
static int gin_check(const LLGinConfig *c) {
    LL_CHECK(c != nullptr, "config is null");
    LL_CHECK(c->num_layer >= 2, "Number of GNN layers must be greater than 1.");
    LL_CHECK(c->hidden >= 1 && c->hidden <= 2048, "hidden=%d must be in [1,2048]", c->hidden);
    LL_CHECK(c->kind == 0 || c->kind == 1, "kind must be 0 (encoder) or 1 (predictor)");
    LL_CHECK(c->dtype == LL_F32 || c->dtype == LL_BF16, "unknown dtype %d", c->dtype);
    if (c->kind == 1) {
        LL_CHECK(c->out_dim >= 1, "predictor needs out_dim >= 1");
        LL_CHECK(c->text_dim >= 1 && c->text_dim <= 16384, "text_dim=%d must be in [1,16384]", c->text_dim);
    }
    return LL_OK;
}

// ------------------------------------------------------------------------------------------ kernels
// One wave per destination node:
//   h_in[v] = h[v] + vn[batch[v]]
//   z0[v]   = (1+eps) h_in[v] + sum_{e: dst(e)=v} GELU(h_in[src(e)] + bond_emb[attr(e)])
// (graph_encoder/model.py:133-134,167-173).  Edges of a graph never cross graphs, so vn[batch[src]] == vn[batch[v]].
template <typename T> __device__ __forceinline__ void gin_store4(T *p, float4 o);
template <> __device__ __forceinline__ void gin_store4<float>(float *p, float4 o) { *reinterpret_cast<float4 *>(p) = o; }
template <> __device__ __forceinline__ void gin_store4<bf16_t>(bf16_t *p, float4 o) {
    uint2 u;
    u.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
    u.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
    *reinterpret_cast<uint2 *>(p) = u;
}
__device__ __forceinline__ float4 gelu4(float4 v) { return make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)); }

// out[r] = act(LayerNorm_affine(in[r]))  -> operand dtype.  One wave per row, any C.
// Second row group (round 3: node rows and virtual-node rows of a GIN layer in one launch): rows [0, n1) take (w, b), rows [n1, r2) are
// padding and skipped, rows [r2, R) take (w2, b2); n1 = r2 = R for a plain call.
template <typename T>
__global__ __launch_bounds__(64) void rows_ln_act_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                          const float *__restrict__ b, T *__restrict__ out, int R,
                                                          int C, int gelu, int n1, int r2, const float *__restrict__ w2,
                                                          const float *__restrict__ b2, int Ct) {
    // C = row pitch, Ct <= C = the checkpoint's width: the statistics run over the Ct true columns; the padded ones are zeros going in
    // and (zero LayerNorm weight / bias) zeros coming out
    const int r = blockIdx.x;   // one wave = one row = one workgroup
    if (r >= R || (r >= n1 && r < r2)) return;
    if (r >= r2) { w = w2; b = b2; }
    const int lane = threadIdx.x;
    const float *x = in + (int64_t)r * C;
    constexpr int MAXE = 32;   // float4 chunks per lane: C <= 8192
    float4 v[MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        v[e] = k < C ? *reinterpret_cast<const float4 *>(x + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += v[e].x + v[e].y + v[e].z + v[e].w;
    }
    const float mean = wave_sum(s) / (float)Ct;
    float vr = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) vr += sq_dev4(v[e], mean, (lane + e * 64) * 4, Ct);
    const float rstd = rsqrtf(wave_sum(vr) / (float)Ct + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        if (k < C) {
            const float4 ww = *reinterpret_cast<const float4 *>(w + k);
            const float4 bb = *reinterpret_cast<const float4 *>(b + k);
            float4 y = make_float4((v[e].x - mean) * rstd * ww.x + bb.x, (v[e].y - mean) * rstd * ww.y + bb.y,
                                   (v[e].z - mean) * rstd * ww.z + bb.z, (v[e].w - mean) * rstd * ww.w + bb.w);
            if (gelu) y = gelu4(y);
            gin_store4<T>(out + (int64_t)r * C + k, y);
        }
    }
}

// The same row operation with WAVES waves per row and every load -- the row, the LayerNorm weight and bias -- issued up front
// (one memory round trip; a lane holds 3 * CH float4 loads in flight instead of up to 32 + a second round trip for w / b).  The
// one-wave kernel above ran the [512 x 2048] rows of the GIN MLPs at 11.7 us per launch (profiles/r2_gin_kernel_stats.csv).
template <typename T, int WAVES, int CH>
__global__ __launch_bounds__(64 * WAVES) void rows_ln_act_mw_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                                    const float *__restrict__ b, T *__restrict__ out, int R, int C,
                                                                    int gelu, int n1, int r2, const float *__restrict__ w2,
                                                                    const float *__restrict__ b2, int Ct) {
    __shared__ float part[2][WAVES];
    const int r = blockIdx.x;
    if (r >= R || (r >= n1 && r < r2)) return;
    if (r >= r2) { w = w2; b = b2; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *x = in + (int64_t)r * C;
    float4 v[CH], ww[CH], bb[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        const int k = (tid + e * 64 * WAVES) * 4;      // C == WAVES * 64 * 4 * CH
        v[e] = *reinterpret_cast<const float4 *>(x + k);
        ww[e] = *reinterpret_cast<const float4 *>(w + k);
        bb[e] = *reinterpret_cast<const float4 *>(b + k);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < CH; ++e) s += v[e].x + v[e].y + v[e].z + v[e].w;
    s = wave_sum(s);
    if (lane == 0) part[0][wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) tot += part[0][i];
    const float mean = tot / (float)Ct;
    float vr = 0.f;
#pragma unroll
    for (int e = 0; e < CH; ++e) vr += sq_dev4(v[e], mean, (tid + e * 64 * WAVES) * 4, Ct);
    vr = wave_sum(vr);
    if (lane == 0) part[1][wave] = vr;
    __syncthreads();
    float tv = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) tv += part[1][i];
    const float rstd = rsqrtf(tv / (float)Ct + 1e-5f);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        const int k = (tid + e * 64 * WAVES) * 4;
        float4 y = make_float4((v[e].x - mean) * rstd * ww[e].x + bb[e].x, (v[e].y - mean) * rstd * ww[e].y + bb[e].y,
                               (v[e].z - mean) * rstd * ww[e].z + bb[e].z, (v[e].w - mean) * rstd * ww[e].w + bb[e].w);
        if (gelu) y = gelu4(y);
        gin_store4<T>(out + (int64_t)r * C + k, y);
    }
}

template <typename T>
static void launch_rows_ln_act2(const float *in, const float *w, const float *b, T *out, int R, int C, int Ct, int gelu, int n1, int r2,
                                const float *w2, const float *b2, hipStream_t st) {
#define LL_RLA(W, CH, BLK) hipLaunchKernelGGL((rows_ln_act_mw_kernel<T, W, CH>), dim3(R), dim3(BLK), 0, st, in, w, b, out, R, C, gelu, n1, r2, w2, b2, Ct)
    if (C % 1024 == 0 && C / 1024 <= 4) {            // 4 waves per row, 1..4 float4 per lane
        switch (C / 1024) {
            case 1: LL_RLA(4, 1, 256); return;
            case 2: LL_RLA(4, 2, 256); return;
            case 3: LL_RLA(4, 3, 256); return;
            default: LL_RLA(4, 4, 256); return;
        }
    }
    if (C % 256 == 0 && C / 256 <= 3) {              // narrow rows (H = 64 .. 192 -> 4H = 256 .. 768): one wave, loads up front
        switch (C / 256) {
            case 1: LL_RLA(1, 1, 64); return;
            case 2: LL_RLA(1, 2, 64); return;
            default: LL_RLA(1, 3, 64); return;
        }
    }
#undef LL_RLA
    hipLaunchKernelGGL((rows_ln_act_kernel<T>), dim3(R), dim3(64), 0, st, in, w, b, out, R, C, gelu, n1, r2, w2, b2, Ct);
}
template <typename T>
static void launch_rows_ln_act(const float *in, const float *w, const float *b, T *out, int R, int C, int Ct, int gelu, hipStream_t st) {
    launch_rows_ln_act2<T>(in, w, b, out, R, C, Ct, gelu, R, R, nullptr, nullptr, st);
}

// Layer tail (gin_post2_kernel below).  Encoder: z = LN_affine(z); predictor: z = LN0(z) * (1 + scale) + shift, residual gated.
//   if not last: z = GELU(z);  h = (gate *) z + h_in         (model.py:137-145 / predictor :331-340)
// ---- round 3: the layer as FIVE launches (aggregate | Linear | LayerNorm+GELU | Linear | tail), the virtual-node MLP riding in the same
// launches as a second row group.  What made that possible: max_v (h[v] + vn[g]) == max_v h[v] + vn[g] exactly (rounding is monotone),
// so the pooled input of the virtual-node MLP depends only on the INPUTS of the aggregation launch and extra workgroups of that launch
// can produce it; everything downstream (Linear -> LN+GELU -> Linear -> += ) has the node MLP's shape with other weights.

// h0 = atom_encoder[x], vn0 = virtual-node embedding row, csilu = SiLU(c or text_dropping row), and the per-node neighbour records
// the L aggregation launches read: one launch instead of three.  Record of node v = 8 ints: up to 6 in-edges as (src << 3 | bond type),
// the in-degree, the graph id -- ONE 32-byte load gives an aggregation wave what took the chain rowptr -> src / attr -> rows before
// (molecular graphs: degree <= 4 almost always; edges beyond the sixth are read from the CSR arrays).
constexpr int ELL_SLOTS = 6;
template <typename T>
__global__ __launch_bounds__(256) void gin_prologue_kernel(const int *__restrict__ x, const float *__restrict__ emb,
                                                            const float *__restrict__ vemb, const float *__restrict__ c,
                                                            const float *__restrict__ drop, float *__restrict__ h, float *__restrict__ vn,
                                                            T *__restrict__ csilu, const int *__restrict__ rowptr, const int *__restrict__ src,
                                                            const int *__restrict__ attr, const int *__restrict__ batch, int *__restrict__ ell,
                                                            int n, int G, int H, int D, int Dt) {
    // H, D: row pitches of the engine (multiples of 64); Dt = width of the caller's `c` rows (the checkpoint's text_dim)
    const int H4 = H / 4, D4 = D / 4;
    const int64_t n4 = (int64_t)n * H4, g4 = (int64_t)G * H4, c4 = csilu ? (int64_t)G * D4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4 + g4 + c4 + n; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n4) {
            const int v = (int)(i / H4), k = (int)(i % H4) * 4;
            *reinterpret_cast<float4 *>(h + (int64_t)v * H + k) = *reinterpret_cast<const float4 *>(emb + (int64_t)x[v] * H + k);
        } else if (i < n4 + g4) {
            const int64_t j = i - n4;
            *reinterpret_cast<float4 *>(vn + j * 4) = *reinterpret_cast<const float4 *>(vemb + (j % H4) * 4);
        } else if (i < n4 + g4 + c4) {
            const int64_t j = i - n4 - g4;
            float4 v;
            if (!c) {
                v = *reinterpret_cast<const float4 *>(drop + (j % D4) * 4);
            } else if (Dt == D) {
                v = *reinterpret_cast<const float4 *>(c + j * 4);
            } else {      // padded text width: element-wise from the caller's narrower rows, zeros behind them (SiLU(0) = 0)
                const int64_t g = j / D4;
                const int k = (int)(j % D4) * 4;
                const float *cr = c + g * Dt;
                v = make_float4(k < Dt ? cr[k] : 0.f, k + 1 < Dt ? cr[k + 1] : 0.f, k + 2 < Dt ? cr[k + 2] : 0.f, k + 3 < Dt ? cr[k + 3] : 0.f);
            }
            gin_store4<T>(csilu + j * 4, make_float4(silu(v.x), silu(v.y), silu(v.z), silu(v.w)));
        } else {
            const int v = (int)(i - n4 - g4 - c4);
            const int e0 = rowptr[v], deg = rowptr[v + 1] - e0;
            int r[8];
#pragma unroll
            for (int j = 0; j < ELL_SLOTS; ++j) r[j] = j < deg ? ((src[e0 + j] << 3) | (attr[e0 + j] & 7)) : 0;
            r[6] = deg;
            r[7] = batch[v];
            *reinterpret_cast<int4 *>(ell + (int64_t)v * 8) = make_int4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<int4 *>(ell + (int64_t)v * 8 + 4) = make_int4(r[4], r[5], r[6], r[7]);
        }
    }
}

// Workgroups [0, nbn): 4 / NPW destination nodes, NPW waves (NPW * 256 features per pass) each: the node's neighbour record arrives with
// one load, then its own row, the virtual-node row and up to six neighbour + bond-embedding rows are all in flight together.
//   h_in[v] = h[v] + vn[batch[v]];   z0[v] = (1+eps) h_in[v] + sum_{e: dst(e)=v} GELU(h_in[src(e)] + bond_emb[attr(e)])
// (graph_encoder/model.py:133-134,167-173; edges never cross graphs, so vn[batch[src]] == vn[batch[v]]; edges in CSR order = the
// reference's summation order).  Workgroups [nbn, ...): (graph g, 256-feature chunk): pool[g] = max over the graph's nodes of h[v],
// + vn[g]  (== segment max of h_in, model.py:147-148) in operand dtype -- the A rows of the virtual-node MLP; the four waves take every
// fourth node with eight rows in flight, partial maxima meet in LDS.
template <typename T, int NPW>
__global__ __launch_bounds__(256) void gin_aggregate2_kernel(const float *__restrict__ h, const float *__restrict__ vn,
                                                             const int *__restrict__ ell, const int *__restrict__ rowptr,
                                                             const int *__restrict__ src, const int *__restrict__ attr,
                                                             const float *__restrict__ bond, const float *__restrict__ eps,
                                                             float *__restrict__ h_in, T *__restrict__ z0, const int *__restrict__ gptr,
                                                             T *__restrict__ pool, int n, int H, int nbn) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= nbn) {
        __shared__ float4 red[4][64];
        const int pb = blockIdx.x - nbn, chunks = (H + 255) / 256;
        const int g = pb / chunks, k = (pb % chunks) * 256 + lane * 4;
        const int v0 = gptr[g], v1 = gptr[g + 1];
        float4 acc = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (k < H) {
            for (int vb = v0 + wave; vb < v1; vb += 32) {
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    t[u] = (vb + 4 * u < v1) ? *reinterpret_cast<const float4 *>(h + (int64_t)(vb + 4 * u) * H + k)
                                             : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc.x = fmaxf(acc.x, t[u].x); acc.y = fmaxf(acc.y, t[u].y); acc.z = fmaxf(acc.z, t[u].z); acc.w = fmaxf(acc.w, t[u].w);
                }
            }
        }
        red[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && k < H) {
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 o = red[w][lane];
                acc.x = fmaxf(acc.x, o.x); acc.y = fmaxf(acc.y, o.y); acc.z = fmaxf(acc.z, o.z); acc.w = fmaxf(acc.w, o.w);
            }
            const float4 vk = *reinterpret_cast<const float4 *>(vn + (int64_t)g * H + k);
            gin_store4<T>(pool + (int64_t)g * H + k, make_float4(acc.x + vk.x, acc.y + vk.y, acc.z + vk.z, acc.w + vk.w));
        }
        return;
    }
    const int v = blockIdx.x * (4 / NPW) + wave / NPW, sub = wave % NPW;
    if (v >= n) return;
    const int rec = ell[(int64_t)v * 8 + (lane & 7)];
    const int deg = __builtin_amdgcn_readlane(rec, 6);
    const float *vr = vn + (int64_t)__builtin_amdgcn_readlane(rec, 7) * H;
    const float e1 = 1.f + eps[0];
    const int nd = deg < ELL_SLOTS ? deg : ELL_SLOTS;
    for (int k = (sub * 64 + lane) * 4; k < H; k += NPW * 256) {
        float4 hn[ELL_SLOTS], bn[ELL_SLOTS];
#pragma unroll
        for (int u = 0; u < ELL_SLOTS; ++u) {
            const int pk = __builtin_amdgcn_readlane(rec, u);      // uniform
            const bool ok = u < nd;
            hn[u] = ok ? *reinterpret_cast<const float4 *>(h + (int64_t)(pk >> 3) * H + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            bn[u] = ok ? *reinterpret_cast<const float4 *>(bond + (int64_t)(pk & 7) * H + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 vk = *reinterpret_cast<const float4 *>(vr + k);
        float4 hv = *reinterpret_cast<const float4 *>(h + (int64_t)v * H + k);
        hv.x += vk.x; hv.y += vk.y; hv.z += vk.z; hv.w += vk.w;
        float4 agg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < ELL_SLOTS; ++u) {
            if (u < nd) {
                const float4 m = gelu4(make_float4(hn[u].x + vk.x + bn[u].x, hn[u].y + vk.y + bn[u].y,
                                                   hn[u].z + vk.z + bn[u].z, hn[u].w + vk.w + bn[u].w));
                agg.x += m.x; agg.y += m.y; agg.z += m.z; agg.w += m.w;
            }
        }
        if (deg > ELL_SLOTS) {      // rare: the remaining in-edges from the CSR arrays, in order
            const int e0 = rowptr[v];
            for (int e = e0 + ELL_SLOTS; e < e0 + deg; ++e) {
                const float4 hs = *reinterpret_cast<const float4 *>(h + (int64_t)src[e] * H + k);
                const float4 bs = *reinterpret_cast<const float4 *>(bond + (int64_t)attr[e] * H + k);
                const float4 m = gelu4(make_float4(hs.x + vk.x + bs.x, hs.y + vk.y + bs.y, hs.z + vk.z + bs.z, hs.w + vk.w + bs.w));
                agg.x += m.x; agg.y += m.y; agg.z += m.z; agg.w += m.w;
            }
        }
        *reinterpret_cast<float4 *>(h_in + (int64_t)v * H + k) = hv;
        gin_store4<T>(z0 + (int64_t)v * H + k, make_float4(e1 * hv.x + agg.x, e1 * hv.y + agg.y, e1 * hv.z + agg.z, e1 * hv.w + agg.w));
    }
}

// Layer tail over the split-K slabs of the second Linear.  Workgroups [0, n): node v, z = sum of `ns` slabs (in order) + bias, then as
// gin_post_kernel (z is also stored to z_keep for the reverse sweep when given).  Workgroups [n, n + G): vn[g] += sum of the slabs of row
// vrow0 + g + vbias  (model.py:149-150).  `mod` rows have pitch modld (all layers' adapters come from one GEMM: [G][L * 3H]).
__global__ __launch_bounds__(64) void gin_post2_kernel(const float *__restrict__ slabs, int64_t slab_stride, int ns,
                                                        const float *__restrict__ bias, const float *__restrict__ h_in,
                                                        const float *__restrict__ lnw, const float *__restrict__ lnb,
                                                        const float *__restrict__ mod, int modld, const int *__restrict__ batch,
                                                        float *__restrict__ h, float *__restrict__ z_keep, int n, int H, int gelu,
                                                        float *__restrict__ vn, const float *__restrict__ vbias, int vrow0, int Ht) {
    // H = row pitch, Ht = the checkpoint's hidden_size (LayerNorm statistics); padded columns: zero in, zero out (zero affine / zero gate)
    const int lane = threadIdx.x;
    constexpr int MAXE = 8;   // H <= 2048
    if ((int)blockIdx.x >= n) {
        const int g = blockIdx.x - n;
        const float *x = slabs + (int64_t)(vrow0 + g) * H;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const int k = (lane + e * 64) * 4;
            if (k < H) {
                float4 a = *reinterpret_cast<const float4 *>(x + k);
                for (int q = 1; q < ns; ++q) {
                    const float4 o = *reinterpret_cast<const float4 *>(x + q * slab_stride + k);
                    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
                }
                const float4 bb = *reinterpret_cast<const float4 *>(vbias + k);
                float4 *dst = reinterpret_cast<float4 *>(vn + (int64_t)g * H + k);
                const float4 old = *dst;
                *dst = make_float4(old.x + (a.x + bb.x), old.y + (a.y + bb.y), old.z + (a.z + bb.z), old.w + (a.w + bb.w));
            }
        }
        return;
    }
    const int v = blockIdx.x;
    const float *x = slabs + (int64_t)v * H;
    float4 t[MAXE], hi[MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        t[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        hi[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < H) {
            float4 a = *reinterpret_cast<const float4 *>(x + k);
            for (int q = 1; q < ns; ++q) {
                const float4 o = *reinterpret_cast<const float4 *>(x + q * slab_stride + k);
                a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            }
            const float4 bb = *reinterpret_cast<const float4 *>(bias + k);
            t[e] = make_float4(a.x + bb.x, a.y + bb.y, a.z + bb.z, a.w + bb.w);
            hi[e] = *reinterpret_cast<const float4 *>(h_in + (int64_t)v * H + k);
            if (z_keep) *reinterpret_cast<float4 *>(z_keep + (int64_t)v * H + k) = t[e];
        }
        s += t[e].x + t[e].y + t[e].z + t[e].w;
    }
    const float mean = wave_sum(s) / (float)Ht;
    float vr = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) vr += sq_dev4(t[e], mean, (lane + e * 64) * 4, Ht);
    const float rstd = rsqrtf(wave_sum(vr) / (float)Ht + 1e-5f);
    const float *m = mod ? mod + (int64_t)batch[v] * modld : nullptr;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        if (k < H) {
            float4 y = make_float4((t[e].x - mean) * rstd, (t[e].y - mean) * rstd, (t[e].z - mean) * rstd, (t[e].w - mean) * rstd);
            float4 gate = make_float4(1.f, 1.f, 1.f, 1.f);
            if (m) {
                const float4 sh = *reinterpret_cast<const float4 *>(m + k);
                const float4 sc = *reinterpret_cast<const float4 *>(m + H + k);
                gate = *reinterpret_cast<const float4 *>(m + 2 * H + k);
                y = make_float4(y.x * (1.f + sc.x) + sh.x, y.y * (1.f + sc.y) + sh.y, y.z * (1.f + sc.z) + sh.z, y.w * (1.f + sc.w) + sh.w);
            } else {
                const float4 ww = *reinterpret_cast<const float4 *>(lnw + k);
                const float4 bb = *reinterpret_cast<const float4 *>(lnb + k);
                y = make_float4(y.x * ww.x + bb.x, y.y * ww.y + bb.y, y.z * ww.z + bb.z, y.w * ww.w + bb.w);
            }
            if (gelu) y = gelu4(y);
            *reinterpret_cast<float4 *>(h + (int64_t)v * H + k) =
                make_float4(gate.x * y.x + hi[e].x, gate.y * y.y + hi[e].y, gate.z * y.z + hi[e].z, gate.w * y.w + hi[e].w);
        }
    }
}

// The same tail with one wave per 256-column chunk of the row (H == WAVES * 256): a lane holds ns + 5 loads in flight instead of
// (ns + 5) * H / 256, the two row reductions go through LDS (wave order: deterministic).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gin_post2_mw_kernel(const float *__restrict__ slabs, int64_t slab_stride, int ns,
                                                                  const float *__restrict__ bias, const float *__restrict__ h_in,
                                                                  const float *__restrict__ lnw, const float *__restrict__ lnb,
                                                                  const float *__restrict__ mod, int modld, const int *__restrict__ batch,
                                                                  float *__restrict__ h, float *__restrict__ z_keep, int n, int H, int gelu,
                                                                  float *__restrict__ vn, const float *__restrict__ vbias, int vrow0, int Ht) {
    __shared__ float part[2][WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = tid * 4;
    if ((int)blockIdx.x >= n) {
        const int g = blockIdx.x - n;
        const float *x = slabs + (int64_t)(vrow0 + g) * H;
        float4 a = *reinterpret_cast<const float4 *>(x + k);
        for (int q = 1; q < ns; ++q) {
            const float4 o = *reinterpret_cast<const float4 *>(x + q * slab_stride + k);
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        const float4 bb = *reinterpret_cast<const float4 *>(vbias + k);
        float4 *dst = reinterpret_cast<float4 *>(vn + (int64_t)g * H + k);
        const float4 old = *dst;
        *dst = make_float4(old.x + (a.x + bb.x), old.y + (a.y + bb.y), old.z + (a.z + bb.z), old.w + (a.w + bb.w));
        return;
    }
    const int v = blockIdx.x;
    const float *x = slabs + (int64_t)v * H;
    const float *m = mod ? mod + (int64_t)batch[v] * modld : nullptr;
    float4 a = *reinterpret_cast<const float4 *>(x + k);
    const float4 hi = *reinterpret_cast<const float4 *>(h_in + (int64_t)v * H + k);
    const float4 bb = *reinterpret_cast<const float4 *>(bias + k);
    float4 p0, p1, p2 = make_float4(1.f, 1.f, 1.f, 1.f);      // shift | scale | gate, or LayerNorm weight | bias
    if (m) {
        p0 = *reinterpret_cast<const float4 *>(m + k);
        p1 = *reinterpret_cast<const float4 *>(m + H + k);
        p2 = *reinterpret_cast<const float4 *>(m + 2 * H + k);
    } else {
        p0 = *reinterpret_cast<const float4 *>(lnw + k);
        p1 = *reinterpret_cast<const float4 *>(lnb + k);
    }
    for (int q = 1; q < ns; ++q) {
        const float4 o = *reinterpret_cast<const float4 *>(x + q * slab_stride + k);
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    const float4 t = make_float4(a.x + bb.x, a.y + bb.y, a.z + bb.z, a.w + bb.w);
    if (z_keep) *reinterpret_cast<float4 *>(z_keep + (int64_t)v * H + k) = t;
    float s = wave_sum(t.x + t.y + t.z + t.w);
    if (lane == 0) part[0][wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) tot += part[0][i];
    const float mean = tot / (float)Ht;
    const float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
    float vr = wave_sum(sq_dev4(t, mean, k, Ht));
    if (lane == 0) part[1][wave] = vr;
    __syncthreads();
    float tv = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) tv += part[1][i];
    const float rstd = rsqrtf(tv / (float)Ht + 1e-5f);
    float4 y = make_float4(d0 * rstd, d1 * rstd, d2 * rstd, d3 * rstd);
    if (m) y = make_float4(y.x * (1.f + p1.x) + p0.x, y.y * (1.f + p1.y) + p0.y, y.z * (1.f + p1.z) + p0.z, y.w * (1.f + p1.w) + p0.w);
    else y = make_float4(y.x * p0.x + p1.x, y.y * p0.y + p1.y, y.z * p0.z + p1.z, y.w * p0.w + p1.w);
    if (gelu) y = gelu4(y);
    *reinterpret_cast<float4 *>(h + (int64_t)v * H + k) = make_float4(p2.x * y.x + hi.x, p2.y * y.y + hi.y, p2.z * y.z + hi.z, p2.w * y.w + hi.w);
}

// Readout: pool[g] = sum over the graph's nodes of h[v] (global_add_pool, model.py:152), f32 and operand dtype.  Workgroup = (graph,
// 256-feature chunk); wave w sums nodes v0 + w, v0 + w + 4, ... with eight rows in flight, the four partial sums are added in wave
// order (deterministic).  The one-workgroup-per-graph form walked the 32 nodes of a molecule in four dependent rounds on half-idle
// waves (9.5 us at 16 graphs x 512 features).
template <typename T>
__global__ __launch_bounds__(256) void gin_pool_add_kernel(const float *__restrict__ h, const int *__restrict__ gptr,
                                                            float *__restrict__ out32, T *__restrict__ outa, int H) {
    __shared__ float4 red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunks = (H + 255) / 256;
    const int g = blockIdx.x / chunks, k = (blockIdx.x % chunks) * 256 + lane * 4;
    const int v0 = gptr[g], v1 = gptr[g + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < H) {
        for (int vb = v0 + wave; vb < v1; vb += 32) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = (vb + 4 * u < v1) ? *reinterpret_cast<const float4 *>(h + (int64_t)(vb + 4 * u) * H + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && k < H) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 o = red[w][lane];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        if (out32) *reinterpret_cast<float4 *>(out32 + (int64_t)g * H + k) = acc;
        if (outa) gin_store4<T>(outa + (int64_t)g * H + k, acc);
    }
}

// segment max / sum over the sorted batch vector: one workgroup per graph, threads over features.
template <typename T, bool MAX>
__global__ __launch_bounds__(256) void segment_pool_kernel(const float *__restrict__ h, const int *__restrict__ gptr,
                                                            float *__restrict__ out32, T *__restrict__ outa, int H) {
    const int g = blockIdx.x;
    const int v0 = gptr[g], v1 = gptr[g + 1];
    for (int k = (blockIdx.y * 256 + threadIdx.x) * 4; k < H; k += gridDim.y * 1024) {
        float4 acc = MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int vb = v0; vb < v1; vb += 8) {   // 8 node rows in flight
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = (vb + u < v1) ? *reinterpret_cast<const float4 *>(h + (int64_t)(vb + u) * H + k)
                                     : (MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MAX) { acc.x = fmaxf(acc.x, t[u].x); acc.y = fmaxf(acc.y, t[u].y); acc.z = fmaxf(acc.z, t[u].z); acc.w = fmaxf(acc.w, t[u].w); }
                else { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
            }
        }
        if (out32) *reinterpret_cast<float4 *>(out32 + (int64_t)g * H + k) = acc;
        if (outa) gin_store4<T>(outa + (int64_t)g * H + k, acc);
    }
}

__global__ void add_rows_kernel(float *__restrict__ dst, const float *__restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
// rows of C columns: `in` has row pitch ldi (the engine's padded width), `out` is the caller's dense [R][C]
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float *__restrict__ in, float *__restrict__ out, int R, int C, int ldi) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int k = lane; k < C; k += 64) {
        const float v = in[(int64_t)r * ldi + k];
        s += v * v;
    }
    const float inv = 1.f / sqrtf(wave_sum(s));
    for (int k = lane; k < C; k += 64) out[(int64_t)r * C + k] = in[(int64_t)r * ldi + k] * inv;
}

// ------------------------------------------------------------------------------------------ softmax + top-k
// One 1024-thread workgroup per row.  Pass 0: online softmax statistics.  Passes 1-4: MSB-first radix select of
// the k-th largest key on the order-preserving uint32 image of the float.  Then gather (> kth, then == kth),
// rank-sort the k survivors in LDS, emit probabilities.
__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// Bin choice of one radix pass by a whole wave (lanes 0..63 of the workgroup): the largest bin b with
// sum_{j > b} hist[j] < need <= sum_{j >= b} hist[j] and acc = sum_{j > b} hist[j]  (b = 0 when the histogram holds fewer than `need`).
__device__ __forceinline__ void radix_pick(const unsigned int *hist, unsigned need, int &bsel, unsigned &acc_above) {
    const int lane = threadIdx.x & 63;
    const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const unsigned tot = h0 + h1 + h2 + h3;
    unsigned suf = tot;      // inclusive suffix sum over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += t;
    }
    const unsigned above = suf - tot;
    const unsigned long long hit = __ballot(above < need && need <= suf);
    const int L = hit ? (63 - __clzll((long long)hit)) : 0;
    unsigned acc = __shfl(above, L, 64);
    const unsigned b3 = __shfl(h3, L, 64), b2 = __shfl(h2, L, 64), b1 = __shfl(h1, L, 64);
    int bb = 3;
    if (acc + b3 < need) {
        acc += b3;
        bb = 2;
        if (acc + b2 < need) {
            acc += b2;
            bb = 1;
            if (acc + b1 < need) {
                acc += b1;
                bb = 0;
            }
        }
    }
    bsel = 4 * L + bb;
    acc_above = acc;
}
// Row scan helper: 4 x float4 per thread in flight (the row is L2-resident after the first pass; a scalar strided
// loop would serialise ~D/1024 dependent round trips per pass).  f(value, index) is called for every element.
template <typename F>
__device__ __forceinline__ void topk_scan_row(const float *__restrict__ x, int D, int tid, F f) {
    const int D4 = (D % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? D / 4 : 0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (int i0 = tid; i0 < D4; i0 += 4 * 1024) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + u * 1024 < D4) ? x4[i0 + u * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 1024;
            if (i < D4) {
                f(v[u].x, 4 * i);
                f(v[u].y, 4 * i + 1);
                f(v[u].z, 4 * i + 2);
                f(v[u].w, 4 * i + 3);
            }
        }
    }
    for (int i = 4 * D4 + tid; i < D; i += 1024) f(x[i], i);
}

// MERGE = false: one workgroup scans a whole row of logits (small rows).  MERGE = true: second stage of the chunked form below --
// the "row" is the list of per-chunk candidates [chunks][k] (values sorted per chunk by value desc / index asc, chunks in index
// order, so POSITION order among equal values is INDEX order), `ids` their template indices, `stats` the per-chunk (max, sum exp).
template <bool MERGE>
__global__ __launch_bounds__(1024) void softmax_topk_kernel(const float *__restrict__ logits, int D, int k,
                                                             float *__restrict__ probs, int *__restrict__ idx,
                                                             const int *__restrict__ ids, const float *__restrict__ stats, int chunks) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int whist[16][256];   // one histogram per wave: the top radix byte (sign + exponent) puts most keys
                                              // into a handful of bins, so a single LDS histogram serialises on atomics
    __shared__ float redm[16], reds[16];
    __shared__ unsigned int s_prefix, s_need, s_cnt, s_ties;
    __shared__ int tie_idx[64];
    __shared__ float selv[64];
    __shared__ int seli[64];
    const float *x = logits + (int64_t)blockIdx.x * D;
    if (MERGE) ids += (int64_t)blockIdx.x * D;
    const int tid = threadIdx.x;
    float gm = -INFINITY, gs = 0.f;
    if (MERGE) {
        // softmax statistics of the whole row from the chunk statistics, summed in chunk order by every thread (deterministic)
        const float *st = stats + (int64_t)blockIdx.x * chunks * 2;
        for (int c = 0; c < chunks; ++c) gm = fmaxf(gm, st[2 * c]);
        for (int c = 0; c < chunks; ++c) gs += (st[2 * c] == -INFINITY) ? 0.f : st[2 * c + 1] * expf(st[2 * c] - gm);
    } else {
        // pass 0: max and sum(exp)  (online softmax)
        float m = -INFINITY, s = 0.f;
        topk_scan_row(x, D, tid, [&](float v, int) {
            if (v > m) {
                s = s * expf(m - v) + 1.f;
                m = v;
            } else {
                s += expf(v - m);
            }
        });
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
            const float mn = fmaxf(m, m2);
            s = ((m == -INFINITY) ? 0.f : s * expf(m - mn)) + ((m2 == -INFINITY) ? 0.f : s2 * expf(m2 - mn));
            m = mn;
        }
        if ((tid & 63) == 0) {
            redm[tid >> 6] = m;
            reds[tid >> 6] = s;
        }
        __syncthreads();
        for (int w = 0; w < 16; ++w) gm = fmaxf(gm, redm[w]);
        for (int w = 0; w < 16; ++w) gs += (redm[w] == -INFINITY) ? 0.f : reds[w] * expf(redm[w] - gm);
    }
    // radix select, most significant byte first
    if (tid == 0) {
        s_prefix = 0;
        s_need = (unsigned)k;
    }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        unsigned int *myh = whist[tid >> 6];
        topk_scan_row(x, D, tid, [&](float v, int) {
            const unsigned key = f2key(v);
            if ((key & pmask) == prefix) atomicAdd(&myh[(key >> shift) & 255u], 1u);
        });
        __syncthreads();
        if (tid < 256) {
            unsigned t = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) t += whist[w][tid];
            hist[tid] = t;
        }
        __syncthreads();
        if (tid < 64) {      // wave 0 picks the bin (a serial scan of 256 LDS bins by one thread cost ~5 us per pass)
            int bsel;
            unsigned acc;
            const unsigned need = s_need;
            radix_pick(hist, need, bsel, acc);
            if (tid == 0) {
                s_need = need - acc;
                s_prefix = prefix | ((unsigned)bsel << shift);
            }
        }
        __syncthreads();
    }
    const unsigned kth = s_prefix;   // key of the k-th largest element
    const unsigned need_eq = s_need;  // how many elements equal to kth belong to the top-k
    if (tid == 0) {
        s_cnt = 0;
        s_ties = 0;
    }
    __syncthreads();
    // gather: everything above the threshold, plus (normally exactly one) element equal to it
    topk_scan_row(x, D, tid, [&](float v, int i) {
        const unsigned key = f2key(v);
        if (key > kth) {
            const unsigned p = atomicAdd(&s_cnt, 1u);
            selv[p] = v;
            seli[p] = MERGE ? ids[i] : i;
        } else if (key == kth) {
            const unsigned p = atomicAdd(&s_ties, 1u);
            if (p < 64) tie_idx[p] = i;
        }
    });
    __syncthreads();
    // ties at the threshold: keep the `need_eq` lowest indices (deterministic); > 64 exact ties -> serial scan
    if (tid == 0) {
        unsigned p = s_cnt;
        if (s_ties <= 64) {
            const unsigned nt = s_ties;
            for (unsigned a2 = 0; a2 < nt; ++a2)
                for (unsigned b2 = a2 + 1; b2 < nt; ++b2)
                    if (tie_idx[b2] < tie_idx[a2]) { const int t2 = tie_idx[a2]; tie_idx[a2] = tie_idx[b2]; tie_idx[b2] = t2; }
            for (unsigned a2 = 0; a2 < need_eq && a2 < nt; ++a2) {
                selv[p] = x[tie_idx[a2]];
                seli[p] = MERGE ? ids[tie_idx[a2]] : tie_idx[a2];
                ++p;
            }
        } else {
            unsigned taken = 0;
            for (int i = 0; i < D && taken < need_eq; ++i)
                if (f2key(x[i]) == kth) {
                    selv[p] = x[i];
                    seli[p] = MERGE ? ids[i] : i;
                    ++p;
                    ++taken;
                }
        }
    }
    __syncthreads();
    if (tid < k) {
        const float v = selv[tid];
        const int id = seli[tid];
        int rank = 0;
        for (int j = 0; j < k; ++j) rank += (selv[j] > v) || (selv[j] == v && seli[j] < id);
        probs[(int64_t)blockIdx.x * k + rank] = expf(v - gm) / gs;
        idx[(int64_t)blockIdx.x * k + rank] = id;
    }
}

// ---- chunked form.  Stage 1: one 256-thread workgroup per (chunk of <= 4096 templates, row): a row of 180 576 logits is 45
// chunks, so 16 rows are 720 workgroups over the whole chip instead of 16 (round 1: 330 us = 35 GB/s on 16 CUs).  The chunk is
// loaded ONCE into registers (16 values per thread); softmax statistics and the selection run on the register copy.  Output per
// chunk: its top-k as (value, template index) sorted by value desc / index asc, and (max, sum exp(v - max)).  Stage 2: one
// workgroup per row selects among the chunks x k candidates and emits probabilities.  An element of the row's top-k under the
// total order (value desc, index asc) is necessarily in its chunk's top-k, so the result -- set, order, ties to the lowest
// template index -- is exactly that of the one-workgroup form.
//
// Selection without a radix pass over everything (LDS atomics on the float exponent byte serialise: 20 us per stage): the k-th
// largest of the 256 per-thread maxima is a lower bound of the k-th largest value, and only ~k elements lie at or above it, so
// those few are gathered and rank-sorted.  Too many candidates (thousands of exact ties) -> k rounds of workgroup-wide argmax.
constexpr int TOPK_VPT = 16, TOPK_CHUNK = 256 * TOPK_VPT, TOPK_CAP = 1024;
struct __attribute__((aligned(16))) TopkShared {
    unsigned long long cand[TOPK_CAP];      // (key << 32) | (0x7fffffff - template index): larger = earlier in the total order
    unsigned tmax[256];
    unsigned cnt, thr;
    unsigned long long red[4];
};
__device__ __forceinline__ float key2f(unsigned key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}
__device__ __forceinline__ unsigned long long topk_pack(unsigned key, int idx) {
    return ((unsigned long long)key << 32) | (unsigned)(0x7fffffff - idx);
}
// key[e] / valid bit e: the VPT values of this thread (order-preserving uint images, all > 0 for non-NaN floats); idx_of(e): template
// index.  out(rank, key, index) is called exactly once for every rank in [0, k), k <= 64 and <= the number of valid values.
//   1. threshold: the k-th largest of 64 group maxima (group = the four threads lane, 64 + lane, ...) is a lower bound of the
//      k-th largest value; one wave ranks the 64 maxima against each other with readlane broadcasts;
//   2. the ~2k values at or above it are gathered (LDS counter) and rank-sorted as packed 64-bit (key, index) words;
//   3. more than TOPK_CAP candidates (thousands of exact ties): k rounds of workgroup-wide argmax instead.
template <int VPT, typename IdxF, typename OutF>
__device__ __forceinline__ void topk_select_block256(const unsigned (&key)[VPT], unsigned valid, IdxF idx_of, int k, TopkShared &sh, OutF out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned tm = 0;
#pragma unroll
    for (int e = 0; e < VPT; ++e) tm = ((valid >> e) & 1u) && key[e] > tm ? key[e] : tm;
    sh.tmax[tid] = tm;
    if (tid == 0) sh.cnt = 0;
    __syncthreads();
    if (wave == 0) {
        const unsigned g = max(max(sh.tmax[lane], sh.tmax[64 + lane]), max(sh.tmax[128 + lane], sh.tmax[192 + lane]));
        int gt = 0, ge = 0;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)g, j);
            gt += o > g ? 1 : 0;
            ge += o >= g ? 1 : 0;
        }
        // lanes that hold the k-th largest group maximum (they all hold the same value); g == 0: a group without a valid value
        const unsigned long long hit = __ballot(g != 0 && gt < k && k <= ge);
        const int src = hit ? (__ffsll((long long)hit) - 1) : 0;
        const unsigned t = (unsigned)__shfl((int)g, src, 64);
        if (lane == 0) sh.thr = hit ? t : 0u;      // fewer than k non-empty groups: every valid value is a candidate
    }
    __syncthreads();
    const unsigned thr = sh.thr;
#pragma unroll
    for (int e = 0; e < VPT; ++e)
        if (((valid >> e) & 1u) && key[e] >= thr) {
            const unsigned p = atomicAdd(&sh.cnt, 1u);
            if (p < (unsigned)TOPK_CAP) sh.cand[p] = topk_pack(key[e], idx_of(e));
        }
    __syncthreads();
    const unsigned C = sh.cnt;
    if (C <= (unsigned)TOPK_CAP) {      // workgroup-uniform
        for (unsigned i = tid; i < C; i += 256) {
            const unsigned long long mine = sh.cand[i];
            int r = 0;
#pragma unroll 8
            for (unsigned j = 0; j < C; ++j) r += sh.cand[j] > mine ? 1 : 0;
            if (r < k) out(r, (unsigned)(mine >> 32), 0x7fffffff - (int)(unsigned)(mine & 0xffffffffu));
        }
        return;
    }
    // k rounds of workgroup-wide argmax of the packed words
    unsigned taken = ~valid;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = 0;
        int be = -1;
#pragma unroll
        for (int e = 0; e < VPT; ++e)
            if (!((taken >> e) & 1u)) {
                const unsigned long long pk = topk_pack(key[e], idx_of(e));
                if (pk > best) { best = pk; be = e; }
            }
        unsigned long long wb = best;
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(wb & 0xffffffffu), o, 64);
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(wb >> 32), o, 64);
            const unsigned long long ob = ((unsigned long long)hi << 32) | lo;
            wb = ob > wb ? ob : wb;
        }
        __syncthreads();
        if (lane == 0) sh.red[wave] = wb;
        __syncthreads();
        unsigned long long g = sh.red[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) g = sh.red[w] > g ? sh.red[w] : g;
        if (be >= 0 && best == g) taken |= 1u << be;      // template indices are unique: exactly one owner
        if (tid == 0) out(r, (unsigned)(g >> 32), 0x7fffffff - (int)(unsigned)(g & 0xffffffffu));
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void topk_chunk_kernel(const float *__restrict__ logits, int D, int k, int chunk_len,
                                                         float *__restrict__ ws_val, int *__restrict__ ws_idx,
                                                         float *__restrict__ ws_stat) {
    __shared__ TopkShared sh;
    __shared__ float redm[4], reds[4];
    const int c = blockIdx.x, row = blockIdx.y, chunks = gridDim.x, tid = threadIdx.x;
    const float *x = logits + (int64_t)row * D;
    const int beg = c * chunk_len;
    const int n = min(D, beg + chunk_len) - beg;            // >= 1 by construction of the grid
    auto pos = [&](int e) { return VEC ? (((e >> 2) * 256 + tid) * 4 + (e & 3)) : (e * 256 + tid); };
    float v[TOPK_VPT];
    if (VEC) {       // D % 4 == 0, 16-byte aligned rows, chunk_len % 4 == 0: whole float4s are valid or not
#pragma unroll
        for (int u = 0; u < TOPK_VPT / 4; ++u) {
            const int p = (u * 256 + tid) * 4;
            const float4 t = p < n ? *reinterpret_cast<const float4 *>(x + beg + p) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < TOPK_VPT; ++e) v[e] = pos(e) < n ? x[beg + pos(e)] : -INFINITY;
    }
    unsigned valid = 0;
#pragma unroll
    for (int e = 0; e < TOPK_VPT; ++e) valid |= (pos(e) < n ? 1u : 0u) << e;
    // ---- softmax statistics of the chunk
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < TOPK_VPT; ++e) m = fmaxf(m, v[e]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) redm[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < TOPK_VPT; ++e) sm += (((valid >> e) & 1u) && m != -INFINITY) ? expf(v[e] - m) : 0.f;
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if ((tid & 63) == 0) reds[tid >> 6] = sm;
    __syncthreads();
    if (tid == 0) {
        ws_stat[((int64_t)row * chunks + c) * 2] = m;
        ws_stat[((int64_t)row * chunks + c) * 2 + 1] = (reds[0] + reds[1]) + (reds[2] + reds[3]);
    }
    float *ov = ws_val + ((int64_t)row * chunks + c) * k;
    int *oi = ws_idx + ((int64_t)row * chunks + c) * k;
    const int kk = min(k, n);        // candidates this chunk hands on (a short last chunk may hold fewer than k templates)
    if (tid >= kk && tid < k) {
        ov[tid] = -INFINITY;         // padding that can never be selected (k <= D)
        oi[tid] = 0x7fffffff;
    }
    unsigned key[TOPK_VPT];
#pragma unroll
    for (int e = 0; e < TOPK_VPT; ++e) key[e] = f2key(v[e]);
    topk_select_block256<TOPK_VPT>(key, valid, [&](int e) { return beg + pos(e); }, kk, sh,
                                   [&](int r, unsigned kq, int id) { ov[r] = key2f(kq); oi[r] = id; });
}

// Stage 2 for chunks * k <= 4096 candidates (D <= 262 144 at k = 64): the candidates in registers, strided over the threads so
// that every thread sees a mix of chunks; softmax statistics from the chunk statistics in chunk order (deterministic).
__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ ws_val, const int *__restrict__ ws_idx,
                                                         const float *__restrict__ ws_stat, int chunks, int k,
                                                         float *__restrict__ probs, int *__restrict__ idx) {
    __shared__ TopkShared sh;
    const int row = blockIdx.x, tid = threadIdx.x;
    const int total = chunks * k;
    const float *cv = ws_val + (int64_t)row * total;
    const int *ci = ws_idx + (int64_t)row * total;
    unsigned key[TOPK_VPT];
    int id[TOPK_VPT];
    unsigned valid = 0;
#pragma unroll
    for (int e = 0; e < TOPK_VPT; ++e) {
        const int p = e * 256 + tid;
        id[e] = p < total ? ci[p] : 0x7fffffff;
        key[e] = f2key(p < total ? cv[p] : -INFINITY);
        valid |= (id[e] != 0x7fffffff ? 1u : 0u) << e;
    }
    // softmax statistics of the row from the chunk statistics: every wave reduces them redundantly with the same fixed shuffle tree
    // (deterministic); a serial loop over the chunks was 90 dependent scalar-load round trips (~20 us)
    const float *st = ws_stat + (int64_t)row * chunks * 2;
    const int lane = tid & 63;
    float gm = -INFINITY, gs = 0.f;
    for (int c0 = 0; c0 < chunks; c0 += 64) gm = fmaxf(gm, c0 + lane < chunks ? st[2 * (c0 + lane)] : -INFINITY);
    for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
    for (int c0 = 0; c0 < chunks; c0 += 64) {
        const int c = c0 + lane;
        const float mc = c < chunks ? st[2 * c] : -INFINITY;
        gs += (mc == -INFINITY) ? 0.f : st[2 * c + 1] * expf(mc - gm);
    }
    for (int o = 32; o > 0; o >>= 1) gs += __shfl_xor(gs, o, 64);
    topk_select_block256<TOPK_VPT>(key, valid, [&](int e) { return id[e]; }, k, sh, [&](int r, unsigned kq, int t) {
        probs[(int64_t)row * k + r] = expf(key2f(kq) - gm) / gs;
        idx[(int64_t)row * k + r] = t;
    });
}

// PyG-style graph batch (int64 x [n], edge_index [2,E], edge_attr [E], sorted batch [n]) -> the int32 CSR-by-destination arrays of
// ll_gin_forward in ONE launch (the ATen route -- stable argsort, two bincounts, two cumsums, casts -- was ~15 launches and two host
// syncs per GIN forward, in a path whose figure of merit is microseconds per graph-layer).  One 1024-thread workgroup: degree
// counts by atomics, workgroup scans, an unordered scatter of edge ids and a per-node insertion sort of each (short) segment by
// edge id, which restores the reference's per-destination summation order (= a stable sort by destination).
__device__ __forceinline__ void wg_exclusive_scan_inplace(int *a, int n, int *part /*[16]*/) {
    // a[0..n) counts -> a[i] = sum of the counts before i.  Thread t owns a contiguous chunk; chunk sums are scanned inside each
    // wave with shuffles and across the 16 waves through LDS (two workgroup barriers in all).
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, per = (n + 1023) / 1024;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    __syncthreads();                    // part[] of a previous scan has been consumed
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w)
        if (w < wave) base += part[w];
    int run = base + inc - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = a[i];
        a[i] = run;
        run += c;
    }
    __syncthreads();
}

// Error flag of both kernels (device memory, or pinned HOST memory so that the caller's stream never carries a read-back): gathered in
// LDS and stored ONCE, by the last instruction of the kernel -- a host that pre-set the word to -1 reads -1 until the conversion has
// run, then 0 (ok) | 1 unsorted or out-of-range `batch` | 2 edge endpoint outside the batch | 3 atom / bond type outside the embedding
// tables.  Whatever the flag says, the arrays written are SAFE to run the GIN kernels on (ids clamped into range, bad edges dropped).
template <bool IN_LDS>
__global__ __launch_bounds__(1024) void graph_csr_kernel(const int64_t *__restrict__ x, const int64_t *__restrict__ edge_index,
                                                          const int64_t *__restrict__ edge_attr, const int64_t *__restrict__ batch, int n,
                                                          int E, int G, int *__restrict__ x32, int *__restrict__ rowptr, int *__restrict__ src,
                                                          int *__restrict__ attr, int *__restrict__ b32, int *__restrict__ gptr,
                                                          int *__restrict__ cursor, int *__restrict__ err) {
    extern __shared__ int sm_csr[];
    __shared__ int part[16];
    __shared__ int s_err;
    const int tid = threadIdx.x;
    const int64_t *esrc = edge_index, *edst = edge_index + E;
    // IN_LDS (molecule batches: 2n + G + E + 2 ints fit the dynamic LDS): degree counts, cursors, graph counts and the scattered edge ids
    // all live in LDS -- LDS atomics and no global round trip between the phases; otherwise the caller's global arrays are the workspace
    int *rp = IN_LDS ? sm_csr : rowptr;
    int *cur = IN_LDS ? rp + n + 1 : cursor;
    int *gp = IN_LDS ? cur + n : gptr;
    int *eid = IN_LDS ? gp + G + 1 : src;
    if (tid == 0) s_err = 0;
    for (int i = tid; i <= n; i += 1024) rp[i] = 0;
    for (int i = tid; i <= G; i += 1024) gp[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const long long xv = x[i];
        if (xv < 0 || xv >= 118) s_err = 3;
        x32[i] = (int)(xv < 0 ? 0 : xv >= 118 ? 117 : xv);
        const long long b = batch[i];
        b32[i] = (int)(b < 0 ? 0 : b >= G ? G - 1 : b);
        cur[i] = 0;
        if (b < 0 || b >= G || (i > 0 && batch[i - 1] > b)) s_err = 1;      // unsorted / out-of-range `batch`
        if (b >= 0 && b < G) atomicAdd(&gp[b], 1);
    }
    for (int e = tid; e < E; e += 1024) {
        const long long d = edst[e], sidx = esrc[e];
        if (d < 0 || d >= n || sidx < 0 || sidx >= n) s_err = 2;
        else atomicAdd(&rp[d], 1);
    }
    __syncthreads();
    wg_exclusive_scan_inplace(rp, n + 1, part);       // rp[n] (count 0) becomes the total
    wg_exclusive_scan_inplace(gp, G + 1, part);
    for (int e = tid; e < E; e += 1024) {
        const long long d = edst[e];
        if (d >= 0 && d < n && esrc[e] >= 0 && esrc[e] < n) eid[rp[d] + atomicAdd(&cur[d], 1)] = e;
    }
    __syncthreads();
    for (int v = tid; v < n; v += 1024) {
        const int lo = rp[v], hi = rp[v + 1];
        for (int i = lo + 1; i < hi; ++i) {               // insertion sort by edge id: molecular degrees are <= ~6
            const int key = eid[i];
            int j = i - 1;
            while (j >= lo && eid[j] > key) {
                eid[j + 1] = eid[j];
                --j;
            }
            eid[j + 1] = key;
        }
        for (int i = lo; i < hi; ++i) {
            const int e = eid[i];
            src[i] = (int)esrc[e];
            const long long av = edge_attr[e];
            if (av < 0 || av >= 5) s_err = 3;
            attr[i] = (int)(av < 0 ? 0 : av >= 5 ? 4 : av);
        }
    }
    for (int i = rp[n] + tid; i < E; i += 1024) {         // slots of dropped edges (beyond rowptr[n], never read by a GIN kernel): defined values
        src[i] = 0;
        attr[i] = 0;
    }
    if (IN_LDS) {
        for (int i = tid; i <= n; i += 1024) rowptr[i] = rp[i];
        for (int i = tid; i <= G; i += 1024) gptr[i] = gp[i];
    }
    __syncthreads();
    if (tid == 0) *err = s_err;
}

// CostMLP: softplus(w3 . relu(W0 fp + b0) + b3), W0 [128][2048].  One workgroup per fingerprint.
__global__ __launch_bounds__(256) void cost_mlp_kernel(const float *__restrict__ w, const float *__restrict__ fps,
                                                        float *__restrict__ out) {
    __shared__ float part[4];
    const float *W0 = w, *b0 = w + 128 * 2048, *w3 = b0 + 128, *b3 = w3 + 128;
    const float *fp = fps + (int64_t)blockIdx.x * 2048;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int hdn = wave; hdn < 128; hdn += 4) {
        float d = 0.f;
        for (int k = lane; k < 2048; k += 64) d = fmaf(W0[(int64_t)hdn * 2048 + k], fp[k], d);
        d = wave_sum(d) + b0[hdn];
        acc += fmaxf(d, 0.f) * w3[hdn];
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float y = part[0] + part[1] + part[2] + part[3] + b3[0];
        out[blockIdx.x] = logf(1.f + expf(y));
    }
}

// ------------------------------------------------------------------------------------------ engine
struct GBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes) return LL_OK;
        debug_guard_check(p, bytes, "before growing");
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        // zeros; LL_DEBUG_POISON=1: 0xFF (NaN / -1) instead -- a kernel that counts on the initial zeros, and would therefore break once a
        // larger earlier call has left other data behind, fails the test suite at once -- plus guard bytes behind the payload (common.h)
        LL_TRY(debug_alloc(&p, n));
        bytes = n;
        return LL_OK;
    }
    void release() {
        debug_guard_check(p, bytes, "at release");
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct GinEngine {
    LLGinConfig cfg;
    GinDims d;                   // checkpoint widths and the engine's padded pitches
    std::vector<GinParam> layout;
    const float *w32 = nullptr;  // f32 master arena in the INTERNAL layout: the caller's arena itself when no width needs padding, else `wpad`
    GBuf wpad;                   // zero-padded f32 copy of the caller's arena (only when d.padded)
    GBuf wop;
    GBuf adcat, adbcat;          // predictor: the L adapter Linears concatenated along N ([L * 3H, text_dim] operand dtype, [L * 3H] f32 bias): one GEMM
    GBuf h, h_in, a0, t1, t1a, zs, vn, pool32, poola, mod, csilu, head1, head1a, head2, ell;
    // ---- training (ll_gin_forward keep=1 + ll_gin_backward_c): per-layer activations, transposed weights, gradients
    bool keep = false;
    int kept_n = -1, kept_G = -1;
    std::vector<GBuf> sv_hin, sv_t1, sv_z;      // sv_t1[l]: rows [0, n) node MLP, rows [round_up(n, 64), +G) virtual-node MLP (pre-LayerNorm)
    GBuf sv_head1, wT, adT;
    GBuf g_dh, g_dz, g_dt1a, g_dt1, g_dz0, g_c3, g_dmod, g_dmoda, g_dvn, g_dvna, g_dvt1a, g_dvt1, g_dpool, g_tmpG, g_dhead1a,
        g_dhead1, g_dlog, g_slabs, g_dcs;
    const float *pf(const std::string &n) const {
        for (auto &p : layout)
            if (p.name == n) return w32 + p.ioffset;
        return nullptr;
    }
    const void *pw(const std::string &n) const {
        for (auto &p : layout)
            if (p.name == n) return cfg.dtype == LL_BF16 ? (const void *)(wop.as<bf16_t>() + p.ioffset) : (const void *)(w32 + p.ioffset);
        return nullptr;
    }
    // the layer's parameters, resolved once (pf / pw compare strings; the forward makes ~15 lookups per layer)
    struct LayerW {
        const float *bond, *eps, *b0, *ln_w, *ln_b, *b4, *norm_w, *norm_b, *vb0, *vln_w, *vln_b, *vb4;
        const void *w0, *w4, *vw0, *vw4;
    };
    std::vector<LayerW> lw;
    const float *atom_emb = nullptr, *vn_emb = nullptr, *text_drop = nullptr;
    const void *w_h0 = nullptr, *w_h2 = nullptr;            // head: proj.fc1 / fc2 or decoder.0 / decoder.4
    const float *b_h0 = nullptr, *b_h2 = nullptr, *hln_w = nullptr, *hln_b = nullptr;
    void cache_params() {
        const int L = cfg.num_layer;
        lw.assign(L, LayerW{});
        for (int l = 0; l < L; ++l) {
            const std::string p = "convs." + std::to_string(l) + ".", q = "mlp_virtualnode_list." + std::to_string(l) + ".";
            LayerW &w = lw[l];
            w.bond = pf(p + "bond_encoder.weight"); w.eps = pf(p + "eps");
            w.w0 = pw(p + "mlp.0.weight"); w.b0 = pf(p + "mlp.0.bias"); w.ln_w = pf(p + "mlp.1.weight"); w.ln_b = pf(p + "mlp.1.bias");
            w.w4 = pw(p + "mlp.4.weight"); w.b4 = pf(p + "mlp.4.bias");
            w.norm_w = cfg.kind == 0 ? pf("norms." + std::to_string(l) + ".weight") : nullptr;
            w.norm_b = cfg.kind == 0 ? pf("norms." + std::to_string(l) + ".bias") : nullptr;
            if (l < L - 1) {
                w.vw0 = pw(q + "0.weight"); w.vb0 = pf(q + "0.bias"); w.vln_w = pf(q + "1.weight"); w.vln_b = pf(q + "1.bias");
                w.vw4 = pw(q + "4.weight"); w.vb4 = pf(q + "4.bias");
            }
        }
        atom_emb = pf("atom_encoder.weight"); vn_emb = pf("virtualnode_embedding.weight");
        text_drop = cfg.kind == 1 ? pf("text_dropping.weight") : nullptr;
        if (cfg.kind == 0) {
            w_h0 = pw("proj.fc1.weight"); b_h0 = pf("proj.fc1.bias"); hln_w = pf("proj.norm1.weight"); hln_b = pf("proj.norm1.bias");
            w_h2 = pw("proj.fc2.weight"); b_h2 = pf("proj.fc2.bias");
        } else {
            w_h0 = pw("decoder.0.weight"); b_h0 = pf("decoder.0.bias"); hln_w = pf("decoder.1.weight"); hln_b = pf("decoder.1.bias");
            w_h2 = pw("decoder.4.weight"); b_h2 = pf("decoder.4.bias");
        }
    }
};

// split-K of the layer's second Linear ([M, 4H] x [H, 4H]^T, 64 x 64 tiles): until the launch has ~256 workgroups, at most 4 slabs
static int gin_k_splits(int M, int H) {
    const long tiles = (long)cdiv(M, 64) * cdiv(H, 64);
    int s = 1;
    while (s < 4 && tiles * s < 256 && (4 * H / (2 * s)) % 64 == 0 && 4 * H / (2 * s) >= 256) s *= 2;
    return s;
}

// One layer = five launches (round 3; was ten):  aggregate (+ the graph max-pool workgroups) | Linear(H, 4H) | LayerNorm + GELU |
// Linear(4H, H) as split-K slabs | tail; rows [0, n) are the nodes, rows [n64, n64 + G) the graphs' virtual-node MLP (other weights, same
// shapes) -- one grouped GEMM instead of two launches, no separate pool / add launches.  Whole predictor forward at L = 5: 2 + 25 + 4
// launches (was ~60) + the CSR conversion and the two top-k launches.
template <typename T>
static int gin_forward_t(GinEngine *e, const int *x, const int *rowptr, const int *src, const int *attr,
                         const int *batch, const int *gptr, int n, int ne, int G, const float *c, float *out,
                         float *pooled, hipStream_t st) {
    const LLGinConfig &cf = e->cfg;
    const int H = e->d.Hp, Ht = e->d.Ht, D = e->d.Dp, L = cf.num_layer, dt = cf.dtype, es = sizeof(T);      // H, D: row pitches (multiples of 64)
    const int n64 = round_up(n, 64), MR = n64 + round_up(G, 64), Gp = round_up(G, 128);
    const int ks = gin_k_splits(n64 + G, H);
    const int64_t zstride = (int64_t)MR * H;
    LL_TRY(e->h.ensure((size_t)n64 * H * 4));
    LL_TRY(e->h_in.ensure((size_t)n64 * H * 4));
    LL_TRY(e->a0.ensure((size_t)MR * H * es));
    LL_TRY(e->t1.ensure((size_t)MR * 4 * H * 4));
    LL_TRY(e->t1a.ensure((size_t)MR * 4 * H * es));
    LL_TRY(e->zs.ensure((size_t)4 * MR * H * 4));
    LL_TRY(e->vn.ensure((size_t)Gp * H * 4));
    LL_TRY(e->pool32.ensure((size_t)Gp * H * 4));
    LL_TRY(e->poola.ensure((size_t)Gp * H * es));
    LL_TRY(e->ell.ensure((size_t)n64 * 8 * 4));
    const dim3 blk(256);
    const int modld = L * 3 * H;
    T *csilu = nullptr;
    if (cf.kind == 1) {
        LL_TRY(e->csilu.ensure((size_t)Gp * D * es));
        LL_TRY(e->mod.ensure((size_t)Gp * modld * 4));
        csilu = e->csilu.as<T>();
    }
    {
        const int64_t items = ((int64_t)n * H + (int64_t)G * H + (csilu ? (int64_t)G * D : 0)) / 4 + n;
        hipLaunchKernelGGL((gin_prologue_kernel<T>), dim3((unsigned)std::min<int64_t>((items + 255) / 256, 2048)), blk, 0, st,
                           x, e->atom_emb, e->vn_emb, c, e->text_drop, e->h.as<float>(), e->vn.as<float>(), csilu, rowptr, src, attr, batch,
                           e->ell.as<int>(), n, G, H, D, e->d.Dt);
        LL_LAUNCH_CHECK();
    }
    if (cf.kind == 1)   // (shift, scale, gate) of every layer: ONE GEMM over the N-concatenated adapters -> mod [G][L * 3H]
        LL_TRY(linear_launch(dt, e->csilu.p, D, e->adcat.p, D, e->adbcat.as<float>(), e->mod.p, modld, G, modld, D, 0, 1, st));
    const int npw = H >= 1024 ? 4 : H >= 512 ? 2 : 1;      // waves per node of the aggregation launch: one pass over the row
    const int nbn = cdiv(n, 4 / npw), chunks = cdiv(H, 256);
    const int post_waves = (H % 256 == 0 && H / 256 <= 8 && ((H / 256) & (H / 256 - 1)) == 0) ? H / 256 : 0;
    for (int l = 0; l < L; ++l) {
        const GinEngine::LayerW &w = e->lw[l];
        const bool last = (l == L - 1);
        float *h_in = e->h_in.as<float>(), *t1 = e->t1.as<float>(), *z_keep = nullptr;
        if (e->keep) {      // the reverse sweep reads these: write them in place instead of copying them out afterwards
            LL_TRY(e->sv_hin[l].ensure((size_t)n64 * H * 4));
            LL_TRY(e->sv_t1[l].ensure((size_t)MR * 4 * H * 4));
            LL_TRY(e->sv_z[l].ensure((size_t)n64 * H * 4));
            h_in = e->sv_hin[l].as<float>(); t1 = e->sv_t1[l].as<float>(); z_keep = e->sv_z[l].as<float>();
        }
        const int M = last ? n : n64 + G;      // rows of the layer's GEMMs
#define LL_AGG(NPW)                                                                                                                \
    hipLaunchKernelGGL((gin_aggregate2_kernel<T, NPW>), dim3(nbn + (last ? 0 : G * chunks)), blk, 0, st, e->h.as<float>(), e->vn.as<float>(), \
                       e->ell.as<int>(), rowptr, src, attr, w.bond, w.eps, h_in, e->a0.as<T>(), gptr, e->a0.as<T>() + (size_t)n64 * H, n, H, nbn)
        if (npw == 4) LL_AGG(4); else if (npw == 2) LL_AGG(2); else LL_AGG(1);
#undef LL_AGG
        LL_LAUNCH_CHECK();
        LL_TRY(linear_grouped2_launch(dt, e->a0.p, H, w.w0, last ? nullptr : w.vw0, H, w.b0, w.vb0, t1, 4 * H, M, n64, 4 * H, H, 1, 0, 0, 1, st));
        launch_rows_ln_act2<T>(t1, w.ln_w, w.ln_b, e->t1a.as<T>(), M, 4 * H, 4 * Ht, 1, n, last ? M : n64, w.vln_w, w.vln_b, st);
        LL_LAUNCH_CHECK();
        LL_TRY(linear_grouped2_launch(dt, e->t1a.p, 4 * H, w.w4, last ? nullptr : w.vw4, 4 * H, nullptr, nullptr, e->zs.p, H, M, n64, H, 4 * H, ks,
                                      zstride, 0, 1, st));
        const float *mod = cf.kind == 1 ? e->mod.as<float>() + (size_t)l * 3 * H : nullptr;
#define LL_POST(KERNEL, BLK)                                                                                                       \
    hipLaunchKernelGGL(KERNEL, dim3(n + (last ? 0 : G)), dim3(BLK), 0, st, e->zs.as<float>(), zstride, ks, w.b4, h_in, w.norm_w, w.norm_b, \
                       mod, modld, batch, e->h.as<float>(), z_keep, n, H, last ? 0 : 1, e->vn.as<float>(), w.vb4, n64, Ht)
        switch (post_waves) {
            case 1: LL_POST(gin_post2_mw_kernel<1>, 64); break;
            case 2: LL_POST(gin_post2_mw_kernel<2>, 128); break;
            case 4: LL_POST(gin_post2_mw_kernel<4>, 256); break;
            case 8: LL_POST(gin_post2_mw_kernel<8>, 512); break;
            default: LL_POST(gin_post2_kernel, 64); break;
        }
#undef LL_POST
        LL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((gin_pool_add_kernel<T>), dim3(G * chunks), blk, 0, st, e->h.as<float>(), gptr, e->pool32.as<float>(), e->poola.as<T>(), H);
    LL_LAUNCH_CHECK();
    if (pooled) LL_HIP(hipMemcpy2DAsync(pooled, (size_t)Ht * 4, e->pool32.p, (size_t)H * 4, (size_t)Ht * 4, G, hipMemcpyDeviceToDevice, st));
    if (cf.kind == 0) {  // ProjectionHead + L2 normalise (model.py:37-41,198-205)
        LL_TRY(e->head1.ensure((size_t)Gp * H * 4));
        LL_TRY(e->head1a.ensure((size_t)Gp * H * es));
        LL_TRY(e->head2.ensure((size_t)Gp * H * 4));
        LL_TRY(linear_launch(dt, e->poola.p, H, e->w_h0, H, e->b_h0, e->head1.p, H, G, H, H, 0, 1, st));
        launch_rows_ln_act<T>(e->head1.as<float>(), e->hln_w, e->hln_b, e->head1a.as<T>(), G, H, Ht, 1, st);
        LL_LAUNCH_CHECK();
        LL_TRY(linear_launch(dt, e->head1a.p, H, e->w_h2, H, e->b_h2, e->head2.p, H, G, H, H, 0, 1, st));
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cdiv(G, 4)), blk, 0, st, e->head2.as<float>(), out, G, Ht, H);
        LL_LAUNCH_CHECK();
    } else {  // decoder: Linear(H,4H) -> LN -> GELU -> Linear(4H,out_dim)  (model.py:272-278)
        LL_TRY(e->head1a.ensure((size_t)Gp * 4 * H * es));
        float *head1 = nullptr;
        if (e->keep) {
            LL_TRY(e->sv_head1.ensure((size_t)Gp * 4 * H * 4));
            head1 = e->sv_head1.as<float>();
            e->kept_n = n;
            e->kept_G = G;
        } else {
            LL_TRY(e->head1.ensure((size_t)Gp * 4 * H * 4));
            head1 = e->head1.as<float>();
        }
        LL_TRY(linear_launch(dt, e->poola.p, H, e->w_h0, H, e->b_h0, head1, 4 * H, G, 4 * H, H, 0, 1, st));
        launch_rows_ln_act<T>(head1, e->hln_w, e->hln_b, e->head1a.as<T>(), G, 4 * H, 4 * Ht, 1, st);
        LL_LAUNCH_CHECK();
        // template head [G, 4H] x [out_dim, 4H]^T (740 MB of bf16 weights at out_dim = 180 576): 3..16 graphs stream it through the
        // 16-row MFMA Linear (line-contiguous loads, wave-private LDS transpose), anything else through the GEMM dispatch
        if (dt == LL_BF16 && G >= 3 && G <= 16 && (4 * H) % 32 == 0)
            LL_TRY(linear_rows16_launch(e->head1a.p, 4 * H, e->w_h2, 4 * H, e->b_h2, nullptr, 0.f, nullptr, 0, out, cf.out_dim, G,
                                        cf.out_dim, 4 * H, 0, 1, st));
        else
            LL_TRY(linear_launch(dt, e->head1a.p, 4 * H, e->w_h2, 4 * H, e->b_h2, out, cf.out_dim, G, cf.out_dim, 4 * H, 0, 1, st));
    }
    return LL_OK;
}


// ================================================================================================ training: d logits -> d c
// Backward of the predictor forward w.r.t. the text condition c only (weights are frozen in Llamole's SFT: the retro
// cross-entropy reaches the LLM through c = lm_to_graph_predictor(hidden), reference modeling_llamole.py:385-419).
// It is still a full reverse sweep over the node features: c enters every layer through (shift, scale, gate).

__device__ __forceinline__ float gelu_grad(float u) {
    return 0.5f * (1.f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * expf(-0.5f * u * u);
}

// dt = d/dt [ GELU(LN_affine(t)) ] . g   (rows of C <= 8192), one wave per row; dt in operand dtype (next GEMM's A)
template <typename T>
__global__ __launch_bounds__(64) void ln_gelu_bwd_kernel(const float *__restrict__ t, const float *__restrict__ w,
                                                          const float *__restrict__ b, const float *__restrict__ g,
                                                          T *__restrict__ dt, int R, int C, int Ct) {
    // C = row pitch, Ct = the checkpoint's width: statistics over the Ct true columns; the padded columns (zero LayerNorm weight: no
    // gradient passes through them) get dt = 0
    const int r = blockIdx.x;
    if (r >= R) return;
    const int lane = threadIdx.x;
    const float *x = t + (int64_t)r * C, *gr = g + (int64_t)r * C;
    constexpr int MAXE = 32;
    float4 v[MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        v[e] = k < C ? *reinterpret_cast<const float4 *>(x + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += v[e].x + v[e].y + v[e].z + v[e].w;
    }
    const float mean = wave_sum(s) / (float)Ct;
    float vr = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) vr += sq_dev4(v[e], mean, (lane + e * 64) * 4, Ct);
    const float rstd = rsqrtf(wave_sum(vr) / (float)Ct + 1e-5f);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {     // v[e] <- xhat; accumulate mean(dxhat), mean(dxhat * xhat)
        const int k = (lane + e * 64) * 4;
        if (k < C) {
            const float4 ww = *reinterpret_cast<const float4 *>(w + k), bb = *reinterpret_cast<const float4 *>(b + k);
            const float4 gg = *reinterpret_cast<const float4 *>(gr + k);
            float xh[4] = {(v[e].x - mean) * rstd, (v[e].y - mean) * rstd, (v[e].z - mean) * rstd, (v[e].w - mean) * rstd};
            const float wv[4] = {ww.x, ww.y, ww.z, ww.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dx = gv[q] * gelu_grad(xh[q] * wv[q] + bv[q]) * wv[q];
                m1 += dx;
                m2 += dx * xh[q];
            }
            v[e] = make_float4(xh[0], xh[1], xh[2], xh[3]);
        }
    }
    m1 = wave_sum(m1) / (float)Ct;
    m2 = wave_sum(m2) / (float)Ct;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        if (k < C) {
            const float4 ww = *reinterpret_cast<const float4 *>(w + k), bb = *reinterpret_cast<const float4 *>(b + k);
            const float4 gg = *reinterpret_cast<const float4 *>(gr + k);
            const float xh[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
            const float wv[4] = {ww.x, ww.y, ww.z, ww.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w};
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dx = gv[q] * gelu_grad(xh[q] * wv[q] + bv[q]) * wv[q];
                o[q] = k + q < Ct ? rstd * (dx - m1 - xh[q] * m2) : 0.f;
            }
            gin_store4<T>(dt + (int64_t)r * C + k, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// Layer tail backward (predictor): h = gate * act(LN0(z) (1 + scale) + shift) + h_in.  One wave per node.
//   dz (operand dtype), per-node contributions to d shift / d scale / d gate (summed per graph afterwards); dh is left in
//   place as the running d h_in (the residual path is the identity).
template <typename T>
__global__ __launch_bounds__(64) void gin_post_bwd_kernel(const float *__restrict__ z, const float *__restrict__ mod, int modld,
                                                           const int *__restrict__ batch, const float *__restrict__ dh,
                                                           T *__restrict__ dz, float *__restrict__ c3 /*[3][n][H]*/, int n,
                                                           int H, int gelu, int Ht) {
    const int v = blockIdx.x;
    if (v >= n) return;
    const int lane = threadIdx.x;
    const float *x = z + (int64_t)v * H;
    constexpr int MAXE = 8;
    float4 t[MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        t[e] = k < H ? *reinterpret_cast<const float4 *>(x + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += t[e].x + t[e].y + t[e].z + t[e].w;
    }
    const float mean = wave_sum(s) / (float)Ht;
    float vr = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) vr += sq_dev4(t[e], mean, (lane + e * 64) * 4, Ht);
    const float rstd = rsqrtf(wave_sum(vr) / (float)Ht + 1e-5f);
    const float *m = mod + (int64_t)batch[v] * modld;
    float4 dxh[MAXE];
    float m1 = 0.f, m2 = 0.f;
    const int64_t plane = (int64_t)n * H;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        dxh[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < H) {
            const float4 sh = *reinterpret_cast<const float4 *>(m + k), sc = *reinterpret_cast<const float4 *>(m + H + k);
            const float4 gt = *reinterpret_cast<const float4 *>(m + 2 * H + k);
            const float4 dd = *reinterpret_cast<const float4 *>(dh + (int64_t)v * H + k);
            const float xh[4] = {(t[e].x - mean) * rstd, (t[e].y - mean) * rstd, (t[e].z - mean) * rstd, (t[e].w - mean) * rstd};
            const float shv[4] = {sh.x, sh.y, sh.z, sh.w}, scv[4] = {sc.x, sc.y, sc.z, sc.w}, gtv[4] = {gt.x, gt.y, gt.z, gt.w};
            const float dv[4] = {dd.x, dd.y, dd.z, dd.w};
            float cs[4], csc[4], cg[4], dx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float y0 = xh[q] * (1.f + scv[q]) + shv[q];
                const float y = gelu ? gelu_erf(y0) : y0;
                cg[q] = dv[q] * y;
                const float dy0 = dv[q] * gtv[q] * (gelu ? gelu_grad(y0) : 1.f);
                cs[q] = dy0;
                csc[q] = dy0 * xh[q];
                dx[q] = dy0 * (1.f + scv[q]);
                m1 += dx[q];
                m2 += dx[q] * xh[q];
            }
            t[e] = make_float4(xh[0], xh[1], xh[2], xh[3]);
            dxh[e] = make_float4(dx[0], dx[1], dx[2], dx[3]);
            *reinterpret_cast<float4 *>(c3 + (int64_t)v * H + k) = make_float4(cs[0], cs[1], cs[2], cs[3]);
            *reinterpret_cast<float4 *>(c3 + plane + (int64_t)v * H + k) = make_float4(csc[0], csc[1], csc[2], csc[3]);
            *reinterpret_cast<float4 *>(c3 + 2 * plane + (int64_t)v * H + k) = make_float4(cg[0], cg[1], cg[2], cg[3]);
        }
    }
    m1 = wave_sum(m1) / (float)Ht;
    m2 = wave_sum(m2) / (float)Ht;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int k = (lane + e * 64) * 4;
        if (k < H)      // (padded columns: d h = 0 there, and no gradient leaves through them)
            gin_store4<T>(dz + (int64_t)v * H + k,
                          make_float4(k < Ht ? rstd * (dxh[e].x - m1 - t[e].x * m2) : 0.f, k + 1 < Ht ? rstd * (dxh[e].y - m1 - t[e].y * m2) : 0.f,
                                      k + 2 < Ht ? rstd * (dxh[e].z - m1 - t[e].z * m2) : 0.f, k + 3 < Ht ? rstd * (dxh[e].w - m1 - t[e].w * m2) : 0.f));
    }
}

// d h_in[v] += (1+eps) dz0[v] + sum over OUT-edges (v -> u, a): dz0[u] * GELU'(h_in[v] + bond[a]).  CSR by SOURCE node.
__global__ __launch_bounds__(64) void gin_aggregate_bwd_kernel(const float *__restrict__ h_in, const float *__restrict__ dz0,
                                                                const int *__restrict__ rowptr_s, const int *__restrict__ dst_s,
                                                                const int *__restrict__ attr_s, const float *__restrict__ bond,
                                                                const float *__restrict__ eps, float *__restrict__ dh, int n, int H) {
    const int v = blockIdx.x;
    if (v >= n) return;
    const int lane = threadIdx.x;
    const float e1 = 1.f + eps[0];
    const int e0 = rowptr_s[v], e_end = rowptr_s[v + 1];
    for (int k = lane * 4; k < H; k += 256) {
        const float4 hv = *reinterpret_cast<const float4 *>(h_in + (int64_t)v * H + k);
        const float4 dzv = *reinterpret_cast<const float4 *>(dz0 + (int64_t)v * H + k);
        float4 acc = *reinterpret_cast<const float4 *>(dh + (int64_t)v * H + k);
        acc.x += e1 * dzv.x; acc.y += e1 * dzv.y; acc.z += e1 * dzv.z; acc.w += e1 * dzv.w;
        for (int e = e0; e < e_end; ++e) {
            const float4 du = *reinterpret_cast<const float4 *>(dz0 + (int64_t)dst_s[e] * H + k);
            const float4 bn = *reinterpret_cast<const float4 *>(bond + (int64_t)attr_s[e] * H + k);
            acc.x += du.x * gelu_grad(hv.x + bn.x);
            acc.y += du.y * gelu_grad(hv.y + bn.y);
            acc.z += du.z * gelu_grad(hv.z + bn.z);
            acc.w += du.w * gelu_grad(hv.w + bn.w);
        }
        *reinterpret_cast<float4 *>(dh + (int64_t)v * H + k) = acc;
    }
}

// segment-max backward: the gradient of pool[g][k] is shared EVENLY by the nodes of graph g that hold the maximum of feature k -- the
// backward of torch's scatter_reduce(amax), which is what torch_geometric 2.6.1's global_max_pool runs without torch_scatter (the
// reference's requirements.txt has none).  Ties are common: atoms with identical k-hop neighbourhoods carry identical features at depth k.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ h_in, const int *__restrict__ gptr,
                                                           const float *__restrict__ dpool, float *__restrict__ dh, int H) {
    const int g = blockIdx.x;
    const int v0 = gptr[g], v1 = gptr[g + 1];
    for (int k = blockIdx.y * 256 + threadIdx.x; k < H; k += gridDim.y * 256) {
        float best = -INFINITY;
        int ties = 0;
        for (int v = v0; v < v1; ++v) {
            const float x = h_in[(int64_t)v * H + k];
            if (x > best) { best = x; ties = 1; }
            else if (x == best) ++ties;
        }
        if (ties == 0) continue;
        const float share = dpool[(int64_t)g * H + k] / (float)ties;
        for (int v = v0; v < v1; ++v)
            if (h_in[(int64_t)v * H + k] == best) dh[(int64_t)v * H + k] += share;
    }
}

__global__ void gather_rows_kernel(float *__restrict__ dst, const float *__restrict__ rows, const int *__restrict__ idx, int n, int H) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n * H; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = rows[(int64_t)idx[i / H] * H + i % H];
}
template <typename T> __global__ void cvt_rows_kernel(const float *__restrict__ src, int64_t lds, T *__restrict__ dst, int64_t ldd, int R, int C) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)R * C; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C, c = i - r * C;
        dst[r * ldd + c] = from_f32<T>(src[r * lds + c]);
    }
}
// out[K][ldo] (first N columns) = in[N][K]^T, operand dtype
template <typename T> __global__ void transpose_w_kernel(const float *__restrict__ in, T *__restrict__ out, int N, int K, int64_t ldo) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    for (int j = threadIdx.y; j < 32; j += 8)
        if (n0 + j < N && k0 + threadIdx.x < K) tile[j][threadIdx.x] = in[(int64_t)(n0 + j) * K + k0 + threadIdx.x];
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8)
        if (k0 + j < K && n0 + threadIdx.x < N) out[(int64_t)(k0 + j) * ldo + n0 + threadIdx.x] = from_f32<T>(tile[threadIdx.x][j]);
}
__global__ void sum_slabs_kernel(const float *__restrict__ slabs, int64_t stride, int splits, float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int z = 0; z < splits; ++z) a += slabs[z * stride + i];
        out[i] = a;
    }
}
// c, dc: the caller's dense [G][D] rows; g: rows of pitch ldg (the engine's padded text width)
__global__ void silu_bwd_kernel(const float *__restrict__ c, const float *__restrict__ g, float *__restrict__ dc, int G, int D, int ldg) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)G * D; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = c[i], sg = 1.f / (1.f + expf(-x));
        dc[i] = g[(i / D) * ldg + i % D] * sg * (1.f + x * (1.f - sg));
    }
}

constexpr int HEAD_SPLITS = 16;
static int64_t head_kp(int out_dim) { return (int64_t)round_up(out_dim, 64 * HEAD_SPLITS); }

// transposed operand-dtype copies of every weight the reverse sweep multiplies by (same slot offsets as the arena; the
// template head gets a K-padded row pitch; the L adapter matrices are concatenated along K for one GEMM)
template <typename T>
static int gin_build_transposes(GinEngine *e, hipStream_t st) {
    const LLGinConfig &cf = e->cfg;
    const int H = e->d.Hp, D = e->d.Dp, L = cf.num_layer;      // padded pitches: the transposes are taken of the internal arena
    int64_t total = 0;
    for (auto &p : e->layout) total = std::max<int64_t>(total, p.ioffset + (p.inumel + 63) / 64 * 64);
    const int64_t kp = head_kp(cf.out_dim);
    LL_TRY(e->wT.ensure((size_t)(total + (int64_t)4 * H * kp) * sizeof(T)));     // head copy appended after the arena image
    LL_TRY(e->adT.ensure((size_t)D * L * 3 * H * sizeof(T)));
    auto tr = [&](const std::string &name, int N, int K, T *dst, int64_t ldo) {
        dim3 grid(cdiv(K, 32), cdiv(N, 32)), blk(32, 8);
        hipLaunchKernelGGL((transpose_w_kernel<T>), grid, blk, 0, st, e->pf(name), dst, N, K, ldo);
    };
    auto slot = [&](const std::string &name) -> T * {
        for (auto &p : e->layout)
            if (p.name == name) return e->wT.as<T>() + p.ioffset;
        return nullptr;
    };
    for (int l = 0; l < L; ++l) {
        const std::string p = "convs." + std::to_string(l) + ".";
        tr(p + "mlp.0.weight", 4 * H, H, slot(p + "mlp.0.weight"), 4 * H);       // [H][4H]
        tr(p + "mlp.4.weight", H, 4 * H, slot(p + "mlp.4.weight"), H);           // [4H][H]
        tr("adapters." + std::to_string(l) + ".1.weight", 3 * H, D, e->adT.as<T>() + (int64_t)l * 3 * H, (int64_t)L * 3 * H);
        if (l < L - 1) {
            const std::string q = "mlp_virtualnode_list." + std::to_string(l) + ".";
            tr(q + "0.weight", 4 * H, H, slot(q + "0.weight"), 4 * H);
            tr(q + "4.weight", H, 4 * H, slot(q + "4.weight"), H);
        }
    }
    tr("decoder.0.weight", 4 * H, H, slot("decoder.0.weight"), 4 * H);
    LL_HIP(hipMemsetAsync(e->wT.as<T>() + total, 0, (size_t)4 * H * kp * sizeof(T), st));
    tr("decoder.4.weight", cf.out_dim, 4 * H, e->wT.as<T>() + total, kp);        // [4H][kp], zero-padded columns
    LL_LAUNCH_CHECK();
    return LL_OK;
}

template <typename T>
static int gin_backward_c_t(GinEngine *e, const int *rowptr_s, const int *dst_s, const int *attr_s, const int *batch,
                            const int *gptr, int n, int G, const float *c, const float *dlogits, float *dc, hipStream_t st) {
    const LLGinConfig &cf = e->cfg;
    const int H = e->d.Hp, Ht = e->d.Ht, D = e->d.Dp, L = cf.num_layer, dt = cf.dtype, es = sizeof(T);
    const int np = round_up(n, 128), Gp = round_up(G, 128);
    const int64_t kp = head_kp(cf.out_dim);
    if (!e->wT.p) LL_TRY(gin_build_transposes<T>(e, st));
    int64_t total = 0;
    for (auto &p : e->layout) total = std::max<int64_t>(total, p.ioffset + (p.inumel + 63) / 64 * 64);
    auto wt = [&](const std::string &name) -> const void * {
        for (auto &p : e->layout)
            if (p.name == name) return (const void *)(e->wT.as<T>() + p.ioffset);
        return nullptr;
    };
    LL_TRY(e->g_dh.ensure((size_t)np * H * 4));
    LL_TRY(e->g_dz.ensure((size_t)np * H * es));
    LL_TRY(e->g_dt1a.ensure((size_t)np * 4 * H * 4));
    LL_TRY(e->g_dt1.ensure((size_t)np * 4 * H * es));
    LL_TRY(e->g_dz0.ensure((size_t)np * H * 4));
    LL_TRY(e->g_c3.ensure((size_t)3 * np * H * 4));
    LL_TRY(e->g_dmod.ensure((size_t)Gp * L * 3 * H * 4));
    LL_TRY(e->g_dmoda.ensure((size_t)Gp * L * 3 * H * es));
    LL_TRY(e->g_dvn.ensure((size_t)Gp * H * 4));
    LL_TRY(e->g_dvna.ensure((size_t)Gp * H * es));
    LL_TRY(e->g_dvt1a.ensure((size_t)Gp * 4 * H * 4));
    LL_TRY(e->g_dvt1.ensure((size_t)Gp * 4 * H * es));
    LL_TRY(e->g_dpool.ensure((size_t)Gp * H * 4));
    LL_TRY(e->g_tmpG.ensure((size_t)Gp * H * 4));
    LL_TRY(e->g_dhead1a.ensure((size_t)Gp * 4 * H * 4));
    LL_TRY(e->g_dhead1.ensure((size_t)Gp * 4 * H * es));
    LL_TRY(e->g_dlog.ensure((size_t)Gp * kp * es));
    LL_TRY(e->g_slabs.ensure((size_t)HEAD_SPLITS * Gp * 4 * H * 4));
    LL_TRY(e->g_dcs.ensure((size_t)Gp * D * 4));
    const dim3 blk(256), w_n(n), w_g(G), wblk(64);
    auto ew = [](int64_t cnt) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((cnt + 255) / 256, 4096))); };

    // ---- readout: logits = GELU(LN(pool Wd0^T + b)) Wd4^T + b4;  pool = segment_sum(h_L)
    LL_HIP(hipMemsetAsync(e->g_dlog.p, 0, (size_t)Gp * kp * es, st));
    hipLaunchKernelGGL((cvt_rows_kernel<T>), ew((int64_t)G * cf.out_dim), blk, 0, st, dlogits, (int64_t)cf.out_dim, e->g_dlog.as<T>(), kp, G, cf.out_dim);
    LL_LAUNCH_CHECK();
    LL_TRY(linear_splitk_launch(dt, e->g_dlog.p, (int)kp, e->wT.as<T>() + total, (int)kp, e->g_slabs.as<float>(), 4 * H,
                                (int64_t)Gp * 4 * H, G, 4 * H, (int)kp, HEAD_SPLITS, st));
    hipLaunchKernelGGL(sum_slabs_kernel, ew((int64_t)G * 4 * H), blk, 0, st, e->g_slabs.as<float>(), (int64_t)Gp * 4 * H, HEAD_SPLITS,
                       e->g_dhead1a.as<float>(), (int64_t)G * 4 * H);
    hipLaunchKernelGGL((ln_gelu_bwd_kernel<T>), w_g, wblk, 0, st, e->sv_head1.as<float>(), e->pf("decoder.1.weight"), e->pf("decoder.1.bias"),
                       e->g_dhead1a.as<float>(), e->g_dhead1.as<T>(), G, 4 * H, 4 * Ht);
    LL_LAUNCH_CHECK();
    LL_TRY(linear_launch(dt, e->g_dhead1.p, 4 * H, wt("decoder.0.weight"), 4 * H, nullptr, e->g_dpool.p, H, G, H, 4 * H, 0, 1, st));
    hipLaunchKernelGGL(gather_rows_kernel, ew((int64_t)n * H), blk, 0, st, e->g_dh.as<float>(), e->g_dpool.as<float>(), batch, n, H);
    LL_HIP(hipMemsetAsync(e->g_dvn.p, 0, (size_t)Gp * H * 4, st));
    LL_LAUNCH_CHECK();

    for (int l = L - 1; l >= 0; --l) {
        const std::string p = "convs." + std::to_string(l) + ".";
        const bool last = (l == L - 1);
        const float *mod = e->mod.as<float>() + (size_t)l * 3 * H;      // rows of pitch L * 3H (one GEMM over all layers' adapters)
        // tail: h = gate * act(LN0(z)(1+scale)+shift) + h_in
        hipLaunchKernelGGL((gin_post_bwd_kernel<T>), w_n, wblk, 0, st, e->sv_z[l].as<float>(), mod, L * 3 * H, batch, e->g_dh.as<float>(), e->g_dz.as<T>(),
                           e->g_c3.as<float>(), n, H, last ? 0 : 1, Ht);
        LL_LAUNCH_CHECK();
        for (int q = 0; q < 3; ++q) {   // d(shift | scale | gate)[g] = sum over the graph's nodes
            hipLaunchKernelGGL((segment_pool_kernel<float, false>), dim3(G, cdiv(H, 1024)), blk, 0, st, e->g_c3.as<float>() + (size_t)q * n * H, gptr,
                               e->g_tmpG.as<float>(), (float *)nullptr, H);
            hipLaunchKernelGGL((cvt_rows_kernel<T>), ew((int64_t)G * H), blk, 0, st, e->g_tmpG.as<float>(), (int64_t)H,
                               e->g_dmoda.as<T>() + (size_t)l * 3 * H + (size_t)q * H, (int64_t)L * 3 * H, G, H);
        }
        LL_LAUNCH_CHECK();
        // MLP: z = GELU(LN(z0 W0^T + b0)) W4^T + b4
        LL_TRY(linear_launch(dt, e->g_dz.p, H, wt(p + "mlp.4.weight"), H, nullptr, e->g_dt1a.p, 4 * H, n, 4 * H, H, 0, 1, st));
        hipLaunchKernelGGL((ln_gelu_bwd_kernel<T>), w_n, wblk, 0, st, e->sv_t1[l].as<float>(), e->pf(p + "mlp.1.weight"), e->pf(p + "mlp.1.bias"),
                           e->g_dt1a.as<float>(), e->g_dt1.as<T>(), n, 4 * H, 4 * Ht);
        LL_LAUNCH_CHECK();
        LL_TRY(linear_launch(dt, e->g_dt1.p, 4 * H, wt(p + "mlp.0.weight"), 4 * H, nullptr, e->g_dz0.p, H, n, H, 4 * H, 0, 1, st));
        // aggregation: z0 = (1+eps) h_in + sum GELU(h_in[src] + bond)
        hipLaunchKernelGGL(gin_aggregate_bwd_kernel, w_n, wblk, 0, st, e->sv_hin[l].as<float>(), e->g_dz0.as<float>(), rowptr_s, dst_s, attr_s,
                           e->pf(p + "bond_encoder.weight"), e->pf(p + "eps"), e->g_dh.as<float>(), n, H);
        LL_LAUNCH_CHECK();
        if (!last) {   // vn_{l+1} = vn_l + MLP_vn(segment_max(h_in_l)): d vn_{l+1} is in g_dvn
            const std::string q = "mlp_virtualnode_list." + std::to_string(l) + ".";
            hipLaunchKernelGGL((cvt_rows_kernel<T>), ew((int64_t)G * H), blk, 0, st, e->g_dvn.as<float>(), (int64_t)H, e->g_dvna.as<T>(), (int64_t)H, G, H);
            LL_LAUNCH_CHECK();
            LL_TRY(linear_launch(dt, e->g_dvna.p, H, wt(q + "4.weight"), H, nullptr, e->g_dvt1a.p, 4 * H, G, 4 * H, H, 0, 1, st));
            hipLaunchKernelGGL((ln_gelu_bwd_kernel<T>), w_g, wblk, 0, st, e->sv_t1[l].as<float>() + (size_t)round_up(n, 64) * 4 * H, e->pf(q + "1.weight"), e->pf(q + "1.bias"),
                               e->g_dvt1a.as<float>(), e->g_dvt1.as<T>(), G, 4 * H, 4 * Ht);
            LL_LAUNCH_CHECK();
            LL_TRY(linear_launch(dt, e->g_dvt1.p, 4 * H, wt(q + "0.weight"), 4 * H, nullptr, e->g_dpool.p, H, G, H, 4 * H, 0, 1, st));
            hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(G, cdiv(H, 256)), blk, 0, st, e->sv_hin[l].as<float>(), gptr, e->g_dpool.as<float>(), e->g_dh.as<float>(), H);
            LL_LAUNCH_CHECK();
        }
        if (l > 0) {   // h_in_l = h_{l-1} + vn_l[batch]: d vn_l += segment_sum(d h_in_l); d h_{l-1} = d h_in_l (in place)
            hipLaunchKernelGGL((segment_pool_kernel<float, false>), dim3(G, cdiv(H, 1024)), blk, 0, st, e->g_dh.as<float>(), gptr, e->g_tmpG.as<float>(),
                               (float *)nullptr, H);
            hipLaunchKernelGGL(add_rows_kernel, dim3(cdiv(G * H, 256)), blk, 0, st, e->g_dvn.as<float>(), e->g_tmpG.as<float>(), (int64_t)G * H);
            LL_LAUNCH_CHECK();
        }
    }
    // ---- (shift, scale, gate)_l = Linear_l(SiLU(c)): one GEMM over the K-concatenated adapters, then SiLU'
    LL_TRY(linear_launch(dt, e->g_dmoda.p, L * 3 * H, e->adT.p, L * 3 * H, nullptr, e->g_dcs.p, D, G, D, L * 3 * H, 0, 1, st));
    hipLaunchKernelGGL(silu_bwd_kernel, ew((int64_t)G * cf.text_dim), blk, 0, st, c, e->g_dcs.as<float>(), dc, G, cf.text_dim, D);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

}  // namespace ll

namespace ll {
struct TopkWs { void *p = nullptr; size_t bytes = 0; };
static std::mutex g_topk_mu;
static std::map<hipStream_t, TopkWs> g_topk_ws;
static int g_topk_single = 0;      // ll_set_topk_single(1): one workgroup per row (round-1 form; A/B and tests)
}  // namespace ll

using namespace ll;

extern "C" {

int ll_gin_param_count(const LLGinConfig *cfg) {
    if (gin_check(cfg) != LL_OK) return LL_EINVAL;
    return (int)gin_layout(*cfg).size();
}
int ll_gin_param_info(const LLGinConfig *cfg, int idx, char *name, int name_cap, int64_t *numel, int64_t *offset) {
    LL_TRY(gin_check(cfg));
    auto v = gin_layout(*cfg);
    LL_CHECK(idx >= 0 && idx < (int)v.size(), "param index %d out of range", idx);
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", v[idx].name.c_str());
    if (numel) *numel = v[idx].numel;
    if (offset) *offset = v[idx].offset;
    return LL_OK;
}
int64_t ll_gin_arena_elems(const LLGinConfig *cfg) {
    if (gin_check(cfg) != LL_OK) return LL_EINVAL;
    auto v = gin_layout(*cfg);
    return v.back().offset + (v.back().numel + 63) / 64 * 64;
}
int ll_gin_create(const LLGinConfig *cfg, const float *d_weights_f32, void **handle) {
    LL_TRY(gin_check(cfg));
    LL_CHECK(d_weights_f32 && handle, "null argument");
    GinEngine *e = new GinEngine();
    e->cfg = *cfg;
    e->layout = gin_layout(*cfg);
    e->d = gin_dims(*cfg);
    e->w32 = d_weights_f32;
    e->sv_hin.resize(cfg->num_layer);
    e->sv_t1.resize(cfg->num_layer);
    e->sv_z.resize(cfg->num_layer);
    const int64_t elems = e->layout.back().ioffset + (e->layout.back().inumel + 63) / 64 * 64;
    if (e->d.padded) {      // the checkpoint's tensors into the zero-padded internal arena
        int rc = e->wpad.ensure((size_t)elems * 4);
        if (rc == LL_OK && hipMemset(e->wpad.p, 0, (size_t)elems * 4) != hipSuccess) { set_error("ll_gin_create: hipMemset failed"); rc = LL_EHIP; }
        for (size_t i = 0; i < e->layout.size() && rc == LL_OK; ++i) {
            const GinParam &pi = e->layout[i];
            const PadMap &m = pi.map;
            const int R = pi.rows, Cc = pi.cols;
            hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)std::min<int64_t>((pi.numel + 255) / 256, 4096)), dim3(256), 0, 0,
                               d_weights_f32 + pi.offset, e->wpad.as<float>() + pi.ioffset, R, Cc, pi.icols, m.rg2 ? m.rg2 : R, m.rgp2 ? m.rgp2 : R,
                               m.rg ? m.rg : R, m.rgp ? m.rgp : R, m.cg ? m.cg : Cc, m.cgp ? m.cgp : Cc);
        }
        if (rc == LL_OK && (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess)) { set_error("ll_gin_create: padding the weights failed"); rc = LL_EHIP; }
        if (rc != LL_OK) {
            ll_gin_destroy(e);
            return rc;
        }
        e->w32 = e->wpad.as<float>();
    }
    if (cfg->dtype == LL_BF16) {
        int rc = e->wop.ensure((size_t)elems * 2);
        if (rc == LL_OK) rc = convert_f32_to_bf16(e->w32, e->wop.as<bf16_t>(), elems, 0);
        if (rc != LL_OK) {
            ll_gin_destroy(e);
            return rc;
        }
    }
    e->cache_params();
    if (cfg->kind == 1) {      // the L adapter Linears as one [L * 3H, text_dim] weight (+ bias): their slots are not adjacent in the arena
        const int L = cfg->num_layer, es = cfg->dtype == LL_BF16 ? 2 : 4;
        const int Hp = e->d.Hp;
        const size_t per = (size_t)3 * Hp * e->d.Dp;
        int rc = e->adcat.ensure(per * L * es);
        if (rc == LL_OK) rc = e->adbcat.ensure((size_t)L * 3 * Hp * 4);
        for (int l = 0; l < L && rc == LL_OK; ++l) {
            const std::string p = "adapters." + std::to_string(l) + ".1.";
            if (hipMemcpy((char *)e->adcat.p + per * l * es, e->pw(p + "weight"), per * es, hipMemcpyDeviceToDevice) != hipSuccess ||
                hipMemcpy(e->adbcat.as<float>() + (size_t)l * 3 * Hp, e->pf(p + "bias"), (size_t)3 * Hp * 4,
                          hipMemcpyDeviceToDevice) != hipSuccess) {
                set_error("ll_gin_create: copying the adapter weights failed");
                rc = LL_EHIP;
            }
        }
        if (rc != LL_OK) {
            ll_gin_destroy(e);
            return rc;
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        set_error("hipDeviceSynchronize failed in ll_gin_create");
        ll_gin_destroy(e);
        return LL_EHIP;
    }
    *handle = e;
    return LL_OK;
}
int ll_gin_destroy(void *handle) {
    GinEngine *e = (GinEngine *)handle;
    if (!e) return LL_OK;
    (void)hipDeviceSynchronize();
    GBuf *bufs[] = {&e->ell, &e->wpad, &e->wop, &e->adcat, &e->adbcat, &e->h, &e->h_in, &e->a0, &e->t1, &e->t1a, &e->zs, &e->vn, &e->pool32, &e->poola,
                    &e->mod, &e->csilu, &e->head1, &e->head1a, &e->head2};
    for (GBuf *b : bufs) b->release();
    GBuf *tb[] = {&e->sv_head1, &e->wT, &e->adT, &e->g_dh, &e->g_dz, &e->g_dt1a, &e->g_dt1, &e->g_dz0, &e->g_c3, &e->g_dmod, &e->g_dmoda,
                  &e->g_dvn, &e->g_dvna, &e->g_dvt1a, &e->g_dvt1, &e->g_dpool, &e->g_tmpG, &e->g_dhead1a, &e->g_dhead1, &e->g_dlog,
                  &e->g_slabs, &e->g_dcs};
    for (GBuf *b : tb) b->release();
    for (auto *v : {&e->sv_hin, &e->sv_t1, &e->sv_z})
        for (GBuf &b : *v) b.release();
    delete e;
    return LL_OK;
}

int ll_graph_csr(const int64_t *x, const int64_t *edge_index, const int64_t *edge_attr, const int64_t *batch, int n_nodes, int n_edges,
                 int n_graphs, int32_t *x32, int32_t *rowptr, int32_t *src, int32_t *attr, int32_t *batch32, int32_t *gptr,
                 int32_t *scratch, int32_t *err, void *stream) {
    LL_CHECK(x && batch && x32 && rowptr && batch32 && gptr && scratch && err, "ll_graph_csr: null argument");
    LL_CHECK(n_nodes >= 1 && n_graphs >= 1 && n_edges >= 0, "ll_graph_csr: empty graph batch");
    LL_CHECK(n_edges == 0 || (edge_index && edge_attr && src && attr), "ll_graph_csr: edges given without edge_index / edge_attr / outputs");
    const size_t lds = (size_t)(2 * (size_t)n_nodes + n_graphs + n_edges + 2) * sizeof(int);
    if (lds <= 60 * 1024)
        hipLaunchKernelGGL(graph_csr_kernel<true>, dim3(1), dim3(1024), lds, (hipStream_t)stream, x, edge_index, edge_attr, batch, n_nodes, n_edges,
                           n_graphs, x32, rowptr, src, attr, batch32, gptr, scratch, err);
    else
        hipLaunchKernelGGL(graph_csr_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, edge_index, edge_attr, batch, n_nodes, n_edges,
                           n_graphs, x32, rowptr, src, attr, batch32, gptr, scratch, err);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_gin_forward(void *handle, const int32_t *x, const int32_t *rowptr, const int32_t *src, const int32_t *attr,
                   const int32_t *batch, const int32_t *gptr, int n_nodes, int n_edges, int n_graphs, const float *c,
                   float *out, float *pooled, void *stream) {
    GinEngine *e = (GinEngine *)handle;
    LL_CHECK(e && x && rowptr && batch && gptr && out, "null argument");
    LL_CHECK(n_nodes >= 1 && n_graphs >= 1 && n_edges >= 0, "empty graph batch");
    LL_CHECK(n_edges == 0 || (src && attr), "edges given without src/attr");
    if (e->cfg.dtype == LL_BF16)
        return gin_forward_t<bf16_t>(e, x, rowptr, src, attr, batch, gptr, n_nodes, n_edges, n_graphs, c, out, pooled, (hipStream_t)stream);
    return gin_forward_t<float>(e, x, rowptr, src, attr, batch, gptr, n_nodes, n_edges, n_graphs, c, out, pooled, (hipStream_t)stream);
}

int ll_gin_forward_train(void *handle, const int32_t *x, const int32_t *rowptr, const int32_t *src, const int32_t *attr,
                         const int32_t *batch, const int32_t *gptr, int n_nodes, int n_edges, int n_graphs, const float *c,
                         float *out, void *stream) {
    GinEngine *e = (GinEngine *)handle;
    LL_CHECK(e && e->cfg.kind == 1 && c, "ll_gin_forward_train: predictor engines with a text condition only");
    e->keep = true;
    const int rc = ll_gin_forward(handle, x, rowptr, src, attr, batch, gptr, n_nodes, n_edges, n_graphs, c, out, nullptr, stream);
    e->keep = false;
    if (rc != LL_OK) e->kept_n = e->kept_G = -1;
    return rc;
}

int ll_gin_backward_c(void *handle, const int32_t *rowptr_src, const int32_t *dst, const int32_t *attr, const int32_t *batch,
                      const int32_t *gptr, int n_nodes, int n_edges, int n_graphs, const float *c, const float *dlogits, float *dc,
                      void *stream) {
    GinEngine *e = (GinEngine *)handle;
    LL_CHECK(e && rowptr_src && batch && gptr && c && dlogits && dc, "null argument");
    LL_CHECK(e->cfg.kind == 1, "ll_gin_backward_c: predictor engines only");
    LL_CHECK(n_edges == 0 || (dst && attr), "edges given without dst/attr");
    if (e->kept_n != n_nodes || e->kept_G != n_graphs) {
        set_error("ll_gin_backward_c: no kept activations for %d nodes / %d graphs: call ll_gin_forward_train first", n_nodes, n_graphs);
        return LL_ESTATE;
    }
    if (e->cfg.dtype == LL_BF16)
        return gin_backward_c_t<bf16_t>(e, rowptr_src, dst, attr, batch, gptr, n_nodes, n_graphs, c, dlogits, dc, (hipStream_t)stream);
    return gin_backward_c_t<float>(e, rowptr_src, dst, attr, batch, gptr, n_nodes, n_graphs, c, dlogits, dc, (hipStream_t)stream);
}

int ll_softmax_topk(const float *logits, int rows, int out_dim, int k, float *probs, int32_t *idx, void *stream) {
    LL_CHECK(logits && probs && idx, "null argument");
    LL_CHECK(rows >= 1 && out_dim >= 1, "empty input");
    LL_CHECK(k >= 1 && k <= 64 && k <= out_dim, "k=%d must be in [1, min(64, out_dim)]", k);
    hipStream_t st = (hipStream_t)stream;
    const int chunks = cdiv(out_dim, TOPK_CHUNK);
    if (chunks == 1 || g_topk_single) {
        hipLaunchKernelGGL(softmax_topk_kernel<false>, dim3(rows), dim3(1024), 0, st, logits, out_dim, k, probs, idx, nullptr, nullptr, 1);
        LL_LAUNCH_CHECK();
        return LL_OK;
    }
    // two stages: (chunks x rows) workgroups reduce the row to chunks x k candidates, one workgroup per row merges them.
    // The candidate workspace belongs to the stream (calls on one stream are ordered; hipFree on growth synchronises).
    const bool vec = out_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
    const int chunk_len = round_up(cdiv(out_dim, chunks), 4);
    const size_t nc = (size_t)rows * chunks * k;
    float *wv = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_topk_mu);
        TopkWs &w = g_topk_ws[st];
        const size_t need = nc * 8 + (size_t)rows * chunks * 8;
        if (w.bytes < need) {
            if (w.p) (void)hipFree(w.p);
            w.p = nullptr;
            w.bytes = 0;
            LL_HIP(hipMalloc(&w.p, need));
            w.bytes = need;
        }
        wv = reinterpret_cast<float *>(w.p);
    }
    int *wi = reinterpret_cast<int *>(wv + nc);
    float *wstat = reinterpret_cast<float *>(wi + nc);
    if (vec) hipLaunchKernelGGL(topk_chunk_kernel<true>, dim3(chunks, rows), dim3(256), 0, st, logits, out_dim, k, chunk_len, wv, wi, wstat);
    else hipLaunchKernelGGL(topk_chunk_kernel<false>, dim3(chunks, rows), dim3(256), 0, st, logits, out_dim, k, chunk_len, wv, wi, wstat);
    if (chunks * k <= TOPK_CHUNK)
        hipLaunchKernelGGL(topk_merge_kernel, dim3(rows), dim3(256), 0, st, wv, wi, wstat, chunks, k, probs, idx);
    else      // more than 4096 candidates per row (> 262 144 templates at k = 64): the radix merge takes any count
        hipLaunchKernelGGL(softmax_topk_kernel<true>, dim3(rows), dim3(1024), 0, st, wv, chunks * k, k, probs, idx, wi, wstat, chunks);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

#if LL_TUNING
int ll_set_topk_single(int on) {
    const int old = g_topk_single;
    g_topk_single = on ? 1 : 0;
    return old;
}
#endif

int ll_cost_mlp(const float *weights, const float *fps, int n, float *out, void *stream) {
    LL_CHECK(weights && fps && out && n >= 1, "bad argument");
    hipLaunchKernelGGL(cost_mlp_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, weights, fps, out);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

}  // extern "C"
