// Fused elementwise kernels for the decode step of an untouched HuggingFace LLM.
//
// At batch-1 decode the HF forward issues ~1100 tiny ATen launches per token (RMSNorm = 7 launches, rotary embedding
// ~14 per layer, SiLU*mul = 2); each costs a launch boundary (~1.5-4 us) for a few KB of work.  These kernels do the
// same arithmetic -- including the intermediate bf16 roundings PyTorch's op-by-op evaluation implies, so results are
// bit-identical -- in one launch each:
//   ll_rmsnorm_bf16   Qwen2RMSNorm / LlamaRMSNorm forward  (transformers modeling_qwen2.py: x.float -> rsqrt(mean(x^2)+eps)
//                     -> to(bf16) -> weight * x)
//   ll_rope_bf16      apply_rotary_pos_emb: q*cos + rotate_half(q)*sin for q and k (strided [B,h,S,d] views)
//   ll_silu_mul_bf16  act_fn(gate) * up of the gated MLP
#include "common.h"
#include "attn_decode.h"

namespace ll {

__device__ __forceinline__ float bfr(float v) { return bf16_to_f32(f32_to_bf16(v)); }   // round through bf16

__global__ __launch_bounds__(256) void rmsnorm_bf16_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ w,
                                                            bf16_t *__restrict__ out, int H, float eps) {
    __shared__ float red[4];
    const bf16_t *xr = x + (int64_t)blockIdx.x * H;
    bf16_t *orow = out + (int64_t)blockIdx.x * H;
    constexpr int MAXC = 4;   // 16-byte chunks per thread: H <= 8192
    uint4 v[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = (threadIdx.x + c * 256) * 8;
        v[c] = k < H ? *reinterpret_cast<const uint4 *>(xr + k) : make_uint4(0, 0, 0, 0);
        const uint32_t u[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float a = __uint_as_float(u[t] << 16), b = __uint_as_float(u[t] & 0xffff0000u);
            ss = fmaf(a, a, ss);
            ss = fmaf(b, b, ss);
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)H;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = (threadIdx.x + c * 256) * 8;
        if (k < H) {
            const uint4 wv = *reinterpret_cast<const uint4 *>(w + k);
            const uint32_t u[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
            const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
            uint32_t o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float a = bfr(__uint_as_float(u[t] << 16) * rstd) * __uint_as_float(ww[t] << 16);
                const float b = bfr(__uint_as_float(u[t] & 0xffff0000u) * rstd) * __uint_as_float(ww[t] & 0xffff0000u);
                o[t] = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
            }
            *reinterpret_cast<uint4 *>(orow + k) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// one wave per (batch, head, position) row of q or k; lane d handles elements d and d + D/2 (D <= 128)
__global__ __launch_bounds__(64) void rope_bf16_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ k,
                                                        const bf16_t *__restrict__ cs, const bf16_t *__restrict__ sn,
                                                        bf16_t *__restrict__ qo, bf16_t *__restrict__ ko, int B, int nh,
                                                        int nkv, int S, int D, int64_t qs0, int64_t qs1, int64_t qs2,
                                                        int64_t ks0, int64_t ks1, int64_t ks2, int64_t cs0, int64_t cs1) {
    int r = blockIdx.x;
    const int rows_q = B * nh * S;
    const bool isq = r < rows_q;
    if (!isq) r -= rows_q;
    const int H = isq ? nh : nkv;
    const int s = r % S, h = (r / S) % H, b = r / (S * H);
    const bf16_t *src = isq ? q + b * qs0 + h * qs1 + s * qs2 : k + b * ks0 + h * ks1 + s * ks2;
    bf16_t *dst = (isq ? qo : ko) + (((int64_t)b * H + h) * S + s) * D;
    const bf16_t *c = cs + b * cs0 + s * cs1;
    const bf16_t *sv = sn + b * cs0 + s * cs1;
    const int half = D / 2;
    const int d = threadIdx.x;
    if (d < half) {
        const float x1 = bf16_to_f32(src[d]), x2 = bf16_to_f32(src[d + half]);
        const float c1 = bf16_to_f32(c[d]), c2 = bf16_to_f32(c[d + half]);
        const float s1 = bf16_to_f32(sv[d]), s2 = bf16_to_f32(sv[d + half]);
        // q*cos -> bf16;  rotate_half(q) = (-x2, x1);  rotate_half*sin -> bf16;  sum -> bf16
        dst[d] = f32_to_bf16(bfr(x1 * c1) + bfr(-x2 * s1));
        dst[d + half] = f32_to_bf16(bfr(x2 * c2) + bfr(x1 * s2));
    }
}

__global__ __launch_bounds__(256) void silu_mul_bf16_kernel(const bf16_t *__restrict__ g, const bf16_t *__restrict__ u,
                                                             bf16_t *__restrict__ out, int rows, int cols8, int64_t ld_in) {
    const int64_t n8 = (int64_t)rows * cols8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols8), c = (int)(i - (int64_t)r * cols8);
        const uint4 gv = *reinterpret_cast<const uint4 *>(g + r * ld_in + c * 8), uv = *reinterpret_cast<const uint4 *>(u + r * ld_in + c * 8);
        const uint32_t gg[4] = {gv.x, gv.y, gv.z, gv.w}, uu[4] = {uv.x, uv.y, uv.z, uv.w};
        uint32_t o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float a = bfr(silu(__uint_as_float(gg[t] << 16))) * __uint_as_float(uu[t] << 16);
            const float b = bfr(silu(__uint_as_float(gg[t] & 0xffff0000u))) * __uint_as_float(uu[t] & 0xffff0000u);
            o[t] = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
        }
        reinterpret_cast<uint4 *>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// KV-cache append for a StaticCache layer: keys/values [B,nkv,maxlen,D] (contiguous) get the S new rows of
// k_new/v_new ([B,nkv,S,D] views given by element strides) at positions *pos .. *pos+S-1.  Replaces the per-layer
// arange + add + add_ + 2 x index_copy_ launches of transformers' StaticLayer.update at decode.
__global__ __launch_bounds__(64) void kv_append_bf16_kernel(bf16_t *__restrict__ K, bf16_t *__restrict__ V,
                                                             const bf16_t *__restrict__ kn, const bf16_t *__restrict__ vn,
                                                             const long long *__restrict__ pos, int nkv, int S, int maxlen,
                                                             int D, int64_t ks0, int64_t ks1, int64_t ks2, int64_t vs0,
                                                             int64_t vs1, int64_t vs2) {
    const int r = blockIdx.x;            // (b*nkv + h)*S + s
    const int s = r % S, h = (r / S) % nkv, b = r / (S * nkv);
    const long long p = *pos + s;
    if (p < 0 || p >= maxlen) return;
    const bf16_t *ksrc = kn + b * ks0 + h * ks1 + s * ks2;
    const bf16_t *vsrc = vn + b * vs0 + h * vs1 + s * vs2;
    const int64_t dst = (((int64_t)b * nkv + h) * maxlen + p) * D;
    for (int d = threadIdx.x * 2; d < D; d += 128) {
        *reinterpret_cast<uint32_t *>(K + dst + d) = *reinterpret_cast<const uint32_t *>(ksrc + d);
        *reinterpret_cast<uint32_t *>(V + dst + d) = *reinterpret_cast<const uint32_t *>(vsrc + d);
    }
}

// Decode attention over a static KV cache (GQA): one workgroup per (query head, batch*query position).
//   scores_j = q . K[j] * scale for keys with mask[b,0,s,j] true; softmax in f32; out = sum_j p_j V[j].
// Replaces repeat_kv (two full-cache copies) + SDPA + mask fills of the HF sdpa path at decode.  D in {64, 128}.
// Body in attn_decode.h (shared with the fused rope + append + attention kernel of llm_layer.hip).
template <int D>
__global__ __launch_bounds__(ATTN_THREADS) void decode_attn_bf16_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ K,
                                                               const bf16_t *__restrict__ V, const unsigned char *__restrict__ mask,
                                                               bf16_t *__restrict__ out, int nh, int nkv, int S, int maxlen,
                                                               float scale, int64_t qs0, int64_t qs1, int64_t qs2,
                                                               int64_t ms0, int64_t ms2) {
    extern __shared__ __attribute__((aligned(16))) float sm_attn[];
    float *qs = sm_attn;                 // [D]     query in f32 (16-B aligned LDS broadcast reads)
    float *part = qs + D;                // [waves][rows per load][D] partial outputs = ATTN_PART_FLOATS for either D
    float *sc = part + ATTN_PART_FLOATS; // [maxlen] scores -> probabilities
    __shared__ float red[2 * ATTN_WAVES];
    const int h = blockIdx.x, bs = blockIdx.y;
    const int b = bs / S, s = bs - b * S;
    const int kvh = h / (nh / nkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t *Kb = K + ((int64_t)b * nkv + kvh) * maxlen * D;
    const bf16_t *Vb = V + ((int64_t)b * nkv + kvh) * maxlen * D;
    const unsigned char *mrow = mask + b * ms0 + s * ms2;
    AttnTile0<D> t0;
    attn_prefetch<D>(t0, Kb, Vb, mrow, maxlen, tid, lane, wave);
    if (tid < D) qs[tid] = bf16_to_f32(q[b * qs0 + h * qs1 + s * qs2 + tid]);
    __syncthreads();
    attn_finish<D, false>(t0, qs, part, sc, red, Kb, Vb, mrow, maxlen, -1, nullptr, nullptr, scale,
                          out + (((int64_t)b * S + s) * nh + h) * D, tid, lane, wave);
}

}  // namespace ll

using namespace ll;

extern "C" {

int ll_rmsnorm_bf16(const void *x, const void *w, void *out, int rows, int H, float eps, void *stream) {
    LL_CHECK(x && w && out && rows >= 1, "bad argument");
    LL_CHECK(H % 8 == 0 && H <= 8192, "ll_rmsnorm_bf16: H=%d must be a multiple of 8 and <= 8192", H);
    hipLaunchKernelGGL(rmsnorm_bf16_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, (const bf16_t *)w,
                       (bf16_t *)out, H, eps);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_rope_bf16(const void *q, const void *k, const void *cos, const void *sin, void *qo, void *ko, int B, int nh, int nkv,
                 int S, int D, const int64_t *qstr, const int64_t *kstr, const int64_t *cstr, void *stream) {
    LL_CHECK(q && k && cos && sin && qo && ko && qstr && kstr && cstr, "null argument");
    LL_CHECK(D % 2 == 0 && D <= 128, "ll_rope_bf16: head_dim %d must be even and <= 128", D);
    const int rows = B * (nh + nkv) * S;
    hipLaunchKernelGGL(rope_bf16_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, (const bf16_t *)q, (const bf16_t *)k,
                       (const bf16_t *)cos, (const bf16_t *)sin, (bf16_t *)qo, (bf16_t *)ko, B, nh, nkv, S, D, qstr[0], qstr[1],
                       qstr[2], kstr[0], kstr[1], kstr[2], cstr[0], cstr[1]);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_silu_mul_bf16(const void *gate, const void *up, void *out, int rows, int cols, int64_t ld_in, void *stream) {
    LL_CHECK(gate && up && out && rows >= 1 && cols >= 8 && cols % 8 == 0 && ld_in % 8 == 0,
             "ll_silu_mul_bf16: cols and ld_in must be positive multiples of 8");
    const int64_t n8 = (int64_t)rows * (cols / 8);
    int blocks = (int)((n8 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(silu_mul_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)gate,
                       (const bf16_t *)up, (bf16_t *)out, rows, cols / 8, ld_in);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_kv_append_bf16(void *K, void *V, const void *k_new, const void *v_new, const int64_t *pos, int B, int nkv, int S,
                      int maxlen, int D, const int64_t *kstr, const int64_t *vstr, void *stream) {
    LL_CHECK(K && V && k_new && v_new && pos && kstr && vstr, "null argument");
    LL_CHECK(D % 2 == 0 && B >= 1 && nkv >= 1 && S >= 1, "bad shape");
    hipLaunchKernelGGL(kv_append_bf16_kernel, dim3(B * nkv * S), dim3(64), 0, (hipStream_t)stream, (bf16_t *)K, (bf16_t *)V,
                       (const bf16_t *)k_new, (const bf16_t *)v_new, (const long long *)pos, nkv, S, maxlen, D, kstr[0], kstr[1],
                       kstr[2], vstr[0], vstr[1], vstr[2]);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_decode_attn_bf16(const void *q, const void *K, const void *V, const void *mask, void *out, int B, int nh, int nkv, int S,
                        int maxlen, int D, float scale, const int64_t *qstr, const int64_t *mstr, void *stream) {
    LL_CHECK(q && K && V && mask && out && qstr && mstr, "null argument");
    LL_CHECK((D == 64 || D == 128) && nh % nkv == 0 && maxlen >= 1 && maxlen <= 16384, "ll_decode_attn_bf16: unsupported shape");
    const size_t lds = ((size_t)maxlen + D + ATTN_PART_FLOATS) * 4;
    dim3 grid(nh, B * S);
    if (D == 128)
        hipLaunchKernelGGL((decode_attn_bf16_kernel<128>), grid, dim3(ATTN_THREADS), lds, (hipStream_t)stream, (const bf16_t *)q,
                           (const bf16_t *)K, (const bf16_t *)V, (const unsigned char *)mask, (bf16_t *)out, nh, nkv, S, maxlen,
                           scale, qstr[0], qstr[1], qstr[2], mstr[0], mstr[1]);
    else
        hipLaunchKernelGGL((decode_attn_bf16_kernel<64>), grid, dim3(ATTN_THREADS), lds, (hipStream_t)stream, (const bf16_t *)q,
                           (const bf16_t *)K, (const bf16_t *)V, (const unsigned char *)mask, (bf16_t *)out, nh, nkv, S, maxlen,
                           scale, qstr[0], qstr[1], qstr[2], mstr[0], mstr[1]);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

}  // extern "C"
