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
                                                             bf16_t *__restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const uint4 gv = reinterpret_cast<const uint4 *>(g)[i], uv = reinterpret_cast<const uint4 *>(u)[i];
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

int ll_silu_mul_bf16(const void *gate, const void *up, void *out, int64_t n, void *stream) {
    LL_CHECK(gate && up && out && n >= 8 && n % 8 == 0, "ll_silu_mul_bf16: n must be a positive multiple of 8");
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(silu_mul_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)gate,
                       (const bf16_t *)up, (bf16_t *)out, n / 8);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

}  // extern "C"
