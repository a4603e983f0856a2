// Shared body of the two decode-attention kernels (llm_ops.hip: ll_decode_attn_bf16; llm_layer.hip: the fused
// rope + append + attention).  One workgroup (256 threads) per (query head, sequence[, query position]).
//
// Latency structure: the first 256 keys -- the whole context of a short prompt -- are fetched BEFORE anything else is
// computed: thread j holds key row j (D/8 x 16 B) and wave w holds its slice of value rows 64w..64w+63 (one 2-/4-byte
// load per row per lane), all issued back to back, so K, V and the mask cost ONE memory round trip that overlaps the
// query preparation (rotary embedding, cache append).  Only contexts beyond 256 keys take further round trips (one per
// 256 keys for K, one per 32 keys per wave for V).  Scores, softmax statistics and P.V accumulate in f32; both kernels
// sum in the same order, so the fused layer stays bit-identical to the op-by-op path.
#pragma once
#include "common.h"

namespace ll {

typedef uint32_t au32x4 __attribute__((ext_vector_type(4)));

template <int D> struct AttnTile0 {
    au32x4 k[D / 8];      // key row `tid`
    uint32_t v[64];       // value rows 64*wave + i, this lane's D/64 elements
    unsigned char mk;     // mask byte of key `tid`
};

template <int D>
__device__ __forceinline__ void attn_prefetch(AttnTile0<D> &t, const bf16_t *__restrict__ Kb, const bf16_t *__restrict__ Vb,
                                              const unsigned char *__restrict__ mrow, int maxlen, int tid, int lane, int wave) {
    constexpr int EPL = D / 64;
    const bool in = tid < maxlen;
    t.mk = in ? mrow[tid] : (unsigned char)0;
#pragma unroll
    for (int c = 0; c < D / 8; ++c) t.k[c] = in ? *reinterpret_cast<const au32x4 *>(Kb + (int64_t)tid * D + c * 8) : (au32x4)(0);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int j = wave * 64 + i;
        uint32_t x = 0;
        if (j < maxlen) {
            if (EPL == 2) x = *reinterpret_cast<const uint32_t *>(Vb + (int64_t)j * D + lane * 2);
            else x = *reinterpret_cast<const unsigned short *>(Vb + (int64_t)j * D + lane);
        }
        t.v[i] = x;
    }
}

__device__ __forceinline__ float attn_dot8(const float *qs, au32x4 kv, float dsum) {
    const float4 q0 = *reinterpret_cast<const float4 *>(qs);
    const float4 q1 = *reinterpret_cast<const float4 *>(qs + 4);
    dsum = fmaf(q0.x, __uint_as_float(kv[0] << 16), dsum);
    dsum = fmaf(q0.y, __uint_as_float(kv[0] & 0xffff0000u), dsum);
    dsum = fmaf(q0.z, __uint_as_float(kv[1] << 16), dsum);
    dsum = fmaf(q0.w, __uint_as_float(kv[1] & 0xffff0000u), dsum);
    dsum = fmaf(q1.x, __uint_as_float(kv[2] << 16), dsum);
    dsum = fmaf(q1.y, __uint_as_float(kv[2] & 0xffff0000u), dsum);
    dsum = fmaf(q1.z, __uint_as_float(kv[3] << 16), dsum);
    dsum = fmaf(q1.w, __uint_as_float(kv[3] & 0xffff0000u), dsum);
    return dsum;
}

// qs [D] f32 query, part [4][D], sc [maxlen], red [8] in LDS; the caller has synchronised after writing qs (and kn / vn).
// NEWKV: key / value `p` are taken from kn / vn (LDS, bf16) instead of the cache.  Writes D outputs to outp.
template <int D, bool NEWKV>
__device__ __forceinline__ void attn_finish(AttnTile0<D> &t, const float *qs, float *part, float *sc, float *red,
                                            const bf16_t *__restrict__ Kb, const bf16_t *__restrict__ Vb,
                                            const unsigned char *__restrict__ mrow, int maxlen, long long p,
                                            const bf16_t *kn, const bf16_t *vn, float scale, bf16_t *__restrict__ outp,
                                            int tid, int lane, int wave) {
    constexpr int EPL = D / 64;
    // ---- scores: tile 0 from registers, further tiles one key per thread with the whole row in flight
    if (tid < maxlen) {
        const bool ok = t.mk != 0;
        float dsum = 0.f;
        if (ok) {
            if (NEWKV && tid == p) {
#pragma unroll
                for (int c = 0; c < D / 8; ++c) t.k[c] = *reinterpret_cast<const au32x4 *>(kn + c * 8);
            }
#pragma unroll
            for (int c = 0; c < D / 8; ++c) dsum = attn_dot8(qs + c * 8, t.k[c], dsum);
        }
        sc[tid] = ok ? dsum * scale : -INFINITY;
    }
    for (int j = tid + 256; j < maxlen; j += 256) {
        const bool ok = mrow[j] != 0;
        float dsum = 0.f;
        if (ok) {
            au32x4 kv[D / 8];
            if (NEWKV && j == p) {
#pragma unroll
                for (int c = 0; c < D / 8; ++c) kv[c] = *reinterpret_cast<const au32x4 *>(kn + c * 8);
            } else {
#pragma unroll
                for (int c = 0; c < D / 8; ++c) kv[c] = *reinterpret_cast<const au32x4 *>(Kb + (int64_t)j * D + c * 8);
            }
#pragma unroll
            for (int c = 0; c < D / 8; ++c) dsum = attn_dot8(qs + c * 8, kv[c], dsum);
        }
        sc[j] = ok ? dsum * scale : -INFINITY;
    }
    __syncthreads();
    // ---- softmax statistics
    float mx = -INFINITY;
    for (int j = tid; j < maxlen; j += 256) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < maxlen; j += 256) {
        const float e = (sc[j] == -INFINITY) ? 0.f : expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float den = red[4] + red[5] + red[6] + red[7];
    const float inv = den > 0.f ? 1.f / den : 0.f;      // a fully masked query row (left padding) yields zeros, not NaN
    // ---- out = P V: wave w takes keys 64w..64w+63 of tile 0 (registers), then keys 256 + w + 4u + 32 it
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int j = wave * 64 + i;
        const float pj = j < maxlen ? sc[j] : 0.f;
        uint32_t x = pj != 0.f ? t.v[i] : 0u;      // masked rows may hold anything (NaN from padded positions): never multiply them
        if (NEWKV && j == p) x = EPL == 2 ? *reinterpret_cast<const uint32_t *>(vn + lane * 2) : (uint32_t)vn[lane];
        acc[0] = fmaf(pj, __uint_as_float(x << 16), acc[0]);
        if (EPL == 2) acc[EPL - 1] = fmaf(pj, __uint_as_float(x & 0xffff0000u), acc[EPL - 1]);
    }
    for (int j0 = 256 + wave; j0 < maxlen; j0 += 32) {     // 8 keys (rows of 2*D bytes, coalesced) in flight per wave
        float pj[8];
        uint32_t vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + 4 * u;
            pj[u] = j < maxlen ? sc[j] : 0.f;
            vv[u] = 0;
            if (pj[u] != 0.f) {
                if (NEWKV && j == p) vv[u] = EPL == 2 ? *reinterpret_cast<const uint32_t *>(vn + lane * 2) : (uint32_t)vn[lane];
                else if (EPL == 2) vv[u] = *reinterpret_cast<const uint32_t *>(Vb + (int64_t)j * D + lane * 2);
                else vv[u] = *reinterpret_cast<const unsigned short *>(Vb + (int64_t)j * D + lane);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = fmaf(pj[u], __uint_as_float(vv[u] << 16), acc[0]);
            if (EPL == 2) acc[EPL - 1] = fmaf(pj[u], __uint_as_float(vv[u] & 0xffff0000u), acc[EPL - 1]);
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) part[wave * D + lane * EPL + e] = acc[e];
    __syncthreads();
    if (tid < D) {
        const float o = (part[tid] + part[D + tid] + part[2 * D + tid] + part[3 * D + tid]) * inv;
        outp[tid] = f32_to_bf16(o);
    }
}

}  // namespace ll
