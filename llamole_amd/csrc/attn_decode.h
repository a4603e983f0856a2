// Shared body of the two decode-attention kernels (llm_ops.hip: ll_decode_attn_bf16; llm_layer.hip: the fused
// rope + append + attention).  One workgroup (ATTN_WAVES waves) per (query head, sequence[, query position]).
//
// Access pattern: a wave-level load instruction covers RPI = 1024/(2 D) whole key (or value) rows -- LPR = D/8 lanes x 16 B
// per row, fully coalesced -- so a tile of 256 keys is 64/RPI load instructions per wave for K and as many for V.
//   scores: lane (r, c) holds q[8c..8c+7] in registers and dots it with its 16 B of key row 4i + r; a DPP reduction over
//           the LPR lanes of the row finishes the dot product (no LDS traffic, no per-thread row walks);
//   P.V   : the same lanes accumulate p[row] * v[row][8c..8c+7] over the rows they see; the RPI row groups are combined
//           through LDS at the end.
// The first tile -- the whole context of a short prompt -- is requested BEFORE anything else is computed (K, V and mask:
// one memory round trip that overlaps the query preparation: rotary embedding, cache append).  Softmax statistics in f32.
// Both kernels share this code, so the fused layer stays bit-identical to the op-by-op path.
#pragma once
#include "common.h"

namespace ll {

typedef uint32_t au32x4 __attribute__((ext_vector_type(4)));

// Waves per workgroup.  A tile is 256 keys; with 8 waves a lane has 2 x 8 K / V loads of the first tile in flight instead of
// 2 x 16 -- beyond ~16 outstanding loads per thread the ingest of a launch slows down (tools/phase_floor_probe.hip).
constexpr int ATTN_WAVES = 8;
constexpr int ATTN_THREADS = ATTN_WAVES * 64;
constexpr int ATTN_PART_FLOATS = ATTN_WAVES * 512;      // [waves][rows per load][D]: RPI * D = 512 for either D

template <int D> struct AttnGeom {
    static constexpr int LPR = D / 8;                      // lanes per row (16 B each)
    static constexpr int RPI = 64 / LPR;                   // rows per wave-level load instruction
    static constexpr int KPW = 256 / ATTN_WAVES;           // keys per wave of a 256-key tile
    static constexpr int NI = KPW / RPI;                   // instructions per wave for its keys of a tile
};

template <int D> struct AttnTile0 {
    au32x4 k[AttnGeom<D>::NI];
    au32x4 v[AttnGeom<D>::NI];
    unsigned char mk;     // mask byte of key `tid`
};

// key index handled by (wave, instruction i, lane) inside a tile starting at j0
template <int D> __device__ __forceinline__ int attn_key(int j0, int wave, int i, int lane) {
    return j0 + wave * AttnGeom<D>::KPW + i * AttnGeom<D>::RPI + lane / AttnGeom<D>::LPR;
}

template <int D>
__device__ __forceinline__ void attn_prefetch(AttnTile0<D> &t, const bf16_t *__restrict__ Kb, const bf16_t *__restrict__ Vb,
                                              const unsigned char *__restrict__ mrow, int maxlen, int tid, int lane, int wave) {
    constexpr int LPR = AttnGeom<D>::LPR, NI = AttnGeom<D>::NI;
    t.mk = tid < maxlen ? mrow[tid] : (unsigned char)0;
    const int c = lane % LPR;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = attn_key<D>(0, wave, i, lane);
        t.k[i] = j < maxlen ? *reinterpret_cast<const au32x4 *>(Kb + (int64_t)j * D + c * 8) : (au32x4)(0);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = attn_key<D>(0, wave, i, lane);
        t.v[i] = j < maxlen ? *reinterpret_cast<const au32x4 *>(Vb + (int64_t)j * D + c * 8) : (au32x4)(0);
    }
}

template <int LPR> __device__ __forceinline__ float attn_row_sum(float v) {
    if (LPR == 16) return row16_sum(v);
    return row8_sum(v);
}

__device__ __forceinline__ float attn_dot8(const float (&q)[8], au32x4 kv) {
    float d = 0.f;
    d = fmaf(q[0], __uint_as_float(kv[0] << 16), d);
    d = fmaf(q[1], __uint_as_float(kv[0] & 0xffff0000u), d);
    d = fmaf(q[2], __uint_as_float(kv[1] << 16), d);
    d = fmaf(q[3], __uint_as_float(kv[1] & 0xffff0000u), d);
    d = fmaf(q[4], __uint_as_float(kv[2] << 16), d);
    d = fmaf(q[5], __uint_as_float(kv[2] & 0xffff0000u), d);
    d = fmaf(q[6], __uint_as_float(kv[3] << 16), d);
    d = fmaf(q[7], __uint_as_float(kv[3] & 0xffff0000u), d);
    return d;
}
__device__ __forceinline__ void attn_axpy8(float (&acc)[8], float p, au32x4 vv) {
    acc[0] = fmaf(p, __uint_as_float(vv[0] << 16), acc[0]);
    acc[1] = fmaf(p, __uint_as_float(vv[0] & 0xffff0000u), acc[1]);
    acc[2] = fmaf(p, __uint_as_float(vv[1] << 16), acc[2]);
    acc[3] = fmaf(p, __uint_as_float(vv[1] & 0xffff0000u), acc[3]);
    acc[4] = fmaf(p, __uint_as_float(vv[2] << 16), acc[4]);
    acc[5] = fmaf(p, __uint_as_float(vv[2] & 0xffff0000u), acc[5]);
    acc[6] = fmaf(p, __uint_as_float(vv[3] << 16), acc[6]);
    acc[7] = fmaf(p, __uint_as_float(vv[3] & 0xffff0000u), acc[7]);
}

// qs [D] f32 query, part [ATTN_WAVES][RPI][D], sc [maxlen], red [2 * ATTN_WAVES] in LDS; the caller has synchronised after writing qs (and kn /
// vn).  NEWKV: key / value `p` are taken from kn / vn (LDS, bf16) instead of the cache.  Writes D outputs to outp.
template <int D, bool NEWKV>
__device__ __forceinline__ void attn_finish(AttnTile0<D> &t, const float *qs, float *part, float *sc, float *red,
                                            const bf16_t *__restrict__ Kb, const bf16_t *__restrict__ Vb,
                                            const unsigned char *__restrict__ mrow, int maxlen, long long p,
                                            const bf16_t *kn, const bf16_t *vn, float scale, bf16_t *__restrict__ outp,
                                            int tid, int lane, int wave) {
    constexpr int LPR = AttnGeom<D>::LPR, RPI = AttnGeom<D>::RPI, NI = AttnGeom<D>::NI;
    const int c = lane % LPR, r = lane / LPR;
    float q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = qs[c * 8 + e];
    // ---- scores (mask applied afterwards from the bytes fetched with the tile)
    for (int j0 = 0; j0 < maxlen; j0 += 256) {
        au32x4 kk[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = attn_key<D>(j0, wave, i, lane);
            if (j0 == 0) kk[i] = t.k[i];
            else kk[i] = (j < maxlen && mrow[j] != 0) ? *reinterpret_cast<const au32x4 *>(Kb + (int64_t)j * D + c * 8) : (au32x4)(0);
            if (NEWKV && j == p) kk[i] = *reinterpret_cast<const au32x4 *>(kn + c * 8);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = attn_key<D>(j0, wave, i, lane);
            const float d = attn_row_sum<LPR>(attn_dot8(q, kk[i]));
            if (c == 0 && j < maxlen) sc[j] = d * scale;
        }
    }
    __syncthreads();
    // ---- softmax statistics over the unmasked keys
    float mx = -INFINITY;
    for (int j = tid; j < maxlen; j += ATTN_THREADS) {
        const bool ok = (j == tid ? t.mk : mrow[j]) != 0;
        const float s = ok ? sc[j] : -INFINITY;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < ATTN_WAVES; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
    for (int j = tid; j < maxlen; j += ATTN_THREADS) {
        const float e = (sc[j] == -INFINITY) ? 0.f : expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[ATTN_WAVES + wave] = sum;
    __syncthreads();
    float den = red[ATTN_WAVES];
#pragma unroll
    for (int w = 1; w < ATTN_WAVES; ++w) den += red[ATTN_WAVES + w];
    const float inv = den > 0.f ? 1.f / den : 0.f;      // a fully masked query row (left padding) yields zeros, not NaN
    // ---- out = P V: lane (r, c) accumulates columns 8c..8c+7 over the rows it sees
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 < maxlen; j0 += 256) {
        au32x4 vv[NI];
        float pj[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = attn_key<D>(j0, wave, i, lane);
            pj[i] = j < maxlen ? sc[j] : 0.f;
            if (j0 == 0) vv[i] = t.v[i];
            else vv[i] = pj[i] != 0.f ? *reinterpret_cast<const au32x4 *>(Vb + (int64_t)j * D + c * 8) : (au32x4)(0);
            if (NEWKV && j == p) vv[i] = *reinterpret_cast<const au32x4 *>(vn + c * 8);
            if (pj[i] == 0.f) vv[i] = (au32x4)(0);      // masked rows may hold anything (NaN from padded positions)
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) attn_axpy8(acc, pj[i], vv[i]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[(wave * RPI + r) * D + c * 8 + e] = acc[e];
    __syncthreads();
    if (tid < D) {
        float o = 0.f;
#pragma unroll
        for (int g = 0; g < ATTN_WAVES * RPI; ++g) o += part[g * D + tid];
        outp[tid] = f32_to_bf16(o * inv);
    }
}

}  // namespace ll
