// Weight-streaming Linear for 17..64 token rows (batched decode of the HF LLM with up to 64 sequences per GPU: BASELINE configs[3]; the
// reference decodes whatever per_device_eval_batch_size hands it in one language_model.generate, eval/workflow.py:89-91,110-124,
// modeling_llamole.py:599-603): out[M,N] = epilogue(x[M,K] . W[N,K]^T + bias), bf16 operands, f32 accumulation on
// v_mfma_f32_16x16x32_bf16.
//
// What shapes the kernel (measured, profiles/r6_rows64_sweep.txt): at 64 rows the rows16 data path -- weight rows through registers into a
// wave-private LDS image -- is bound by the bytes a CU can keep in flight (LDS holds weights AND x: 64 KB of weight loads per CU,
// 3.1 TB/s on gate|up).  So the weights do not touch LDS at all here:
//   * W is read from a copy in MFMA A-operand order (ll_rows64_pack_bf16: the 16 rows x 32 k block one fragment instruction consumes
//     is 1 KB contiguous, row-tile major, as pack_mfma16 of the GraphDiT engine): one fully coalesced 16-byte load per lane per
//     fragment straight into the operand registers, two x-stages (16 loads, 16 KB per wave, 128 KB per CU) ahead of their use;
//   * x goes through a two-slot LDS ring of [64 token rows x 512 k] stages (1040-byte row pitch: conflict-free ds_read_b128 fragments),
//     loaded by the whole workgroup one stage ahead; ONE barrier per stage;
//   * a workgroup of 8 waves = 2 tile pairs x 4 k-parts owns 64 weight rows (SILU_MUL: 32 gate + the matching 32 up rows) x all 64 token
//     rows over its K range: x is read once per workgroup, i.e. L2 -> CU traffic for x equals the HBM traffic for W; every weight
//     fragment feeds four MFMA column blocks, every x fragment two weight tiles;
//   * the loads are inline asm with hand-counted s_waitcnt vmcnt(n) (hipcc drains vmcnt at a loop header: gemm.hip gemm_rs_kernel);
//   * the four k-part partial tiles are summed through LDS in k-part order (deterministic), each k-part wave finishing one column block.
// Matrices with few row groups (o_proj, down_proj: N / 64 = 64 groups on 256 CUs) split K over blockIdx.y into f32 slabs, and
// rows64_reduce_kernel (one workgroup per token row) sums the slabs in slice order, applies bias / residual and -- because it owns whole
// rows -- can also emit the RMSNorm of its output for the next Linear (Qwen2RMSNorm / LlamaRMSNorm as HF evaluates it), so that no
// normalisation arithmetic sits in a GEMM's main loop.
#include "common.h"

namespace ll {

typedef uint32_t r64_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 r64_bf16x8;
typedef __attribute__((ext_vector_type(4))) float r64_f32x4;

enum { R64_PLAIN = 0, R64_RESIDUAL = 1, R64_SILU_MUL = 2 };

__device__ __forceinline__ float r64_bfr(float v) { return bf16_to_f32(f32_to_bf16(v)); }

constexpr int R64_WAVES = 8, R64_KP = 4;        // waves per workgroup; k-parts
// KS = k per x stage (512: 16 MFMA k-steps; LDS row pitch KS * 2 + 16 bytes = 260 dwords = 4 mod 64).  KS = 256 halves the ring (67 KB) and
// the registers (<= 128: two workgroups per CU, or one next to another stream's): measured level with 512 alone (47.3 vs 47.9 us on gate|up
// at 64 rows) AND under the overlapped GraphDiT trajectory (66.2 vs 66.3 molecules/s, profiles/r6_ab_rows64_stage*.json): only 512 is
// instantiated

// [N, K] row-major (row pitch ldw) -> MFMA A-operand order, rows padded to a multiple of 16 with zeros:
// out[((tile * K/32 + kstep) * 64 + lane) * 8 + e] = W[tile * 16 + (lane & 15)][kstep * 32 + (lane >> 4) * 8 + e]
__global__ __launch_bounds__(256) void rows64_pack_kernel(const bf16_t *__restrict__ W, int ldw, bf16_t *__restrict__ out, int N, int K) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;        // one 16-byte piece per thread
    const int kts = K / 32;
    const int64_t total = (int64_t)((N + 15) / 16) * kts * 64;
    if (g >= total) return;
    const int l = (int)(g & 63);
    const int64_t blk = g >> 6;
    const int kt = (int)(blk % kts);
    const int64_t row = (blk / kts) * 16 + (l & 15);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N) v = *reinterpret_cast<const uint4 *>(W + row * (int64_t)ldw + kt * 32 + (l >> 4) * 8);
    *reinterpret_cast<uint4 *>(out + g * 8) = v;
}

#define R64_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// Persistent: grid = min(units, CUs) workgroups of 512 threads; a unit = (row group, K slice), unit u = group * ksg + slice, and
// workgroup w walks units w, w + gridDim.x, ... as ONE sequence of x stages ("positions"): the loads of the next unit's first stages are
// in flight while the current unit is finished, so a CU's weight stream never stops between units (with one unit per launch slot the
// prologue + drain of every workgroup cost 11 of 49 us on gate|up at 64 rows).
// slab != nullptr: raw f32 partial sums [ksg][M][N] (no bias / epilogue).  tiles_n = padded weight tiles per matrix half (SILU_MUL: the up
// tiles follow the gate tiles at +tiles_n).
template <int EPI, int MB, int KS>
__global__ __launch_bounds__(512, KS == 256 ? 4 : 2) void rows64_kernel(const bf16_t *__restrict__ X, int ldx, const bf16_t *__restrict__ Wp,
                                                     const float *__restrict__ bias, const bf16_t *__restrict__ res, int ldr,
                                                     bf16_t *__restrict__ C, int ldc, float *__restrict__ slab,
                                                     const float *__restrict__ row_ssq, int ssq_chunks, float eps, int M, int N, int K,
                                                     int tiles_n, int units, int ksg) {
    constexpr int R64_KS = KS, R64_XPITCH = KS * 2 + 16;
    constexpr int R64_KPW = KS / 32 / R64_KP;                   // k-steps per wave per stage (4 | 2)
    constexpr int R64_NW = R64_KPW * 2;                         // weight loads per wave per stage (two tiles)
    constexpr int NX = MB * 16 * (R64_KS / 8) / 512;            // 16-byte x loads per thread per stage
    constexpr int XLPR = KS / 8;                                // lanes per x row of a stage (64 | 32)
    constexpr int SLOT = MB * 16 * R64_XPITCH;
    static_assert(R64_WAVES * MB * 1024 <= SLOT, "the partial-tile exchange of a unit (one tile at a time) uses the ring slot its last stage was read from");
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_r64[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tp = wid & 1, kp = wid >> 1;
    const int kts = K / 32;
    const int nst = (K + KS - 1) / KS;                  // x stages of K
    const int sper = (nst + ksg - 1) / ksg;             // stages per K slice (a trailing slice may run past nst: empty stages)
    const int nmine = (int)blockIdx.x < units ? (units - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int P = nmine * sper;                         // positions of this workgroup
    // position p -> unit, x stage, this wave's two weight tiles
    auto unit_of = [&](int p) { return (int)blockIdx.x + (p / sper) * (int)gridDim.x; };
    // (sweeping the stages of a unit from a start that depends on its row group -- the packed tiles of all row groups are a power of
    // two apart -- changed nothing: 48.6..48.8 us on gate|up at 64 rows for rotations 0, 1, 3, 5; not kept)
    auto stage_of = [&](int p) { return (unit_of(p) % ksg) * sper + p % sper; };
    auto otile_of = [&](int u) { return EPI == R64_SILU_MUL ? (u / ksg) * 2 + tp : (u / ksg) * 4 + tp * 2; };
    r64_f32x4 acc[2][MB];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[t][b] = (r64_f32x4)(0.f);
    r64_u32x4 wr[2][R64_NW], xr[NX];
    // x chunk c = tid + 512 i of a stage: token row c / XLPR, 16-byte piece c % XLPR (one wave instruction = one or two whole rows of the stage).
    // Loads are `global_load_dwordx4 vdst, voffset, sbase`: the per-lane part of an address is a loop-invariant 32-bit offset, everything
    // that changes per stage is wave-uniform and lives in the scalar base -- no vector address arithmetic in the loop (the first version,
    // with 64-bit per-lane pointers and zero-selects, spent 188 VALU instructions per stage next to 32 MFMAs).  Token rows >= M re-read row
    // M - 1 (their results are never stored); k-steps past K are never multiplied.
    const int xrow0 = tid / XLPR, xpiece = tid % XLPR;
    constexpr int XRS = 512 / XLPR;                             // token rows per pass of the workgroup (8 | 16)
    uint32_t xoff[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xoff[i] = (uint32_t)min(xrow0 + i * XRS, M - 1) * (uint32_t)ldx * 2u + (uint32_t)xpiece * 16u;
    const uint32_t woff = (uint32_t)lane * 16u;
    auto xload = [&](int p) {
        const int st = stage_of(p);
        if ((st + 1) * KS <= K) {
            const unsigned char *base = reinterpret_cast<const unsigned char *>(X) + (int64_t)st * (KS * 2);
#pragma unroll
            for (int i = 0; i < NX; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xr[i]) : "v"(xoff[i]), "s"(base) : "memory");
        } else {            // the stage that straddles K (K % KS != 0) or lies past it: columns clamped per lane
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int k = min(st * KS + xpiece * 8, K - 8);
                const bf16_t *q = X + (int64_t)min(xrow0 + i * XRS, M - 1) * ldx + k;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xr[i]) : "v"(q) : "memory");
            }
        }
    };
    auto xwrite = [&](int p, int slot) {
        unsigned char *b = sm_r64 + slot * SLOT + xrow0 * R64_XPITCH + xpiece * 16;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            asm volatile("" : "+v"(xr[i]));
            *reinterpret_cast<r64_u32x4 *>(b + i * XRS * R64_XPITCH) = xr[i];
        }
    };
    auto wload = [&](int set, int p) {          // k-steps kp*KPW .. of position p, both tiles; k-steps past K re-read the last one (unused)
        const int st = stage_of(p), ot = otile_of(unit_of(p));
        const int t0 = min(ot, tiles_n - 1);
        const int t1 = EPI == R64_SILU_MUL ? t0 + tiles_n : min(ot + 1, tiles_n - 1);
        const unsigned char *w0 = reinterpret_cast<const unsigned char *>(Wp) + (int64_t)t0 * kts * 1024;
        const unsigned char *w1 = reinterpret_cast<const unsigned char *>(Wp) + (int64_t)t1 * kts * 1024;
#pragma unroll
        for (int i = 0; i < R64_KPW; ++i) {
            const int ks = min(st * (KS / 32) + kp * R64_KPW + i, kts - 1);
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(wr[set][2 * i]) : "v"(woff), "s"(w0 + (int64_t)ks * 1024) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(wr[set][2 * i + 1]) : "v"(woff), "s"(w1 + (int64_t)ks * 1024) : "memory");
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int set, int p, int slot) {
        const int st = stage_of(p);
        const unsigned char *b = sm_r64 + slot * SLOT;
#pragma unroll
        for (int i = 0; i < R64_KPW; ++i) {
            const int ksl = kp * R64_KPW + i;           // k-step within the stage
            if (st * (KS / 32) + ksl >= kts) break;     // wave-uniform: past K
            asm volatile("" : "+v"(wr[set][2 * i]), "+v"(wr[set][2 * i + 1]));
            r64_bf16x8 bf[MB];
#pragma unroll
            for (int c = 0; c < MB; ++c) bf[c] = *reinterpret_cast<const r64_bf16x8 *>(b + (c * 16 + fr) * R64_XPITCH + (ksl * 4 + fq) * 16);
            const r64_bf16x8 a0 = __builtin_bit_cast(r64_bf16x8, wr[set][2 * i]), a1 = __builtin_bit_cast(r64_bf16x8, wr[set][2 * i + 1]);
#pragma unroll
            for (int c = 0; c < MB; ++c) {
                acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf[c], acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf[c], acc[1][c], 0, 0, 0);
            }
        }
    };
    // end of a unit (the barrier behind its last stage has passed: ring slot `slot` is free until the next stage's x write): the partial
    // tiles of the four k-parts -> LDS; k-part wave kp finishes column block kp of its tile pair (summed in k-part order); accumulators reset
    auto finish = [&](int u, int slot) {
        // [tile][waves][MB][64 lanes][4]: both tiles of the pair at once when the slot holds them (512-k stages), else one tile at a time
        float *part = reinterpret_cast<float *>(sm_r64 + slot * SLOT);
        constexpr bool BOTH = 2 * R64_WAVES * MB * 1024 <= SLOT;
        constexpr int TSTRIDE = BOTH ? R64_WAVES * MB * 256 : 0;
        const int cb = kp, m = cb * 16 + (lane & 15);
        r64_f32x4 s0 = (r64_f32x4)(0.f), s1 = (r64_f32x4)(0.f);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int c = 0; c < MB; ++c) {
                *reinterpret_cast<r64_f32x4 *>(part + t * TSTRIDE + ((wid * MB + c) * 64 + lane) * 4) = acc[t][c];
                acc[t][c] = (r64_f32x4)(0.f);
            }
            if (BOTH && t == 0) continue;
            __syncthreads();
            if (kp < MB) {
#pragma unroll
                for (int tt = (BOTH ? 0 : t); tt <= t; ++tt) {
                    r64_f32x4 sum = (r64_f32x4)(0.f);
#pragma unroll
                    for (int q = 0; q < R64_KP; ++q) sum += *reinterpret_cast<const r64_f32x4 *>(part + tt * TSTRIDE + (((q * 2 + tp) * MB + cb) * 64 + lane) * 4);
                    if (tt == 0) s0 = sum; else s1 = sum;
                }
            }
            __syncthreads();        // the exchange buffer is free again (the second tile; then a ring slot)
        }
        if (kp < MB && m < M) {
            // s0[j] / s1[j] = C[weight row tile * 16 + (lane>>4)*4 + j][token cb*16 + (lane & 15)] of the wave's first / second tile
            if (row_ssq) {          // x was bf16(h * w_norm): rsqrt(mean(h^2) + eps) of the token row scales the accumulator (rows16 form)
                float t = 0.f;
                for (int c = 0; c < ssq_chunks; ++c) t += row_ssq[m * ssq_chunks + c];
                const float rstd = rsqrtf(t / (float)K + eps);
                s0 *= rstd;
                s1 *= rstd;
            }
            const int otile = otile_of(u), slice = u % ksg;
            constexpr int NOUT = EPI == R64_SILU_MUL ? 1 : 2;
#pragma unroll
            for (int t = 0; t < NOUT; ++t) {
                const r64_f32x4 sv = t == 0 ? s0 : s1;
                const int nb = (otile + t) * 16 + (lane >> 4) * 4;
                if (nb >= N) continue;
                if (slab) {
                    float *dst = slab + ((int64_t)slice * M + m) * N + nb;
                    if (nb + 3 < N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                        *reinterpret_cast<float4 *>(dst) = make_float4(sv[0], sv[1], sv[2], sv[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (nb + j < N) dst[j] = sv[j];
                    }
                    continue;
                }
                uint16_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = nb + j;
                    const bool ok = n < N;
                    if (EPI == R64_SILU_MUL) {
                        const float g = r64_bfr(s0[j] + ((bias && ok) ? bias[n] : 0.f));
                        const float up = r64_bfr(s1[j] + ((bias && ok) ? bias[n + N] : 0.f));
                        o[j] = f32_to_bf16(r64_bfr(silu(g)) * up);
                    } else {
                        float v = sv[j] + ((bias && ok) ? bias[n] : 0.f);
                        if (EPI == R64_RESIDUAL) v = (ok ? bf16_to_f32(res[(int64_t)m * ldr + n]) : 0.f) + r64_bfr(v);
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
        }
    };
    if (P == 0) return;
    // VMEM issue order: X(0) W(0) X(1) W(1), then per position p: X(p+2) at its top, W(p+2) behind its MFMAs.  At the top of position p
    // the loads in flight are W(p) < X(p+1) < W(p+1): one wait leaves W(p+1) flying.  (The stores and bias / residual loads of `finish`
    // are younger than all of them: they only make that wait conservative.)  A third weight register set -- W(p+2) requested at the top
    // of position p, 24 + 8 loads per lane in flight -- measured SLOWER (gate|up at 64 rows 50.1 vs 46.5 us): not kept.
    xload(0);
    wload(0, 0);
    R64_WAIT(R64_NW);
    xwrite(0, 0);
    if (1 < P) {
        xload(1);
        wload(1, 1);
    }
    __syncthreads();
    for (int p0 = 0; p0 < P; p0 += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {           // position p: weights in wr[h], x in slot h
            const int p = p0 + h;
            if (p >= P) break;
            if (p + 1 < P) {
                R64_WAIT(R64_NW);               // W(p), X(p+1) landed
                xwrite(p + 1, h ^ 1);           // slot h^1 was last read at position p-1 (barrier passed)
                if (p + 2 < P) xload(p + 2);
            } else {
                R64_WAIT(0);
            }
            compute(h, p, h);
            if (p + 2 < P) wload(h, p + 2);
            __syncthreads();
            if (p % sper == sper - 1) finish(unit_of(p), h);
        }
    }
}

// grid (token rows, 1024-column chunks), 256 threads x 4 columns: out[m][n] = epilogue(sum over slices (in slice order) of slab[z][m][n]
// x the optional input row scale + bias[n]).  normw != nullptr: also XS[m][n] = bf16(out[m][n] * normw[n]) and ssq[m][chunk] = sum over the chunk of out[m][n]^2 (of the
// ROUNDED output) -- the two halves of the next Linear's RMSNorm: the consumer multiplies its accumulator by rsqrt(sum(ssq) / N + eps).
// splits == 0, C == nullptr: the pre-norm of the residual rows alone (the embedding rows ahead of the first layer).
template <int EPI>
__global__ __launch_bounds__(256) void rows64_reduce_kernel(const float *__restrict__ slab, int splits, const float *__restrict__ bias,
                                                            const bf16_t *__restrict__ res, int ldr, bf16_t *__restrict__ C, int ldc,
                                                            const bf16_t *__restrict__ normw, bf16_t *__restrict__ XS, int ldn,
                                                            float *__restrict__ ssq, const float *__restrict__ row_ssq, int ssq_chunks,
                                                            float eps, int K, int M, int N) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n = (blockIdx.y * 256 + tid) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    if (n < N) {
        const bool full = n + 3 < N && (N & 3) == 0;
        if (full) {
            float4 t[8];
#pragma unroll
            for (int z = 0; z < 8; ++z)
                if (z < splits) t[z] = *reinterpret_cast<const float4 *>(slab + ((int64_t)z * M + m) * N + n);
#pragma unroll
            for (int z = 0; z < 8; ++z)
                if (z < splits) v[0] += t[z].x, v[1] += t[z].y, v[2] += t[z].z, v[3] += t[z].w;
        } else {
            for (int z = 0; z < splits; ++z)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < N) v[j] += slab[((int64_t)z * M + m) * N + n + j];
        }
        if (row_ssq) {          // the input row scale of a split RMSNorm (see rows64_kernel), applied to the complete sum
            float t = 0.f;
            for (int c = 0; c < ssq_chunks; ++c) t += row_ssq[m * ssq_chunks + c];
            const float rstd = rsqrtf(t / (float)K + eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= rstd;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j >= N) break;
            float r = v[j] + (bias ? bias[n + j] : 0.f);
            if (EPI == R64_RESIDUAL) r = bf16_to_f32(res[(int64_t)m * ldr + n + j]) + r64_bfr(r);
            const bf16_t ob = f32_to_bf16(r);
            if (C) C[(int64_t)m * ldc + n + j] = ob;
            r = bf16_to_f32(ob);
            ss = fmaf(r, r, ss);
            if (normw) XS[(int64_t)m * ldn + n + j] = f32_to_bf16(r * bf16_to_f32(normw[n + j]));
        }
    }
    if (!normw) return;
    ss = wave_sum(ss);
    if (lane == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) ssq[m * gridDim.y + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
}

static int g_rows64_ksg = 0;      // 0: by shape; else K slices over workgroups (tuning)

static int r64_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int EPI, int MB, int KS>
static int launch_rows64_ks(int ksg, hipStream_t s, const bf16_t *X, int ldx, const bf16_t *Wp, const float *bias, const bf16_t *res, int ldr,
                            bf16_t *C, int ldc, float *slab, const float *row_ssq, int ssq_chunks, float eps, int M, int N, int K) {
    const size_t lds = (size_t)2 * MB * 16 * (KS * 2 + 16);
    static bool attr_set = false;
    if (!attr_set) {
        LL_HIP(hipFuncSetAttribute((const void *)rows64_kernel<EPI, MB, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int tiles_n = (N + 15) / 16;
    const int groups = EPI == R64_SILU_MUL ? cdiv(tiles_n, 2) : cdiv(tiles_n, 4);
    const int units = groups * ksg;
    // workgroups resident at once: one per CU with 512-k stages (133 KB of LDS at 64 rows), two with 256-k stages (<= 128 VGPRs)
    const int slots = r64_cus() * (KS == 256 ? 2 : 1);
    const dim3 grid(units < slots ? units : slots);
    hipLaunchKernelGGL((rows64_kernel<EPI, MB, KS>), grid, dim3(512), lds, s, X, ldx, Wp, bias, res, ldr, C, ldc, slab, row_ssq, ssq_chunks, eps, M, N,
                       K, tiles_n, units, ksg);
    return LL_OK;
}

template <int EPI, int MB>
static int launch_rows64_mb(int ksg, hipStream_t s, const bf16_t *X, int ldx, const bf16_t *Wp, const float *bias, const bf16_t *res, int ldr,
                            bf16_t *C, int ldc, float *slab, const float *row_ssq, int ssq_chunks, float eps, int M, int N, int K) {
    return launch_rows64_ks<EPI, MB, 512>(ksg, s, X, ldx, Wp, bias, res, ldr, C, ldc, slab, row_ssq, ssq_chunks, eps, M, N, K);
}

template <int EPI>
static int launch_reduce(hipStream_t s, const float *slab, int ksg, const float *bias, const bf16_t *res, int ldr, bf16_t *C, int ldc,
                         const bf16_t *normw, bf16_t *XS, int ldn, float *ssq, const float *row_ssq, int ssq_chunks, float eps, int K, int M, int N) {
    hipLaunchKernelGGL((rows64_reduce_kernel<EPI>), dim3(M, cdiv(N, 1024)), dim3(256), 0, s, slab, ksg, bias, res, ldr, C, ldc, normw, XS, ldn, ssq,
                       row_ssq, ssq_chunks, eps, K, M, N);
    return LL_OK;
}

int linear_rows64_launch(const void *x, int ldx, const void *Wp, const float *bias, const void *residual, int ldr, void *out, int ldc, int M,
                         int N, int K, int epi, const float *row_ssq, int ssq_chunks, float eps, const void *next_norm_w, void *scaled_out, int ldn,
                         float *ssq_out, void *workspace, size_t workspace_bytes, hipStream_t s) {
    LL_CHECK(x && Wp && out, "ll_linear_rows64_bf16: null argument");
    LL_CHECK(M >= 1 && M <= 64, "ll_linear_rows64_bf16: M=%d rows (1..64)", M);
    LL_CHECK(N >= 1 && K >= 32 && K % 32 == 0 && ldx % 8 == 0, "ll_linear_rows64_bf16: K must be a multiple of 32, ldx of 8");
    LL_CHECK(epi >= R64_PLAIN && epi <= R64_SILU_MUL, "ll_linear_rows64_bf16: epilogue %d", epi);
    LL_CHECK(epi != R64_RESIDUAL || residual, "ll_linear_rows64_bf16: residual epilogue without a residual");
    LL_CHECK(epi != R64_SILU_MUL || N % 16 == 0, "ll_linear_rows64_bf16: SILU_MUL needs N %% 16 == 0 (N=%d)", N);
    LL_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)Wp & 15) == 0 && (!workspace || ((uintptr_t)workspace & 15) == 0),
             "ll_linear_rows64_bf16: operands must be 16-byte aligned");
    LL_CHECK(!row_ssq || ssq_chunks >= 1, "ll_linear_rows64_bf16: row_ssq needs ssq_chunks >= 1");
    const bool norm = next_norm_w != nullptr;
    LL_CHECK(!norm || (scaled_out && ssq_out && epi != R64_SILU_MUL && ldn >= N),
             "ll_linear_rows64_bf16: the output pre-norm needs scaled_out, ssq_out and a plain / residual epilogue");
    const bf16_t *X = (const bf16_t *)x, *W = (const bf16_t *)Wp, *rs = (const bf16_t *)residual;
    bf16_t *C = (bf16_t *)out;
    int ksg = 1;
    if (epi != R64_SILU_MUL) {          // the SiLU needs complete sums
        if (g_rows64_ksg) {
            ksg = g_rows64_ksg;
        } else {
            // few row groups (o_proj, down_proj: 56..64 on 256 CUs; q|k|v: 72..96): split K over workgroups until three quarters of the CUs
            // have a unit; a slice keeps >= 1-2 x-stages (profiles/r6_rows64_sweep.txt: o_proj 16.0 us at 4 slices against 24.3 in one piece,
            // down_proj 29.8 against 63.8, q|k|v 18.9 at 2 against 21.9)
            const int groups = cdiv(N, 64), nst = cdiv(K, 512);
            while (ksg < 8 && groups * ksg * 4 < r64_cus() * 3 && nst / (ksg * 2) >= 1) ksg *= 2;      // Qwen2-7B o_proj (K = 3584: 7 stages): 15.0 us at 4 slices, 17.0 at 2
        }
    }
    const size_t need = (size_t)ksg * M * N * 4;
    const bool two = ksg > 1 || norm;
    if (two && (!workspace || workspace_bytes < need)) {
        LL_CHECK(!norm, "ll_linear_rows64_bf16: the output pre-norm needs %zu bytes of workspace", need);
        ksg = 1;
    }
    float *slab = (ksg > 1 || norm) ? (float *)workspace : nullptr;
    const int mb = M <= 32 ? 2 : 4;
#define LL_R64(EPI_)                                                                                                                     \
    do {                                                                                                                                 \
        if (mb == 2) LL_TRY((launch_rows64_mb<EPI_, 2>(ksg, s, X, ldx, W, bias, rs, ldr, C, ldc, slab, slab ? nullptr : row_ssq, ssq_chunks, eps, M, N, K))); \
        else LL_TRY((launch_rows64_mb<EPI_, 4>(ksg, s, X, ldx, W, bias, rs, ldr, C, ldc, slab, slab ? nullptr : row_ssq, ssq_chunks, eps, M, N, K)));        \
    } while (0)
    if (epi == R64_PLAIN) LL_R64(R64_PLAIN);
    else if (epi == R64_RESIDUAL) LL_R64(R64_RESIDUAL);
    else LL_R64(R64_SILU_MUL);
#undef LL_R64
    if (slab) {
        if (epi == R64_RESIDUAL)
            LL_TRY((launch_reduce<R64_RESIDUAL>(s, slab, ksg, bias, rs, ldr, C, ldc, (const bf16_t *)next_norm_w, (bf16_t *)scaled_out, ldn, ssq_out,
                                                row_ssq, ssq_chunks, eps, K, M, N)));
        else
            LL_TRY((launch_reduce<R64_PLAIN>(s, slab, ksg, bias, rs, ldr, C, ldc, (const bf16_t *)next_norm_w, (bf16_t *)scaled_out, ldn, ssq_out, row_ssq,
                                             ssq_chunks, eps, K, M, N)));
    }
    LL_LAUNCH_CHECK();
    return LL_OK;
}

}  // namespace ll

using namespace ll;

extern "C" {

int64_t ll_rows64_packed_elems(int N, int K) { return (N < 1 || K < 32 || K % 32) ? -1 : (int64_t)((N + 15) / 16) * 16 * K; }

int ll_rows64_pack_bf16(const void *W, int ldw, int N, int K, void *packed, void *stream) {
    LL_CHECK(W && packed && N >= 1 && K >= 32 && K % 32 == 0 && ldw % 8 == 0 && ldw >= K, "ll_rows64_pack_bf16: K must be a multiple of 32, ldw of 8");
    LL_CHECK(((uintptr_t)W & 15) == 0 && ((uintptr_t)packed & 15) == 0, "ll_rows64_pack_bf16: operands must be 16-byte aligned");
    const int64_t pieces = ll_rows64_packed_elems(N, K) / 8;
    hipLaunchKernelGGL(rows64_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)W, ldw,
                       (bf16_t *)packed, N, K);
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int ll_linear_rows64_bf16(const void *x, int ldx, const void *Wp, const float *bias, const void *residual, int ldr, void *out, int ldc, int M,
                          int N, int K, int epi, const float *row_ssq, int ssq_chunks, float eps, const void *next_norm_w, void *scaled_out,
                          int ldn, float *ssq_out, void *workspace, int64_t workspace_bytes, void *stream) {
    return linear_rows64_launch(x, ldx, Wp, bias, residual, ldr, out, ldc, M, N, K, epi, row_ssq, ssq_chunks, eps, next_norm_w, scaled_out, ldn,
                                ssq_out, workspace, workspace_bytes > 0 ? (size_t)workspace_bytes : 0, (hipStream_t)stream);
}

int ll_rows64_ssq_chunks(int N) { return N >= 1 ? (N + 1023) / 1024 : 0; }

int ll_rows64_prenorm_bf16(const void *x, int ldx, const void *norm_w, void *scaled_out, int ldn, float *ssq_out, int M, int N, void *stream) {
    LL_CHECK(x && norm_w && scaled_out && ssq_out && M >= 1 && N >= 1 && ldn >= N, "ll_rows64_prenorm_bf16: bad argument");
    // no slabs, no output row: XS = bf16(x * norm_w), ssq = per-chunk sums of squares of x
    LL_TRY((launch_reduce<R64_RESIDUAL>((hipStream_t)stream, nullptr, 0, nullptr, (const bf16_t *)x, ldx, nullptr, 0, (const bf16_t *)norm_w,
                                        (bf16_t *)scaled_out, ldn, ssq_out, nullptr, 0, 0.f, 0, M, N)));
    LL_LAUNCH_CHECK();
    return LL_OK;
}

int64_t ll_linear_rows64_workspace_bytes(int M, int N) { return (int64_t)8 * (M > 0 ? M : 0) * (N > 0 ? N : 0) * 4; }

#if LL_TUNING
int ll_set_rows64_ksplit(int ksg) {
    const int old = g_rows64_ksg;
    g_rows64_ksg = (ksg >= 1 && ksg <= 8) ? ksg : 0;
    return old;
}
#endif

// Times ll_linear_rows64_bf16 on synthetic operands over `nweights` distinct packed weight matrices (defeats the Infinity Cache);
// norm & 1: with the output pre-norm (two launches); norm & 2: with the input row scale.
#if LL_TUNING
int ll_rows64_bench(int M, int N, int K, int epi, int norm, int iters, int nweights, float *ms) {
    LL_CHECK(ms && iters > 0 && nweights > 0 && M >= 1 && M <= 64, "bad argument");
    const int rowsW = epi == R64_SILU_MUL ? 2 * N : N;
    const size_t welems = (size_t)ll_rows64_packed_elems(rowsW, K);
    bf16_t *X = nullptr, *W = nullptr, *C = nullptr, *R = nullptr, *XN = nullptr;
    float *ws = nullptr, *SS = nullptr;
    const size_t wsb = (size_t)ll_linear_rows64_workspace_bytes(M, N);
    LL_HIP(hipMalloc(&SS, (size_t)64 * 64 * 4));
    LL_HIP(hipMemset(SS, 0x3c, (size_t)64 * 64 * 4));
    LL_HIP(hipMalloc(&X, (size_t)64 * K * 2));
    LL_HIP(hipMalloc(&W, (size_t)nweights * welems * 2));
    LL_HIP(hipMalloc(&C, (size_t)64 * N * 2));
    LL_HIP(hipMalloc(&XN, (size_t)64 * N * 2));
    LL_HIP(hipMalloc(&R, (size_t)64 * N * 2));
    LL_HIP(hipMalloc(&ws, wsb));
    LL_HIP(hipMemset(X, 0x11, (size_t)64 * K * 2));
    LL_HIP(hipMemset(R, 0x11, (size_t)64 * N * 2));
    LL_HIP(hipMemset(W, 0x11, (size_t)nweights * welems * 2));
    hipStream_t st;
    LL_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    LL_HIP(hipEventCreate(&e0));
    LL_HIP(hipEventCreate(&e1));
    int rc = LL_OK;
    for (int pass = 0; pass < 2 && rc == LL_OK; ++pass) {
        if (pass == 1) (void)hipEventRecord(e0, st);
        for (int i = 0; i < (pass ? iters : nweights) && rc == LL_OK; ++i)
            rc = linear_rows64_launch(X, K, W + (size_t)(i % nweights) * welems, nullptr, R, N, C, N, M, N, K, epi, (norm & 2) ? SS : nullptr, 4, 1e-6f,
                                      ((norm & 1) && epi != R64_SILU_MUL) ? X : nullptr, XN, N, SS, ws, wsb, st);
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
    (void)hipFree(XN);
    (void)hipFree(R);
    (void)hipFree(ws);
    (void)hipFree(SS);
    if (rc != LL_OK) return rc;
    LL_HIP(he);
    return LL_OK;
}
#endif

}  // extern "C"
