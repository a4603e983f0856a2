// Next-token sampler of the LLM decode loop as one launch (two for top-k at a few rows): HF TemperatureLogitsWarper +
// TopPLogitsWarper + softmax + multinomial (transformers generation/logits_process.py; the reference calls them through
// language_model.generate(do_sample, temperature, top_p), modeling_llamole.py:599/:849) plus the loop's bookkeeping
// (pad rows that already stopped, EOS test, write the token into the output buffer and the next step's input, advance the
// position counters).  The op-by-op PyTorch version is ~25 launches per token incl. a full sort of the vocabulary
// (~0.2 ms at V = 152 k); here one workgroup per row keeps the whole row of bf16 logits in registers:
//   1. order-preserving 16-bit keys, row maximum;
//   2. an EXACT count histogram in LDS over the 36 864 keys below the maximum (bf16 logits take few distinct values: one
//      bin per value; integer atomics, so the result is deterministic and order-independent); keys further down (more than
//      288 binades away: mass 0 for any practical temperature) are summed into one tail mass;
//   3. per non-empty bin mass = count * exp(l/T - max) in 2^-40 fixed point; descending scan -> the lowest value whose
//      mass-above is < top_p * Z = the nucleus boundary, and the kept mass M;
//   4. one Philox draw R in [0, M) -> the sampled VALUE and a rank among the tokens sharing it (equal logits = equal
//      probability);  5. the rank-th such token.
// exp is evaluated per distinct value (a few thousand), not per token; the per-token work is integer compares.
// Nucleus semantics: a value v is kept iff the probability mass strictly above v is < top_p (HF: ascending cumulative sum
// > 1 - top_p); tokens tied with the boundary value are all kept (torch.sort leaves their order unspecified).
// Two shortcuts under top-k, each producing the same token as the whole-row count for every seed (tests/test_llm_sampler_gpu.py):
//   * sample_token_kernel counts only keys >= a lower bound of the row's k-th largest key (at least k threads hold such a key);
//   * with a workspace and <= 4 rows, sample_candidates_kernel (V/2048 workgroups per row) first hands on the tokens that can be
//     among the k largest, and sample_token_kernel finishes on that list (sample_fast) instead of reading the row.
#include "common.h"

namespace ll {

struct SampleArgs {
    const bf16_t *logits;      // [B, V] bf16, row stride ld
    int64_t ld;
    int V;
    float inv_temp;            // 1 / temperature (f32, as PyTorch's tensor / python-scalar computes it)
    float top_p;
    int top_k;                 // HF TopKLogitsWarper ahead of top-p: keep the k highest logits (0 = off); ties with the k-th are kept
    int greedy;
    const long long *seed;     // device, [1]
    const long long *eos;      // device, [n_eos] (-1 = unused slot)
    int n_eos;
    long long pad;
    unsigned char *done;       // [B]
    long long *tok;            // [B]   next step's input ids
    long long *out_tokens;     // [B, max_new], row stride ld_out, column step[b]
    int64_t ld_out;
    int max_new;
    long long *step;           // [B]
    long long *posid;          // [B] or null
    long long *pos;            // [1] or null
    int advance;               // also advance posid / pos (decode-loop use)
    unsigned long long *dbg;   // optional [B,4]: Z, kept mass, boundary key, sampled key
    unsigned int *cand_total;  // optional [B,4]: row header of sample_candidates_kernel -- count, largest key, 0xffff - smallest key (reset to 0 here)
    const uint2 *cand;         // [B, SAMPLE_CAND_CAP] (key, token index)
};

__device__ __forceinline__ uint32_t key_of(uint32_t x) {   // x: raw bf16 bits (16 low bits), NaN -> 0, inf -> +-max
    const uint32_t ex = (x >> 7) & 0xffu;
    if (ex == 0xffu) x = (x & 0x7fu) ? 0u : ((x & 0x8000u) | 0x7f7fu);
    return (x & 0x8000u) ? (~x & 0xffffu) : (x | 0x8000u);
}
// both halves of a packed pair of bf16 -> packed pair of keys
__device__ __forceinline__ uint32_t keys_of_pair(uint32_t wv) {
    uint32_t lo = wv & 0xffffu, hi = wv >> 16;
    const uint32_t alo = lo & 0x7fffu, ahi = hi & 0x7fffu;
    lo = alo > 0x7f80u ? 0u : (alo == 0x7f80u ? lo - 1u : lo);       // NaN -> 0, +-inf -> +-max finite
    hi = ahi > 0x7f80u ? 0u : (ahi == 0x7f80u ? hi - 1u : hi);
    const uint32_t x = lo | (hi << 16);
    const uint32_t msk = (((x >> 15) & 0x00010001u) * 0xffffu) | 0x80008000u;   // negative: flip all bits; else set the sign bit
    return x ^ msk;
}
typedef short pk_i16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
// keys of a pair WITHOUT the inf / NaN canonicalisation: valid when neither half has an all-ones exponent (pair_special == 0)
__device__ __forceinline__ uint32_t keys_of_finite_pair(uint32_t wv) {
    const pk_i16 sg = __builtin_bit_cast(pk_i16, wv) >> 15;                     // 0xffff per negative half
    return wv ^ (__builtin_bit_cast(uint32_t, sg) | 0x80008000u);                // negative: flip all bits; else set the sign bit
}
// bit 15 / 31 set iff that half is inf or NaN (0x7f80 + 0x0080 carries into the half's top bit, never beyond it)
__device__ __forceinline__ uint32_t pair_special(uint32_t wv) { return (wv & 0x7f807f80u) + 0x00800080u; }
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_u16, a), __builtin_bit_cast(pk_u16, b)));
}
// a value every lane of the wave holds alike -> scalar registers (the sampler is short of vector registers, not of scalar ones)
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)), lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ float val_of(uint32_t k) {
    const uint32_t x = (k & 0x8000u) ? (k & 0x7fffu) : (~k & 0xffffu);
    return __uint_as_float(x << 16);
}
__device__ __forceinline__ float scaled(uint32_t k, float inv_temp) {
    return fminf(fmaxf(val_of(k) * inv_temp, -3.0e38f), 3.0e38f);
}
__device__ __forceinline__ unsigned long long mass_of(uint32_t k, float inv_temp, float m) {
    return (unsigned long long)(__expf(scaled(k, inv_temp) - m) * 1099511627776.0f);   // 2^40 fixed point
}

// Opaque to the optimiser: stops it from unpacking all keys once and keeping 2x the registers live across the passes.
template <int CH> __device__ __forceinline__ void reg_fence(uint32_t (&w)[CH][4]) {
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(w[k][t]));
}

constexpr int SAMPLE_BPT = 36;                      // histogram bins per thread
constexpr int SAMPLE_W = SAMPLE_BPT * 1024;         // key window below the row maximum held in LDS (147 KB of counters)
constexpr int SAMPLE_TM_BINS = SAMPLE_W / 32;       // 1152
constexpr int SAMPLE_CAND_CAP = 16384;              // candidates per row the split sampler keeps (more: the one-workgroup path)
constexpr int SAMPLE_SPLIT_MAX_ROWS = 4;            // more rows already occupy as many CUs in the one-workgroup-per-row path (8 rows: 22.2 vs 21.2 us, 64: 41 vs 22)
constexpr int SAMPLE_SPLIT_MAX_K = 128;             // ~1.15 k candidates per 2048 tokens: 128 x 1.15 x 80 workgroups < the cap

// First half of the split top-k sampler: 256 threads x 8 tokens per workgroup, many workgroups per row.  Every workgroup hands on the
// tokens that can be among the row's k largest: those with key >= klo_w, where at least k of the workgroup's THREADS hold such a key (so
// the row's k-th largest key is >= klo_w, and a token of this workgroup below klo_w cannot survive TopKLogitsWarper).  klo_w = the
// smallest, over the four waves, of the ceil(k/4)-th largest thread maximum of the wave (64-lane bitonic sort).  Candidates
// (key, token index) are appended to the row's list in the workspace; the list's order is arbitrary and nothing downstream depends on it.
__global__ __launch_bounds__(256) void sample_candidates_kernel(const bf16_t *__restrict__ logits, int64_t ld, int V, int top_k,
                                                                unsigned int *__restrict__ hdr, uint2 *__restrict__ cand) {
    __shared__ uint32_t list[2048];
    __shared__ uint32_t sh_b[4];
    __shared__ unsigned int sh_n, sh_base, sh_mx, sh_mninv;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = V / 8, first = blockIdx.x * 256;
    // the last workgroup of a row looks at the row's LAST 256 chunks (a full set of thread maxima for the bound: the tokens it shares
    // with its neighbour are tokens of the row all the same) and hands on only the ones from its own range
    const int start = (first + 256 > nchunk && nchunk >= 256) ? nchunk - 256 : first;
    const int c = start + tid;
    const bool own = c >= first;
    uint32_t k[4] = {0u, 0u, 0u, 0u};
    if (c < nchunk) {
        const uint4 v = *reinterpret_cast<const uint4 *>(logits + (int64_t)b * ld + (int64_t)c * 8);
        const uint32_t r[4] = {v.x, v.y, v.z, v.w};
        const bool special = ((pair_special(r[0]) | pair_special(r[1]) | pair_special(r[2]) | pair_special(r[3])) & 0x80008000u) != 0u;
#pragma unroll
        for (int t = 0; t < 4; ++t) k[t] = special ? keys_of_pair(r[t]) : keys_of_finite_pair(r[t]);
    }
    const uint32_t kp = pk_max_u16(pk_max_u16(k[0], k[1]), pk_max_u16(k[2], k[3]));
    uint32_t v = max(kp & 0xffffu, kp >> 16);         // thread maximum (0: no token)
    if (tid == 0) { sh_n = 0; sh_mx = 0; sh_mninv = 0; }
    // ascending bitonic sort of the wave's 64 thread maxima
#pragma unroll
    for (int sz = 2; sz <= 64; sz <<= 1) {
#pragma unroll
        for (int st = sz >> 1; st > 0; st >>= 1) {
            const uint32_t o = __shfl_xor(v, st, 64);
            v = (((lane & sz) == 0) == ((lane & st) == 0)) ? min(v, o) : max(v, o);
        }
    }
    const int kq = (top_k + 3) / 4;                   // 1..64
    const uint32_t bw = __shfl(v, 64 - kq, 64);       // the kq-th largest of this wave
    if (lane == 0) sh_b[wave] = bw;
    __syncthreads();
    const uint32_t klo = max(1u, min(min(sh_b[0], sh_b[1]), min(sh_b[2], sh_b[3])));
    uint32_t mx = 0, mninv = 0;
    if (own) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const uint32_t key = hh ? (k[t] >> 16) : (k[t] & 0xffffu);
                if (key >= klo) {
                    list[atomicAdd(&sh_n, 1u)] = key | ((uint32_t)(tid * 8 + t * 2 + hh) << 16);
                    mx = max(mx, key);
                    mninv = max(mninv, 0xffffu - key);
                }
            }
        }
    }
    if (mx) { atomicMax(&sh_mx, mx); atomicMax(&sh_mninv, mninv); }
    __syncthreads();
    const uint32_t n = sh_n;
    if (tid == 0 && n) {
        // row header: [0] candidates so far, [1] their largest key, [2] 0xffff - their smallest key (all three start from, and are put
        // back to, zero)
        sh_base = atomicAdd(&hdr[b * 4], n);
        atomicMax(&hdr[b * 4 + 1], sh_mx);
        atomicMax(&hdr[b * 4 + 2], sh_mninv);
    }
    __syncthreads();
    const uint32_t base = sh_base;
    uint2 *dst = cand + (int64_t)b * SAMPLE_CAND_CAP;
    for (uint32_t j = tid; j < n; j += 256) {
        if (base + j < (uint32_t)SAMPLE_CAND_CAP) {
            const uint32_t e = list[j];
            dst[base + j] = make_uint2(e & 0xffffu, (uint32_t)start * 8u + (e >> 16));
        }
    }
}

// LDS of the per-row workgroup (1024 threads)
struct SampleLds {
    uint32_t cnt[SAMPLE_W];              // cnt[d] = number of tokens whose key is kmax - d
    unsigned long long wsum[16];
    float redf[16];
    unsigned int redu[16];
    unsigned long long tail, R, Zk;
    unsigned int d, key, rank, tok, dk, klo, nm;
    uint32_t tmh[SAMPLE_TM_BINS + 1];    // thread maxima per 32-key bin below the row maximum (top-k lower bound)
    long long eos[64];                   // the first EOS ids, the row's step counter and stop flag: requested at kernel start
    long long step;
    unsigned int done;
};

// Steps 3 and 4 on the histogram L.cnt[0 .. kmax - klo]: top-k cut, nucleus boundary, one Philox draw -> the sampled VALUE k2 and the rank
// of the token among those sharing it.  Expects a barrier behind the last histogram update.
__device__ __forceinline__ void nucleus_scan(SampleLds &L, const SampleArgs &a, int b, int tid, int lane, int wave, uint32_t kmax,
                                             uint32_t klo, float m, uint32_t &k2, uint32_t &rank) {
    // ---- 3. thread t owns d in [bpt t, bpt t + bpt), descending values: masses = count * 2^-40 fixed-point exp.  All 36 bins per
    // thread without the bound; with it only [0, kmax - klo] can be non-empty (typically one bin per thread)
    const int bpt = klo ? (int)((kmax - klo) >> 10) + 1 : SAMPLE_BPT;
    const int d0 = tid * bpt;
    unsigned long long lsum = 0;
    for (int i = 0; i < bpt; ++i) {
        const uint32_t c = L.cnt[d0 + i];
        if (c && (uint32_t)(d0 + i) < kmax) lsum += (unsigned long long)c * mass_of(kmax - (d0 + i), a.inv_temp, m);
    }
    unsigned long long inc = lsum;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const unsigned long long o = __shfl_up(inc, dd, 64);
        if (lane >= dd) inc += o;
    }
    if (lane == 63) L.wsum[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, Z = L.tail;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < wave) base += L.wsum[i];
        Z += L.wsum[i];
    }
    base = uniform64(base);
    Z = uniform64(Z);
    const unsigned long long excl = base + inc - lsum;
    // top-k ahead of top-p (HF order: temperature, TopKLogitsWarper, TopPLogitsWarper; the reference's GeneratingArguments
    // default top_k = 50): the cut is the value of the k-th largest token, tokens tied with it stay (HF removes
    // logits < kth value); the nucleus is then taken over the renormalised survivors, i.e. against their mass Zk
    uint32_t dk = 0xffffffffu;
    unsigned long long Zk = Z;
    if (a.top_k > 0) {
        uint32_t lc = 0;
        for (int i = 0; i < bpt; ++i) {
            const uint32_t c = L.cnt[d0 + i];
            if (c && (uint32_t)(d0 + i) < kmax) lc += c;
        }
        uint32_t incc = lc;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = __shfl_up(incc, dd, 64);
            if (lane >= dd) incc += o;
        }
        if (lane == 63) L.redu[wave] = incc;
        __syncthreads();
        uint32_t basec = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < wave) basec += L.redu[i];
        const uint32_t exclc = basec + incc - lc;
        const uint32_t kk = (uint32_t)a.top_k;
        if (exclc < kk && exclc + lc >= kk) {        // exactly one thread: its bins hold the k-th largest token
            uint32_t run = exclc;
            unsigned long long A = excl;
            for (int i = 0; i < bpt; ++i) {
                const uint32_t c = L.cnt[d0 + i];
                if (c && (uint32_t)(d0 + i) < kmax) {
                    run += c;
                    A += (unsigned long long)c * mass_of(kmax - (d0 + i), a.inv_temp, m);
                    if (run >= kk) {
                        L.dk = (unsigned int)(d0 + i);
                        L.Zk = A;
                        break;
                    }
                }
            }
        }
        __syncthreads();
        dk = __builtin_amdgcn_readfirstlane(L.dk);                                  // stays ~0 when fewer than k tokens lie in the window: nothing is cut
        if (dk != 0xffffffffu) Zk = uniform64(L.Zk);
    }
    const unsigned long long Tq = a.top_p >= 1.f ? ~0ull : (unsigned long long)((double)a.top_p * (double)Zk);
    // boundary: the lowest value whose mass-above is still < Tq (mass-above is non-decreasing in d)
    {
        unsigned long long A = excl;
        int last = -1;
        for (int i = 0; i < bpt; ++i) {
            const uint32_t c = L.cnt[d0 + i];
            if (c && (uint32_t)(d0 + i) < kmax) {
                if (A < Tq && (uint32_t)(d0 + i) <= dk) last = d0 + i;
                A += (unsigned long long)c * mass_of(kmax - (d0 + i), a.inv_temp, m);
            }
        }
        if (last >= 0) atomicMax(&L.d, (unsigned int)last);
    }
    __syncthreads();
    const uint32_t dtau = __builtin_amdgcn_readfirstlane(L.d);      // 0 when nothing else qualifies: the top value is always kept
    if (dtau >= (uint32_t)d0 && dtau < (uint32_t)(d0 + bpt)) {
        unsigned long long A = excl;
        for (int i = 0; i < bpt; ++i) {
            const uint32_t c = L.cnt[d0 + i];
            if (c && (uint32_t)(d0 + i) < kmax) {
                A += (unsigned long long)c * mass_of(kmax - (d0 + i), a.inv_temp, m);
                if ((uint32_t)(d0 + i) == dtau) break;
            }
        }
        unsigned long long M = A;    // kept mass: everything down to and including the boundary value
        if (a.top_p >= 1.f && dk == 0xffffffffu) M = Z;   // tail included when nothing is filtered
        const unsigned long long sd = (unsigned long long)*a.seed;
        const unsigned long long st = (unsigned long long)a.step[b];
        const uint4 rnd = philox4x32(make_uint4((uint32_t)st, (uint32_t)(st >> 32), (uint32_t)b, 0x5A17u),
                                     make_uint2((uint32_t)sd, (uint32_t)(sd >> 32)));
        const unsigned long long r64 = ((unsigned long long)rnd.x << 32) | rnd.y;
        L.R = __umul64hi(r64, M);
        if (a.dbg) { a.dbg[b * 4 + 0] = Z; a.dbg[b * 4 + 1] = M; a.dbg[b * 4 + 2] = kmax - dtau; }
    }
    if (tid == 0) { L.key = kmax; L.rank = 0; }
    __syncthreads();
    // ---- 4. the value whose mass interval contains R, and the rank among the tokens sharing it
    const unsigned long long R = uniform64(L.R);
    if (R >= excl && R < excl + lsum) {
        unsigned long long A = excl;
        for (int i = 0; i < bpt; ++i) {
            const uint32_t c = L.cnt[d0 + i];
            if (c && (uint32_t)(d0 + i) < kmax) {
                const unsigned long long q = mass_of(kmax - (d0 + i), a.inv_temp, m);
                const unsigned long long ms = (unsigned long long)c * q;
                if (R < A + ms) {
                    unsigned long long r = q ? (R - A) / q : 0;
                    if (r >= c) r = c - 1;
                    L.key = kmax - (d0 + i);
                    L.rank = (uint32_t)r;
                    break;
                }
                A += ms;
            }
        }
    }
    __syncthreads();
    k2 = __builtin_amdgcn_readfirstlane(L.key);      // (R in the tail mass, possible only with top_p >= 1 and a > 288-binade spread: the top value)
    rank = __builtin_amdgcn_readfirstlane(L.rank);
    if (a.dbg && tid == 0) a.dbg[b * 4 + 3] = k2;
}

// Requested at kernel start, used by sample_finish behind many barriers: the row's step counter / stop flag and the EOS ids.
__device__ __forceinline__ void sample_prefetch(SampleLds &L, const SampleArgs &a, int b, int tid) {
    if (tid < 64 && tid < a.n_eos) L.eos[tid] = a.eos[tid];
    if (tid == 64) { L.step = a.step[b]; L.done = a.done[b]; }
}

// Loop bookkeeping of the row (after a barrier): pad rows that already stopped, EOS test, token -> output buffer and next input, counters.
__device__ __forceinline__ void sample_finish(SampleLds &L, const SampleArgs &a, int b, int tid) {
    __syncthreads();
    if (tid < 64) {
        const long long t = L.step;
        const long long nxt = L.done ? a.pad : (long long)L.tok;
        bool hit = tid < a.n_eos && L.eos[tid] == nxt;
        for (int i = 64 + tid; i < a.n_eos; i += 64) hit = hit || (a.eos[i] == nxt);
        const bool stop = __any(hit);
        if (tid == 0) {
            if (a.cand_total) *reinterpret_cast<uint4 *>(a.cand_total + b * 4) = make_uint4(0, 0, 0, 0);   // the row's list is consumed
            if (t >= 0 && t < a.max_new) a.out_tokens[(int64_t)b * a.ld_out + t] = nxt;
            a.tok[b] = nxt;
            if (stop) a.done[b] = 1;
            a.step[b] = t + 1;
            if (a.advance) {
                if (a.posid) a.posid[b] += 1;
                if (a.pos && b == 0) a.pos[0] += 1;
            }
        }
    }
}

// Second half of the split top-k sampler: the row's candidates (sample_candidates_kernel) instead of the row.  Every token whose key
// is >= the row's k-th largest key is a candidate, so the histogram bins down to the top-k cut are complete; bins further down may
// miss tokens of other workgroups and nothing reads them (the nucleus boundary lies above the cut).  Returns false -- nothing but LDS
// scratch touched -- when the list overflowed (ties) or spans more than the key window: the caller then reads the row.
template <int CH>
__device__ __forceinline__ bool sample_fast(SampleLds &L, const SampleArgs &a, int b, int tid, int lane, int wave) {
    const uint4 hd = *reinterpret_cast<const uint4 *>(a.cand_total + b * 4);
    const uint32_t ncand = __builtin_amdgcn_readfirstlane(hd.x), kmax = __builtin_amdgcn_readfirstlane(hd.y),
                   klo = 0xffffu - __builtin_amdgcn_readfirstlane(hd.z);
    if (a.greedy || a.top_k <= 0 || ncand < 1u || ncand > (uint32_t)SAMPLE_CAND_CAP) return false;
    if (klo == 0u || kmax < klo || kmax - klo >= (uint32_t)SAMPLE_W) return false;       // (a spread of more than 288 binades among the candidates)
    // the candidates stay in registers: up to 16 per thread = the cap
    const uint2 *cand = a.cand + (int64_t)b * SAMPLE_CAND_CAP;
    uint2 cr[SAMPLE_CAND_CAP / 1024];
#pragma unroll
    for (int i = 0; i < SAMPLE_CAND_CAP / 1024; ++i) {
        const uint32_t j = tid + i * 1024;
        cr[i] = j < ncand ? cand[j] : make_uint2(0u, 0u);                               // key 0 = none
    }
    if (tid == 0) { L.tail = 0; L.d = 0; L.tok = 0; L.dk = 0xffffffffu; L.Zk = 0; L.nm = 0; }
    const int bpt = (int)((kmax - klo) >> 10) + 1;     // the scan's bins per thread: zero exactly those
    for (int i = 0; i < bpt; ++i) L.cnt[tid * bpt + i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SAMPLE_CAND_CAP / 1024; ++i)
        if (cr[i].x) atomicAdd(&L.cnt[kmax - cr[i].x], 1u);
    __syncthreads();
    uint32_t k2 = kmax, rank = 0;
    nucleus_scan(L, a, b, tid, lane, wave, kmax, klo, scaled(kmax, a.inv_temp), k2, rank);
    // the rank-th token with the value k2, counted in the order the one-workgroup path enumerates a row -- thread (chunk mod 1024)
    // first, then chunk / 1024, then the element -- so both paths pick the same token.  The matches' order keys go to the histogram's
    // LDS (free now); each match counts the smaller ones.
#pragma unroll
    for (int i = 0; i < SAMPLE_CAND_CAP / 1024; ++i) {
        if (cr[i].x == k2) {
            const uint32_t c = cr[i].y >> 3;
            L.cnt[atomicAdd(&L.nm, 1u)] = (c & 1023u) * (uint32_t)(CH * 8) + (c >> 10) * 8u + (cr[i].y & 7u);
        }
    }
    __syncthreads();
    const uint32_t nm = L.nm;
    for (uint32_t t = tid; t < nm; t += 1024) {
        const uint32_t o = L.cnt[t];
        uint32_t below = 0;
        for (uint32_t j = 0; j < nm; ++j) below += L.cnt[j] < o ? 1u : 0u;
        if (below == rank) {
            const uint32_t rem = o % (uint32_t)(CH * 8);
            L.tok = ((rem >> 3) * 1024u + o / (uint32_t)(CH * 8)) * 8u + (rem & 7u);
        }
    }
    sample_finish(L, a, b, tid);
    return true;
}

template <int CH>
__global__ __launch_bounds__(1024) void sample_token_kernel(SampleArgs a) {
    __shared__ SampleLds L;
    sample_prefetch(L, a, blockIdx.x, threadIdx.x);
    if (a.cand_total && sample_fast<CH>(L, a, blockIdx.x, threadIdx.x, threadIdx.x & 63, threadIdx.x >> 6)) return;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t *row = a.logits + (int64_t)b * a.ld;
    const int nchunk = a.V / 8;
    uint32_t w[CH][4];   // two 16-bit keys per word; key 0 = no element
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = tid + k * 1024;
        uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);          // past the end of the row: NaN, which converts to key 0 = no element
        if (c < nchunk) v = *reinterpret_cast<const uint4 *>(row + (int64_t)c * 8);
        w[k][0] = v.x; w[k][1] = v.y; w[k][2] = v.z; w[k][3] = v.w;
    }
    if (!a.greedy) {
#pragma unroll
        for (int i = 0; i < SAMPLE_BPT; ++i) L.cnt[i * 1024 + tid] = 0;
    }
    if (tid == 0) { L.tail = 0; L.d = 0; L.tok = 0; L.dk = 0xffffffffu; L.Zk = 0; L.klo = 0; }
    for (int i = tid; i < SAMPLE_TM_BINS + 1; i += 1024) L.tmh[i] = 0;
    reg_fence<CH>(w);
    // keys: the finite conversion for every pair (7 VALU instructions per pair instead of ~25 with the inf / NaN canonicalisation);
    // the inf / NaN test is taken per 16-byte chunk and such a chunk (rare) is converted again; the all-ones fill past the end of the
    // row becomes key 0 = no element
    uint32_t kmaxp = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const uint32_t r0 = w[k][0], r1 = w[k][1], r2 = w[k][2], r3 = w[k][3];
        uint32_t k0 = keys_of_finite_pair(r0), k1 = keys_of_finite_pair(r1), k2_ = keys_of_finite_pair(r2), k3 = keys_of_finite_pair(r3);
        if (((pair_special(r0) | pair_special(r1) | pair_special(r2) | pair_special(r3)) & 0x80008000u) && tid + k * 1024 < nchunk) {
            k0 = keys_of_pair(r0); k1 = keys_of_pair(r1); k2_ = keys_of_pair(r2); k3 = keys_of_pair(r3);
        }
        w[k][0] = k0; w[k][1] = k1; w[k][2] = k2_; w[k][3] = k3;
        kmaxp = pk_max_u16(pk_max_u16(kmaxp, k0), pk_max_u16(pk_max_u16(k1, k2_), k3));
        __builtin_amdgcn_sched_barrier(0);
    }
    const uint32_t kmaxi = max(kmaxp & 0xffffu, kmaxp >> 16);
    float kmaxf = wave_max((float)kmaxi);
    if (lane == 0) L.redf[wave] = kmaxf;
    __syncthreads();
    float km = L.redf[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) km = fmaxf(km, L.redf[i]);
    const uint32_t kmax = __builtin_amdgcn_readfirstlane((uint32_t)km);      // uniform values live in scalar registers
    const float m = scaled(kmax, a.inv_temp);
    uint32_t k2 = kmax;      // the value to pick a token of
    uint32_t rank = 0;
    if (!a.greedy) {
        // ---- 1b. with top-k on, a lower bound of the k-th largest key: at least k THREADS hold a key >= klo, so nothing below klo
        // survives TopKLogitsWarper and it need not be counted.  klo = lower edge of the 32-key bin in which the count of thread
        // maxima, taken from the row maximum down, reaches k (0 = no bound: k > 1024 threads, or the maxima leave the window).
        // The dbg tap reports the mass of the whole row, so it counts everything.
        uint32_t klo = 0;
        if (a.top_k > 0 && a.top_k <= 1024 && !a.dbg) {
            const uint32_t dm = kmax - kmaxi;
            if (kmaxi) atomicAdd(&L.tmh[dm < (uint32_t)SAMPLE_W ? (dm >> 5) : SAMPLE_TM_BINS], 1u);
            __syncthreads();
            const uint32_t c0 = 2 * tid < SAMPLE_TM_BINS ? L.tmh[2 * tid] : 0u, c1 = 2 * tid + 1 < SAMPLE_TM_BINS ? L.tmh[2 * tid + 1] : 0u;
            uint32_t inct = c0 + c1;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const uint32_t o = __shfl_up(inct, dd, 64);
                if (lane >= dd) inct += o;
            }
            if (lane == 63) L.redu[wave] = inct;
            __syncthreads();
            uint32_t baset = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < wave) baset += L.redu[i];
            const uint32_t exclt = baset + inct - c0 - c1, kk = (uint32_t)a.top_k;
            if (exclt < kk && exclt + c0 + c1 >= kk) {
                const uint32_t bin = exclt + c0 >= kk ? 2 * tid : 2 * tid + 1;
                const uint32_t dlow = bin * 32 + 31;                 // the lowest key of that bin is kmax - dlow
                L.klo = dlow < kmax ? kmax - dlow : 1u;
            }
            __syncthreads();
            klo = __builtin_amdgcn_readfirstlane(L.klo);
        }
        // ---- 2. exact count histogram over the key window [kmax - W + 1, kmax]; anything below goes to one tail mass
        reg_fence<CH>(w);
        if (klo) {
            // nothing below klo is counted and the tail mass is not needed (it only enters Z, which top-k replaces by the survivors'
            // mass): one packed compare per pair of keys, the counting itself is rare
            const uint32_t below = (klo - 1u) * 0x00010001u;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t kk = w[k][t];
                    if (pk_max_u16(kk, below) != below) {
                        if ((kk & 0xffffu) >= klo) atomicAdd(&L.cnt[kmax - (kk & 0xffffu)], 1u);
                        if ((kk >> 16) >= klo) atomicAdd(&L.cnt[kmax - (kk >> 16)], 1u);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const uint32_t key = hh ? (w[k][t] >> 16) : (w[k][t] & 0xffffu);
                        const uint32_t d = kmax - key;
                        if (key) {
                            if (d < (uint32_t)SAMPLE_W) atomicAdd(&L.cnt[d], 1u);
                            else {
                                const unsigned long long q = mass_of(key, a.inv_temp, m);
                                if (q) atomicAdd(&L.tail, q);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        nucleus_scan(L, a, b, tid, lane, wave, kmax, klo, m, k2, rank);
    }
    // ---- 5. the rank-th token whose key is k2 (greedy: the lowest index holding the maximum, like torch.argmax)
    reg_fence<CH>(w);
    uint32_t cntm = 0, minidx = 0xffffffffu;
    uint32_t tid_a = tid;                       // opaque copies: keep the 4*CH token indices from being hoisted and
    asm volatile("" : "+v"(tid_a));             // held live across the two passes below
    const uint32_t k2x2 = k2 * 0x00010001u;
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t x = w[k][t] ^ k2x2;                    // a zero half = a token with the value k2
            if ((x - 0x00010001u) & ~x & 0x80008000u) {           // some half may be zero (never misses one): look closer, rarely
                const uint32_t base_idx = (tid_a + k * 1024) * 8 + t * 2;
                if ((x & 0xffffu) == 0u) { ++cntm; minidx = min(minidx, base_idx); }
                if ((x >> 16) == 0u) { ++cntm; minidx = min(minidx, base_idx + 1); }
            }
        }
    if (a.greedy) {
        float mi = -(float)minidx;            // indices < 2^24: exact in f32
        mi = wave_max(mi);
        if (lane == 0) L.redf[wave] = mi;
        __syncthreads();
        if (tid == 0) {
            float best = L.redf[0];
            for (int i = 1; i < 16; ++i) best = fmaxf(best, L.redf[i]);
            L.tok = (uint32_t)(-best);
        }
    } else {
        uint32_t inc = cntm;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += o;
        }
        if (lane == 63) L.redu[wave] = inc;
        __syncthreads();
        uint32_t base = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < wave) base += L.redu[i];
        const uint32_t excl = base + inc - cntm;
        if (cntm && rank >= excl && rank < excl + cntm) {
            uint32_t left = rank - excl, found = 0xffffffffu;
            reg_fence<CH>(w);
            uint32_t tid_b = tid;
            asm volatile("" : "+v"(tid_b));
#pragma unroll
            for (int k = 0; k < CH; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t base_idx = (tid_b + k * 1024) * 8 + t * 2;
                    if ((w[k][t] & 0xffffu) == k2) { if (left == 0 && found == 0xffffffffu) found = base_idx; --left; }
                    if ((w[k][t] >> 16) == k2) { if (left == 0 && found == 0xffffffffu) found = base_idx + 1; --left; }
                }
            L.tok = found;
        }
    }
    sample_finish(L, a, b, tid);
}

}  // namespace ll

using namespace ll;

extern "C" int ll_sample_token_topk_ws_bf16(const void *logits, int64_t ld, int B, int V, float inv_temp, float top_p, int top_k, int greedy,
                                            const int64_t *seed, const int64_t *eos, int n_eos, int64_t pad, void *done, int64_t *tok,
                                            int64_t *out_tokens, int64_t ld_out, int max_new, int64_t *step, int64_t *posid,
                                            int64_t *pos, int advance, uint64_t *dbg, void *workspace, int64_t workspace_bytes, void *stream);

extern "C" int ll_sample_token_bf16(const void *logits, int64_t ld, int B, int V, float inv_temp, float top_p, int greedy,
                                    const int64_t *seed, const int64_t *eos, int n_eos, int64_t pad, void *done, int64_t *tok,
                                    int64_t *out_tokens, int64_t ld_out, int max_new, int64_t *step, int64_t *posid,
                                    int64_t *pos, int advance, uint64_t *dbg, void *stream) {
    return ll_sample_token_topk_ws_bf16(logits, ld, B, V, inv_temp, top_p, 0, greedy, seed, eos, n_eos, pad, done, tok, out_tokens, ld_out,
                                        max_new, step, posid, pos, advance, dbg, nullptr, 0, stream);
}

extern "C" int ll_sample_token_topk_bf16(const void *logits, int64_t ld, int B, int V, float inv_temp, float top_p, int top_k, int greedy,
                                         const int64_t *seed, const int64_t *eos, int n_eos, int64_t pad, void *done, int64_t *tok,
                                         int64_t *out_tokens, int64_t ld_out, int max_new, int64_t *step, int64_t *posid,
                                         int64_t *pos, int advance, uint64_t *dbg, void *stream) {
    return ll_sample_token_topk_ws_bf16(logits, ld, B, V, inv_temp, top_p, top_k, greedy, seed, eos, n_eos, pad, done, tok, out_tokens, ld_out,
                                        max_new, step, posid, pos, advance, dbg, nullptr, 0, stream);
}

extern "C" int64_t ll_sample_workspace_bytes(int B) { return B <= 0 ? 0 : (int64_t)B * (16 + (int64_t)SAMPLE_CAND_CAP * 8); }

extern "C" int ll_sample_token_topk_ws_bf16(const void *logits, int64_t ld, int B, int V, float inv_temp, float top_p, int top_k, int greedy,
                                            const int64_t *seed, const int64_t *eos, int n_eos, int64_t pad, void *done, int64_t *tok,
                                            int64_t *out_tokens, int64_t ld_out, int max_new, int64_t *step, int64_t *posid,
                                            int64_t *pos, int advance, uint64_t *dbg, void *workspace, int64_t workspace_bytes, void *stream) {
    LL_CHECK(logits && seed && done && tok && out_tokens && step && (n_eos == 0 || eos), "ll_sample_token_bf16: null argument");
    LL_CHECK(B >= 1 && V >= 8 && V % 8 == 0 && V <= 1024 * 8 * 20 && ld % 8 == 0,
             "ll_sample_token_bf16: vocabulary %d must be a multiple of 8 and <= 163840", V);
    LL_CHECK(greedy || (inv_temp > 0.f && top_p >= 0.f), "ll_sample_token_bf16: temperature and top_p must be positive");
    LL_CHECK(!workspace || (workspace_bytes >= ll_sample_workspace_bytes(B) && ((uintptr_t)workspace & 15) == 0),
             "ll_sample_token_topk_ws_bf16: workspace of %lld bytes, need %lld (16-byte aligned)", (long long)workspace_bytes,
             (long long)ll_sample_workspace_bytes(B));
    SampleArgs a;
    a.logits = (const bf16_t *)logits; a.ld = ld; a.V = V; a.inv_temp = inv_temp; a.top_p = top_p; a.top_k = top_k < 0 ? 0 : top_k; a.greedy = greedy;
    a.seed = (const long long *)seed; a.eos = (const long long *)eos; a.n_eos = n_eos; a.pad = pad;
    a.done = (unsigned char *)done; a.tok = (long long *)tok; a.out_tokens = (long long *)out_tokens; a.ld_out = ld_out;
    a.max_new = max_new; a.step = (long long *)step; a.posid = (long long *)posid; a.pos = (long long *)pos;
    a.advance = advance; a.dbg = (unsigned long long *)dbg;
    a.cand_total = nullptr; a.cand = nullptr;
    hipStream_t s = (hipStream_t)stream;
    // sampling a few rows with top-k of at most 128 and a workspace: many workgroups pick the row's candidates, one workgroup per row finishes on them
    // (the same token as the one-workgroup path, which still takes over when a row has more than 16384 candidates)
    if (workspace && !greedy && a.top_k >= 1 && a.top_k <= SAMPLE_SPLIT_MAX_K && !dbg && B <= SAMPLE_SPLIT_MAX_ROWS) {
        a.cand_total = (unsigned int *)workspace;
        a.cand = (const uint2 *)((char *)workspace + (size_t)B * 16);
        hipLaunchKernelGGL(sample_candidates_kernel, dim3(cdiv(V / 8, 256), B), dim3(256), 0, s, (const bf16_t *)logits, ld, V, a.top_k,
                           a.cand_total, (uint2 *)a.cand);
    }
    const int per = cdiv(V, 8 * 1024);
    if (per <= 2) hipLaunchKernelGGL((sample_token_kernel<2>), dim3(B), dim3(1024), 0, s, a);
    else if (per <= 8) hipLaunchKernelGGL((sample_token_kernel<8>), dim3(B), dim3(1024), 0, s, a);
    else if (per <= 16) hipLaunchKernelGGL((sample_token_kernel<16>), dim3(B), dim3(1024), 0, s, a);
    else if (per <= 19) hipLaunchKernelGGL((sample_token_kernel<19>), dim3(B), dim3(1024), 0, s, a);     // 152 064 (Qwen2): 76 key registers
    else hipLaunchKernelGGL((sample_token_kernel<20>), dim3(B), dim3(1024), 0, s, a);
    LL_LAUNCH_CHECK();
    return LL_OK;
}
