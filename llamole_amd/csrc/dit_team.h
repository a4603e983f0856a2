// GraphDiT reverse diffusion as ONE persistent launch: a graph-stationary trajectory kernel for MI355X (gfx950).  Included by graphdit.hip.
//
// The launch chain (graphdit.hip: denoise_body) spends a reverse step in ~200 strictly dependent launches that each span the chip.  The
// graphs of a batch are independent (diffusion_model.py:279-289) and everything inside a block except attention is row-local, so this
// kernel turns the parallelisation around: the 32 CUs of ONE XCD (a "team", formed from HW_REG_XCC_ID at run time) take ONE graph -- its
// conditional and unconditional sequence, 2 x 32 token rows -- through all L blocks, the output layer, the posterior / sampling kernel and
// the next step's embedding, for all T steps, without ever talking to another XCD.  What that buys:
//   * hand-offs inside a team go through the XCD's own L2: producer = plain stores + s_waitcnt vmcnt(0), one relaxed atomic on a
//     counter, consumer = sc1 (L1-bypassing) loads.  No buffer_wbl2 / buffer_inv, ~1.1 us per barrier (tools/team_probe.hip), and the
//     64-row activation panel of the next phase is an L2 hit instead of an Infinity-Cache fetch;
//   * eight graphs advance at once, each at the speed of its own dependent-phase chain;
//   * weights never wait for a phase boundary: every CU has four MFMA waves that own a 32-deep FIFO of weight fragments in registers
//     (1 KB per fragment, non-temporal loads from the pack_mfma16 copies) which runs ~128 KB per CU ahead of the multiplications,
//     across phases, blocks and steps.  Those waves touch no other global memory (an in-order vmcnt would make any store of theirs wait
//     for the whole FIFO); the other four waves of the workgroup ("I/O waves") stage panels, run attention / AdaLN / sampling,
//     copy the output tiles the MFMA waves leave in LDS to global memory and do the team synchronisation.
// The price: every XCD streams every weight (8 x 705 MB per step through ~7.2 TB/s of L2-miss bandwidth, tools/team_probe.hip) --
// the floor of this design is ~0.78 ms per step for 8..64 rows per XCD, against ~1.5 ms for the chain at batch 8.
//
// Reference: Transformer.forward transformer.py:93-187, Block.forward :132-145, Attention layers.py:56-87, sample_p_zs_given_zt
// diffusion_model.py:309-399.  Same arithmetic as the launch chain (bf16 operands, f32 accumulation, the same rounding points:
// q|k|v, attention output, GELU(fc1) and the GEMM operand copy of the residual stream are bf16); sums over K run in another order.
#pragma once
#include "dit_kernels.h"

namespace ll {
namespace team {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TEAM = 32;          // CUs (workgroups) per team = CUs per XCD
constexpr int THREADS = 512;      // 4 MFMA waves + 4 I/O waves
constexpr int NP = 32;            // token rows per sequence in the panel (graphs of <= 32 nodes)
constexpr int ROWS = 2 * NP;      // conditional + unconditional sequence of one graph
constexpr int FLAG_BYTES = 4096;  // LDS behind the panel: counters, bias rows

struct Ctl {                       // device memory, zeroed by the host before every launch
    unsigned int census[8][32];    // [xcc][0]: workgroups that reported for the team of this XCC (own 128-byte line each)
    unsigned int bar[8][32];       // [xcc][0]: monotonic team barrier counter
    unsigned int error;            // bit 0: a bounded spin ran out; bit 1: more than TEAM workgroups on one XCC
    unsigned int teams_done;
    unsigned long long stamps[64]; // LL_TEAM_PROBE: wall-clock ticks (100 MHz) of I/O wave 0 of rank 0 of team 0 through block 3 of the first step
};
#ifdef LL_TEAM_PROBE
#define LL_TSTAMP(i) do { if (w.xcc == 0 && w.rank == 0 && w.iw == 0 && w.lane == 0 && l == 3 && step == 0) a.ctl->stamps[i] = wall_clock64(); } while (0)
#else
#define LL_TSTAMP(i) do { } while (0)
#endif

struct Args {
    // geometry
    int B, N, F, T, L, heads;
    int s_first, n_steps;          // reverse steps s_first, s_first - 1, ... (n_steps of them)
    int run_post;                  // 1 = posterior + sampling + state update inside the launch (trajectory); 0 = stop after the output layer (taps)
    float guide;
    // packed weights (pack_mfma16 order), [L] matrices back to back
    const bf16_t *wqkv, *wproj, *wfc1, *wfc2, *wout1, *wout2;
    // f32 vectors of block 0 and the stride (floats) to the same vector of the next block
    const float *proj_b, *fc1_b, *fc2_b, *qn_w, *qn_b, *kn_w, *kn_b;
    int64_t blk_stride;
    const float *out1_b, *out2_b, *WxT, *xe_w, *xe_b;
    // hoisted tables
    const float *modtab, *modo;
    // activations (the engine's buffers, same row layout as the launch chain: rows [0, B N) conditional, [B N, 2 B N) unconditional)
    float *x32;
    bf16_t *xa, *qkv, *ao, *h1, *ho;
    float *ybuf;
    int64_t slab_stride;           // floats between split-K slabs of ybuf
    float *outF;
    PostArgs post;                 // posterior / sampling arguments (state, tables, scratch)
    Ctl *ctl;
};

__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }      // HW_REG_XCC_ID[3:0]

// ---------------------------------------------------------------------------------------------------------------- memory access forms
// Data another CU of the team wrote during this launch is read with sc1 loads (served by the XCD's L2, never by this CU's L1).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 ld16_sc1(const void *base, uint32_t byte_off) {      // base: wave-uniform; byte_off < 2 GB
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(base), (int)byte_off, 0, 16);
}
__device__ __forceinline__ float4 ldf4_sc1(const float *base, int64_t idx) {
    return __builtin_bit_cast(float4, ld16_sc1(base, (uint32_t)(idx * 4)));
}
template <typename T> __device__ __forceinline__ T ld_sc1(const T *p) {                  // 1 / 4 / 8 byte scalars, per-lane pointers
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------- LDS layout
// [0, PANEL)            activation panel [ROWS][KCH * 2 bytes], 16-byte piece p of row r at p ^ (r & 15); later the output image of the
//                       MFMA waves, the attention images, the LayerNorm exchange
// [PANEL, PANEL + 4 KB) counters and the bias row of the running GEMM
struct Flags {
    int io_drain, io_go, io_bar, m_done, m_part;      // monotonic counters (LDS)
    unsigned int gen_io, pad0, pad1;
    float bias[128];                                   // bias of the CU's output columns (<= 128)
    float red[2][4][64];                               // LayerNorm row statistics exchange
};
static_assert(sizeof(Flags) <= FLAG_BYTES, "flag block");

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // leaves global loads in flight
// The counters live in LDS and must be accessed with DS instructions: through a generic pointer the compiler emits FLAT operations, which
// count on vmcnt as well -- and a wait for one of them drains the MFMA waves' whole weight FIFO.
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ lds_int_t *as_lds(int *p) { return (lds_int_t *)p; }
__device__ __forceinline__ void lds_signal(int *ctr, int lane) {
    lds_fence();
    if (lane == 0) __hip_atomic_fetch_add(as_lds(ctr), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// bounded: a protocol error must end in an error word, never in a hung GPU (the caller's `alive` turns false and every later wait of the
// wave falls through at once)
__device__ __forceinline__ void lds_wait(int *ctr, int target, bool &alive, unsigned int *err) {
    unsigned int spins = 0;
    while (alive && __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile lds_int_t *>(as_lds(ctr))) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 30000000u) {
            alive = false;
            atomicOr(err, 4u);
        }
    }
    asm volatile("" ::: "memory");
}
// an opaque zero: adding it to a per-lane value keeps the compiler from hoisting that value out of the phase loops (hoisted, dozens of
// loop-invariant addresses compete with the weight FIFO for registers and end up in scratch -- whose reloads drain the FIFO)
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}

// Per-wave view of the team / workgroup
struct Who {
    int xcc, rank;                 // team (XCC) and CU rank inside it
    int wave, lane;                // wave 0..7
    bool io;                       // waves 4..7
    int iw;                        // I/O wave index 0..3 (MFMA wave index for !io)
    Flags *fl;
    unsigned char *lds;
    Ctl *ctl;
    unsigned int bar_gen;          // team barriers passed so far
    int io_gen, m_gen, p_tgt;      // uses of the LDS group barriers (p_tgt: expected value of the partial-tile counter)
    unsigned int *err;             // error word of the launch's control block
    bool ok;
};

// barrier of the four I/O waves (LDS counter)
__device__ __forceinline__ void io_barrier(Who &w) {
    ++w.io_gen;
    lds_signal(&w.fl->io_bar, w.lane);
    lds_wait(&w.fl->io_bar, 4 * w.io_gen, w.ok, w.err);
}

// Team barrier, called by the four I/O waves: every global store of this workgroup that the next phase's readers need was issued by
// one of them.  Each drains its own stores; wave 0 then bumps the team counter and polls it.
__device__ __forceinline__ void team_barrier(Who &w) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ++w.bar_gen;
    lds_signal(&w.fl->io_drain, w.lane);
    if (w.iw == 0) {
        lds_wait(&w.fl->io_drain, 4 * (int)w.bar_gen, w.ok, w.err);
        unsigned int *ctr = &w.ctl->bar[w.xcc][0];
        if (w.lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = w.bar_gen * TEAM;
        unsigned int spins = 0;                                        // the whole wave polls the one word (wave-uniform control flow)
        while (w.ok && (unsigned int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000u) {                                 // seconds: a team member never arrived
                atomicOr(w.err, 1u);
                w.ok = false;
            }
        }
        if (w.lane == 0) __hip_atomic_store(as_lds(&w.fl->io_go), (int)w.bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    lds_wait(&w.fl->io_go, (int)w.bar_gen, w.ok, w.err);
}

// global row of panel row r of graph g: rows [0, NP) conditional, [NP, 2 NP) unconditional; -1 for padding rows
__device__ __forceinline__ int grow_of(int r, int g, int B, int N) {
    const int node = r & (NP - 1);
    if (node >= N) return -1;
    return (r >> 5) * B * N + g * N + node;
}

// ---------------------------------------------------------------------------------------------------------------- panel staging (I/O waves)
// [ROWS][KCH] bf16 of `src` (row stride ld elements, K offset k0) -> LDS, XOR-swizzled 16-byte pieces.  256 threads; every load of the
// panel is in flight before the first LDS write.  Thread t owns column piece t % PPR of the rows t / PPR + i * RPI: the per-lane part of
// the address is ONE VGPR, the row part of step i is wave-uniform (soffset) -- no per-load address registers, no branches (N is a multiple
// of 16, so whether step i is a padding row is wave-uniform too).
template <int KCH>
__device__ __forceinline__ void stage_panel(const Who &w, const bf16_t *src, int ld, int k0, int g, int B, int N) {
    constexpr int PPR = KCH / 8;                       // 16-byte pieces per row
    constexpr int RPI = 256 / PPR;                     // rows covered by one step of the 256 threads (2 at KCH = 1024, 8 at 256)
    constexpr int PT = ROWS / RPI;                     // steps = loads per thread
    static_assert(256 % PPR == 0 && ROWS % RPI == 0 && NP % RPI == 0, "panel pieces");
    const int tid = w.iw * 64 + w.lane + opaque_zero();      // (per-lane values recomputed per phase: see opaque_zero)
    const int r0 = tid / PPR, col = tid % PPR;
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(src);
    const int voff = (r0 * ld + k0 + col * 8) * 2;     // bytes; row r0 of the sequence
    const int seq0 = g * N * ld * 2, seq1 = (B * N + g * N) * ld * 2;      // byte offsets of the two sequences (< 2 GB)
    u32x4 v[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int node0 = (i * RPI) & (NP - 1);        // first node of the step (the step's rows are node0 + r0)
        // padding rows: a lane offset beyond the descriptor's range -- the buffer load returns zeros, no branch
        // (2 GB in the VGPR offset: the range check compares the START of the access, voffset + inst_offset, with num_records = 2 GB - 1;
        // measured: 0x7ffffff0 in either offset register is served, and faults)
        const int soff = ((i * RPI) >> 5 ? seq1 : seq0) + (node0 < N ? node0 : 0) * ld * 2;
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, node0 < N ? voff : (int)0x80000000u, soff, 16);      // sc1
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int row = r0 + i * RPI;
        *reinterpret_cast<u32x4 *>(w.lds + row * (KCH * 2) + ((col ^ (row & 15)) << 4)) = v[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight FIFO (MFMA waves)
// One MFMA wave's share of a block's weights, in consumption order: q|k|v, proj, fc1, fc2.  Fragment (tile t, k-step j) of a phase is the
// 1 KB block at base + t * tile_stride + j * 1024 of the packed copy; `base` carries layer, CU and wave offsets.
template <int H> struct Geom {
    static constexpr int KS = H / 32;                  // k-steps of one K chunk of H
    static constexpr int HM = 4 * H;
    // tiles per CU / tiles per wave / K split over waves / k-steps per wave
    static constexpr int QKV_T = 3 * H / 16 / TEAM, QKV_TW = QKV_T / 2, QKV_KW = 2, QKV_NJ = KS / 2;      // 6 / 3 / 2 / 16 at H = 1024
    static constexpr int PRJ_T = H / 16 / 8, PRJ_TW = PRJ_T / 4, PRJ_NJ = KS / 4;                          // 8 / 2 / 8: a K quarter per CU
    static constexpr int FC_T = HM / 16 / TEAM, FC_TW = FC_T / 4, FC_NJ = KS;                              // 8 / 2 / 32
    static constexpr int NF_QKV = QKV_TW * QKV_NJ, NF_PRJ = PRJ_TW * PRJ_NJ, NF_FC = FC_TW * FC_NJ;        // 48 / 16 / 64
    static constexpr int OFF_QKV = 0, OFF_PRJ = NF_QKV, OFF_FC1 = OFF_PRJ + NF_PRJ, OFF_FC2 = OFF_FC1 + NF_FC;
    static constexpr int NF_BLOCK = OFF_FC2 + NF_FC;                                                       // 192
    static constexpr int D = 32;                                                 // FIFO depth (fragments per wave)
    static_assert(QKV_T % 2 == 0 && PRJ_T % 4 == 0 && FC_T % 4 == 0, "tiles must divide among the MFMA waves");
    static_assert(NF_BLOCK % D == 0, "the FIFO slot of a fragment must not depend on the block");
    static constexpr int PANEL = ROWS * H * 2;         // bytes
};

struct WBase {                     // per MFMA wave: buffer descriptor of each phase's packed weights, byte offset of this wave's fragment
    __amdgpu_buffer_rsrc_t r[4];   // (0, 0) at layer 0, and the layer stride.  Fragments are fetched with raw buffer loads: the per-lane part
    uint32_t off[4];               // of the address is ONE VGPR (lane * 16) shared by every load, everything else lives in SGPRs
    uint32_t ls[4];
};

template <int H, int PH> struct PhaseOf;               // static description of phase PH (0 q|k|v, 1 proj, 2 fc1, 3 fc2)
template <int H> struct PhaseOf<H, 0> { static constexpr int OFF = Geom<H>::OFF_QKV, TW = Geom<H>::QKV_TW, NJ = Geom<H>::QKV_NJ, TS = Geom<H>::KS * 1024; };
template <int H> struct PhaseOf<H, 1> { static constexpr int OFF = Geom<H>::OFF_PRJ, TW = Geom<H>::PRJ_TW, NJ = Geom<H>::PRJ_NJ, TS = Geom<H>::KS * 1024; };
template <int H> struct PhaseOf<H, 2> { static constexpr int OFF = Geom<H>::OFF_FC1, TW = Geom<H>::FC_TW, NJ = Geom<H>::FC_NJ, TS = Geom<H>::KS * 1024; };
template <int H> struct PhaseOf<H, 3> { static constexpr int OFF = Geom<H>::OFF_FC2, TW = Geom<H>::FC_TW, NJ = Geom<H>::FC_NJ, TS = 4 * Geom<H>::KS * 1024; };

// byte offset (inside the phase's weight array) of fragment `F` (index in the block's sequence, 0 <= F < NF_BLOCK) of layer `layer`
template <int H, int F> struct FragOf {
    using G = Geom<H>;
    static constexpr int PH = F < G::OFF_PRJ ? 0 : F < G::OFF_FC1 ? 1 : F < G::OFF_FC2 ? 2 : 3;
    using P = PhaseOf<H, PH>;
    static constexpr int I = F - P::OFF, T = I % P::TW, J = I / P::TW;
    static constexpr uint32_t REL = (uint32_t)T * P::TS + (uint32_t)J * 1024;
};
template <int H, int F>
__device__ __forceinline__ u32x4 load_frag(const WBase &wb, int layer, int voff) {
    using FO = FragOf<H, F>;
    const uint32_t soff = wb.off[FO::PH] + (uint32_t)layer * wb.ls[FO::PH] + FO::REL;
    return __builtin_amdgcn_raw_buffer_load_b128(wb.r[FO::PH], voff, (int)soff, 2);      // aux 2 = nt: streamed once per CU
}

template <int N> struct IC { static constexpr int value = N; };
template <int N, int I = 0, typename Fn> __device__ __forceinline__ void static_for(Fn &&fn) {
    if constexpr (I < N) {
        fn(IC<I>());
        static_for<N, I + 1>(fn);
    }
}

template <int H> struct Fifo { u32x4 s[Geom<H>::D]; };

// weight fragment outside the FIFO sequence (output layer): 1 KB at byte `off` of the array behind `r`
__device__ __forceinline__ u32x4 load_frag_at(__amdgpu_buffer_rsrc_t r, uint32_t off, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, (int)off, 2);
}

// ---------------------------------------------------------------------------------------------------------------- GEMM phase (MFMA waves)
// Output epilogues
enum { EP_BF16 = 0, EP_BIAS_GELU_BF16 = 1, EP_RAW_F32 = 2, EP_BIAS_F32 = 3 };

// GELU(x) = x Phi(x) with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, four orders below the bf16 rounding of the result) on the
// hardware exp / rcp: ~15 instructions against ~40 of the library erff -- the epilogue of fc1 runs on the four MFMA waves only
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);      // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

// acc tile -> output image in LDS.  acc[mt][r]: token row mt * 16 + (lane & 15), column (lane >> 4) * 4 + r of the 16-column tile.
template <int EPI>
__device__ __forceinline__ void put_tile(unsigned char *img, int irow_bytes, int col0, const af32x4 (&acc)[4], const float *bias_lds, int lane) {
    const int lz = lane + opaque_zero();
    const int fi = lz & 15, cq = (lz >> 4) * 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float o[4] = {acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]};
        if (EPI == EP_BIAS_GELU_BF16 || EPI == EP_BIAS_F32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += bias_lds[col0 + cq + r];
        }
        if (EPI == EP_BIAS_GELU_BF16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = gelu_fast(o[r]);
        }
        unsigned char *dst = img + (mt * 16 + fi) * irow_bytes;
        if (EPI == EP_BF16 || EPI == EP_BIAS_GELU_BF16) {
            *reinterpret_cast<uint2 *>(dst + (col0 + cq) * 2) = make_uint2((uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16),
                                                                          (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16));
        } else {
            *reinterpret_cast<float4 *>(dst + (col0 + cq) * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// One GEMM phase of an MFMA wave.  Consumes NF = TW * NJ fragments from FIFO slots (OFF + i) % D and re-issues, into each slot, the
// fragment D places further down the wave's sequence (same block, or the next one: `layer_next`, wrapped by the caller).
//   PH    phase 0..3 (static description above);  KW = K split between waves (1 | 2);  kq = this wave's K part;  grp_col0 = first output
//   column (relative to the CU's columns) of this wave's tiles;  ks0 = first k-step of this wave inside the staged panel chunk
template <int H, int PH, int KW, int EPI>
__device__ __forceinline__ void gemm_phase_mfma(Who &w, Fifo<H> &ff, const WBase &wb, int layer, int layer_next, int kq, int grp, int ks0,
                                                int icols) {
    using G = Geom<H>;
    using P = PhaseOf<H, PH>;
    constexpr int TW = P::TW, NJ = P::NJ, D = G::D;
    constexpr int ESZ = (EPI == EP_BF16 || EPI == EP_BIAS_GELU_BF16) ? 2 : 4;
    const int lane = w.lane;
    const int fi = lane & 15, fq = lane >> 4;
    af32x4 acc[TW][4];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[t][mt] = af32x4{0.f, 0.f, 0.f, 0.f};
    // panel chunk in LDS: row pitch = (k-steps staged) * 64 bytes; q|k|v / fc1 / fc2 stage H columns, proj a quarter of H
    constexpr int KCH = PH == 1 ? H / 4 : H;
    constexpr int ROWB = KCH * 2;
    // fragment of k-step ks, m-tile mt: row mt * 16 + fi, 16-byte piece (ks * 4 + fq) ^ fi.  The XOR touches only the low four bits of the
    // piece index, i.e. fq and ks & 3: four per-lane patterns, everything else is an immediate offset (ks0 is a multiple of 4)
    uint32_t ax[4];
    const int oz = opaque_zero();
#pragma unroll
    for (int m = 0; m < 4; ++m) ax[m] = (uint32_t)(fi * ROWB + ((((m << 2) | fq) ^ fi) << 4) + (ks0 >> 2) * 256 + oz);
    // k-step J: m-tile by m-tile, the TW tiles of an m-tile back to back; as soon as an m-tile's fragment has fed its last MFMA the same
    // registers receive the fragment of k-step J + 1, so the LDS latency hides behind the MFMAs of the other m-tiles (no second register set)
    abf16x8 afr[4];
    {
        const unsigned char *ap = w.lds + ax[0];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) afr[mt] = *reinterpret_cast<const abf16x8 *>(ap + mt * 16 * ROWB);
    }
    static_for<NJ>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const unsigned char *an = w.lds + ax[(J + 1) & 3] + ((J + 1) >> 2) * 256;
        static_for<4>([&](auto mc) {
            constexpr int MT = decltype(mc)::value;
            static_for<TW>([&](auto tc) {
                constexpr int T = decltype(tc)::value;
                constexpr int SLOT = (P::OFF + J * TW + T) % D;
                acc[T][MT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(abf16x8, ff.s[SLOT]), afr[MT], acc[T][MT], 0, 0, 0);
            });
            if constexpr (J + 1 < NJ) afr[MT] = *reinterpret_cast<const abf16x8 *>(an + MT * 16 * ROWB);
            __builtin_amdgcn_sched_barrier(0);
        });
        // the slots of this k-step are free: each takes the fragment D places further down the wave's sequence
        static_for<TW>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            constexpr int F = P::OFF + J * TW + T;
            constexpr int SLOT = F % D;
            constexpr int FN = F + D;
#ifndef LL_TEAM_NOSTREAM     // (probe switch: no weight traffic at all -- results are garbage; what do the phases cost without the stream?)
            if constexpr (FN < G::NF_BLOCK) ff.s[SLOT] = load_frag<H, FN>(wb, layer, lane * 16);
            else ff.s[SLOT] = load_frag<H, FN - G::NF_BLOCK>(wb, layer_next, lane * 16);
#endif
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    // every MFMA wave is done with the panel before its LDS turns into the output image
    ++w.m_gen;
    lds_signal(&w.fl->m_done, lane);
    lds_wait(&w.fl->m_done, 4 * w.m_gen, w.ok, w.err);
    const int irow = icols * ESZ + 16;                               // padded row pitch of the image
    if (KW == 2) {
        // partial tiles of the upper K half go through LDS (behind the image), summed lower + upper
        float *part = reinterpret_cast<float *>(w.lds + 48 * 1024) + (size_t)grp * (TW * 4 * 64 * 4);
        w.p_tgt += 2;                                  // the two upper-half waves publish
        if (kq == 1) {
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<af32x4 *>(part + ((t * 4 + mt) * 64 + lane) * 4) = acc[t][mt];
            lds_signal(&w.fl->m_part, lane);
        } else {
            lds_wait(&w.fl->m_part, w.p_tgt, w.ok, w.err);
#pragma unroll
            for (int t = 0; t < TW; ++t) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const af32x4 o = *reinterpret_cast<const af32x4 *>(part + ((t * 4 + mt) * 64 + lane) * 4);
                    acc[t][mt][0] += o[0]; acc[t][mt][1] += o[1]; acc[t][mt][2] += o[2]; acc[t][mt][3] += o[3];
                }
                put_tile<EPI>(w.lds, irow, (grp * TW + t) * 16, acc[t], w.fl->bias, lane);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < TW; ++t) put_tile<EPI>(w.lds, irow, (grp * TW + t) * 16, acc[t], w.fl->bias, lane);
    }
}

// I/O waves: output image [ROWS][icols] (row pitch icols * ESZ + 16) -> global rows of graph g, columns [col0, col0 + icols) of `dst`
template <int ESZ>
__device__ __forceinline__ void copy_out(const Who &w, void *dst, int ld, int col0, int icols, int g, int B, int N) {
    const int ppr = icols * ESZ / 16;                                  // 16-byte pieces per row
    const int irow = icols * ESZ + 16;
    const int tid = w.iw * 64 + w.lane + opaque_zero();
    for (int pc = tid; pc < ROWS * ppr; pc += 256) {
        const int row = pc / ppr, col = pc - row * ppr;
        const int gr = grow_of(row, g, B, N);
        if (gr >= 0)
            *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(dst) + ((int64_t)gr * ld + col0) * ESZ + col * 16) =
                *reinterpret_cast<const u32x4 *>(w.lds + row * irow + col * 16);
    }
}

// bias row of the CU's columns -> LDS (I/O wave 0), read by the MFMA waves' epilogue
__device__ __forceinline__ void stage_bias(const Who &w, const float *bias, int col0, int ncols) {
    if (w.iw == 0)
        for (int i = w.lane; i < ncols; i += 64) w.fl->bias[i] = bias[col0 + i];
}

// ---------------------------------------------------------------------------------------------------------------- attention phase (I/O waves)
template <int H>
__device__ __forceinline__ void attn_phase(Who &w, const Args &a, int layer, int g) {
    const int heads = a.heads;
    const int unit = w.rank;                                          // (sequence half, head)
    if (unit >= 2 * heads) return;
    const int half = unit / heads, head = unit - half * heads;
    const int row0 = half * a.B * a.N + g * a.N;                      // first global row of the sequence
    const int nv = a.post.n_nodes[g];
    const float *qw = a.qn_w + layer * a.blk_stride, *qb = a.qn_b + layer * a.blk_stride;
    const float *kw = a.kn_w + layer * a.blk_stride, *kb = a.kn_b + layer * a.blk_stride;
    const bf16_t *qkv = a.qkv;
    const int64_t ld3 = 3 * (int64_t)H;
    struct IoSync {
        Who *w;
        __device__ __forceinline__ void operator()() const { io_barrier(*w); }
    };
    attn_mfma_body<NP, 64, 4>(
        [&](int row, int which, int d0) {
            const u32x4 v = ld16_sc1(qkv, (uint32_t)((((int64_t)(row0 + row)) * ld3 + (int64_t)which * H + head * 64 + d0) * 2));
            return make_uint4(v[0], v[1], v[2], v[3]);
        },
        a.ao + (int64_t)row0 * H + head * 64, qw, qb, kw, kb, a.N, nv, H, 64, w.lds, w.iw, w.lane, IoSync{&w});
}

// ---------------------------------------------------------------------------------------------------------------- AdaLN epilogue (I/O waves)
// x += gate * (LN0(sum of slabs + bias) * (1 + scale) + shift) for the CU's two panel rows; the arithmetic of ln_mod_res_mw_kernel
// (bf16 engine: sum and sum of squares in one exchange), one wave per 256-column chunk of a row.
template <int H>
__device__ __forceinline__ void ln_phase(Who &w, const Args &a, const float *bias, int nslab, int layer, int sel, int s, int g) {
    // both rows of the CU at once: I/O waves 0, 1 take panel row 2 rank, waves 2, 3 row 2 rank + 1; a wave owns half a row (NC float4 chunks per
    // lane, column chunk (half * NC + k) * 256 + lane * 4), so the whole phase is ONE round of loads and ONE exchange of (sum, sum of squares)
    constexpr int NC = H / 512;                        // float4 chunks per lane
    static_assert(H % 512 == 0, "two waves per row");
    const int lane = w.lane + opaque_zero();
    const int rsel = w.iw >> 1, half = w.iw & 1;
    const int prow = 2 * w.rank + rsel;
    const int gr = grow_of(prow, g, a.B, a.N);
    const int ci = (prow >> 5) == 0 ? g : a.B;
    const float *mod = a.modtab + (((int64_t)s * (a.B + 1) + ci) * a.L + layer) * (6 * (int64_t)H) + (int64_t)sel * 3 * H;
    float4 v[NC], xr[NC], sh[NC], sc[NC], ga[NC];
    float ps = 0.f, pq = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int h = ((half * NC + k) * 64 + lane) * 4;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        xr[k] = sh[k] = sc[k] = ga[k] = v[k];
        if (gr >= 0) {
            float4 t[4];
#pragma unroll
            for (int z = 0; z < 4; ++z)
                if (z < nslab) t[z] = ldf4_sc1(a.ybuf, z * a.slab_stride + (int64_t)gr * H + h);
            const float4 bb = *reinterpret_cast<const float4 *>(bias + h);
            xr[k] = ldf4_sc1(a.x32, (int64_t)gr * H + h);
            sh[k] = *reinterpret_cast<const float4 *>(mod + h);
            sc[k] = *reinterpret_cast<const float4 *>(mod + H + h);
            ga[k] = *reinterpret_cast<const float4 *>(mod + 2 * H + h);
#pragma unroll
            for (int z = 0; z < 4; ++z)
                if (z < nslab) { v[k].x += t[z].x; v[k].y += t[z].y; v[k].z += t[z].z; v[k].w += t[z].w; }
            v[k].x += bb.x; v[k].y += bb.y; v[k].z += bb.z; v[k].w += bb.w;
        }
        ps += v[k].x + v[k].y + v[k].z + v[k].w;
        pq += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
    }
    // bf16 engine's form of ln_mod_res_mw_kernel: var = E[y^2] - mean^2 in f32, clamped at 0
    ps = wave_sum(ps);
    pq = wave_sum(pq);
    if (lane == 0) {
        w.fl->red[0][w.iw][0] = ps;
        w.fl->red[1][w.iw][0] = pq;
    }
    io_barrier(w);
    const float sum = w.fl->red[0][rsel * 2][0] + w.fl->red[0][rsel * 2 + 1][0];
    const float sq = w.fl->red[1][rsel * 2][0] + w.fl->red[1][rsel * 2 + 1][0];
    const float mean = sum / (float)H;
    const float rstd = rsqrtf(fmaxf(sq / (float)H - mean * mean, 0.f) + 1e-5f);
    if (gr >= 0) {
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int h = ((half * NC + k) * 64 + lane) * 4;
            float4 o;
            o.x = xr[k].x + ga[k].x * ((v[k].x - mean) * rstd * (1.f + sc[k].x) + sh[k].x);
            o.y = xr[k].y + ga[k].y * ((v[k].y - mean) * rstd * (1.f + sc[k].y) + sh[k].y);
            o.z = xr[k].z + ga[k].z * ((v[k].z - mean) * rstd * (1.f + sc[k].z) + sh[k].z);
            o.w = xr[k].w + ga[k].w * ((v[k].w - mean) * rstd * (1.f + sc[k].w) + sh[k].w);
            *reinterpret_cast<float4 *>(a.x32 + (int64_t)gr * H + h) = o;
            store4<bf16_t>(a.xa + (int64_t)gr * H + h, o);
        }
    }
    io_barrier(w);                                     // the exchange words are free again
}

// ---------------------------------------------------------------------------------------------------------------- x_embedder (I/O waves)
// embed_kernel's arithmetic for node `rank` of graph g (transformer.py:41-44, 95-96): gather-sum of <= N + 1 rows of W_x^T, affine LN,
// written to the conditional and the unconditional row.  256 threads.
template <int H>
__device__ __forceinline__ void embed_phase(Who &w, const Args &a, int s, int g) {
    const int i = w.rank, N = a.N, B = a.B;
    const int tid = w.iw * 64 + w.lane + opaque_zero();
    int *gidx = reinterpret_cast<int *>(w.lds);        // [72] + count at [72]
    float *red = reinterpret_cast<float *>(w.lds + 512);
    const bool live = i < N;
    if (live && w.iw == 0) {
        const int lane = w.lane;
        const int half = (s + 1) & 1;
        const int8_t *er = a.post.E + (((int64_t)half * B + g) * N + i) * N;
        const int xi = ld_sc1(a.post.X + (int64_t)half * B * N + g * N + i);
        const int e = (lane < N) ? (int)ld_sc1(er + lane) : -1;
        const unsigned long long m = __ballot(e >= 0);
        const int base = (xi >= 0) ? 1 : 0;
        if (lane == 0 && xi >= 0) gidx[0] = xi;
        if (e >= 0) gidx[base + __popcll(m & ((1ull << lane) - 1ull))] = XD + ED * lane + e;
        if (lane == 0) gidx[72] = base + __popcll(m);
    }
    io_barrier(w);
    constexpr int MAXE = (H + 1023) / 1024;            // float4 chunks per thread
    float4 v[MAXE];
    const int n = live ? gidx[72] : 0;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int h = (tid + e * 256) * 4;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h < H) {
            for (int g0 = 0; g0 < n; g0 += 8) {
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int gg = g0 + u;
                    t[u] = (gg < n) ? *reinterpret_cast<const float4 *>(a.WxT + (int64_t)gidx[gg] * H + h) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { sum.x += t[u].x; sum.y += t[u].y; sum.z += t[u].z; sum.w += t[u].w; }
            }
        }
        v[e] = sum;
    }
    auto block_sum = [&](float x) {                    // block_sum_256's order: wave sums, then wave 0 + 1 + 2 + 3
        x = wave_sum(x);
        io_barrier(w);
        if (w.lane == 0) red[w.iw] = x;
        io_barrier(w);
        return red[0] + red[1] + red[2] + red[3];
    };
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) ls += v[e].x + v[e].y + v[e].z + v[e].w;
    const float mean = block_sum(ls) / (float)H;
    float lv = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        if ((tid + e * 256) * 4 < H) {
            const float d0 = v[e].x - mean, d1 = v[e].y - mean, d2 = v[e].z - mean, d3 = v[e].w - mean;
            lv += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
    }
    const float rstd = rsqrtf(block_sum(lv) / (float)H + 1e-5f);
    if (live) {
        const int64_t M = (int64_t)B * N, row = (int64_t)g * N + i;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const int h = (tid + e * 256) * 4;
            if (h < H) {
                const float4 ww = *reinterpret_cast<const float4 *>(a.xe_w + h);
                const float4 bb = *reinterpret_cast<const float4 *>(a.xe_b + h);
                float4 o;
                o.x = (v[e].x - mean) * rstd * ww.x + bb.x;
                o.y = (v[e].y - mean) * rstd * ww.y + bb.y;
                o.z = (v[e].z - mean) * rstd * ww.z + bb.z;
                o.w = (v[e].w - mean) * rstd * ww.w + bb.w;
                *reinterpret_cast<float4 *>(a.x32 + row * H + h) = o;
                *reinterpret_cast<float4 *>(a.x32 + (M + row) * H + h) = o;
                store4<bf16_t>(a.xa + row * H + h, o);
                store4<bf16_t>(a.xa + (M + row) * H + h, o);
            }
        }
    }
    io_barrier(w);                                     // gidx / red are panel LDS: nobody stages a panel before everybody has read them
}

// ---------------------------------------------------------------------------------------------------------------- the two roles
// Both roles walk the same sequence of phases; the workgroup barriers (A) "panel staged" and (B) "output image written" of every GEMM
// phase are the only points where they meet, so each role executes exactly two wg_barrier() per GEMM phase.
constexpr int NTEAMS = 8;

template <int H>
__device__ __forceinline__ void io_main(Who &w, const Args &a) {
    using G = Geom<H>;
    const int c = w.rank, B = a.B, N = a.N, lane = w.lane;
    const int ftiles = a.F / 16;
    constexpr int T1 = H / 16 / TEAM;                  // output-layer tiles per CU of Linear(H, H)
    for (int step = 0; step < a.n_steps; ++step) {
        const int s = a.s_first - step;
        for (int g = w.xcc; g < B; g += NTEAMS) {
            // ================================================================ embedding of z_{s+1}
            embed_phase<H>(w, a, s, g);
            team_barrier(w);
            for (int l = 0; l < a.L; ++l) {
                // ------------------------------------------------------------ q|k|v
                LL_TSTAMP(0);
                stage_panel<H>(w, a.xa, H, 0, g, B, N);
                LL_TSTAMP(1);
                wg_barrier();                                          // (A)
                wg_barrier();                                          // (B)
                LL_TSTAMP(2);
                copy_out<2>(w, a.qkv, 3 * H, c * G::QKV_T * 16, G::QKV_T * 16, g, B, N);
                LL_TSTAMP(3);
                team_barrier(w);
                LL_TSTAMP(4);
                attn_phase<H>(w, a, l, g);
                LL_TSTAMP(5);
                team_barrier(w);
                LL_TSTAMP(6);
                // ------------------------------------------------------------ proj (K quarter per CU -> raw slabs)
                stage_panel<H / 4>(w, a.ao, H, (c >> 3) * (H / 4), g, B, N);
                LL_TSTAMP(7);
                wg_barrier();
                wg_barrier();
                LL_TSTAMP(8);
                copy_out<4>(w, a.ybuf + (int64_t)(c >> 3) * a.slab_stride, H, (c & 7) * G::PRJ_T * 16, G::PRJ_T * 16, g, B, N);
                LL_TSTAMP(9);
                team_barrier(w);
                LL_TSTAMP(10);
                ln_phase<H>(w, a, a.proj_b + l * a.blk_stride, 4, l, 0, s, g);
                LL_TSTAMP(11);
                team_barrier(w);
                LL_TSTAMP(12);
                // ------------------------------------------------------------ fc1
                stage_bias(w, a.fc1_b + l * a.blk_stride, c * G::FC_T * 16, G::FC_T * 16);
                stage_panel<H>(w, a.xa, H, 0, g, B, N);
                LL_TSTAMP(13);
                wg_barrier();
                wg_barrier();
                LL_TSTAMP(14);
                copy_out<2>(w, a.h1, G::HM, c * G::FC_T * 16, G::FC_T * 16, g, B, N);
                LL_TSTAMP(15);
                team_barrier(w);
                LL_TSTAMP(16);
                // ------------------------------------------------------------ fc2 (K chunk per CU -> raw slabs)
                stage_panel<H>(w, a.h1, G::HM, (c >> 3) * H, g, B, N);
                LL_TSTAMP(17);
                wg_barrier();
                wg_barrier();
                LL_TSTAMP(18);
                copy_out<4>(w, a.ybuf + (int64_t)(c >> 3) * a.slab_stride, H, (c & 7) * G::FC_T * 16, G::FC_T * 16, g, B, N);
                LL_TSTAMP(19);
                team_barrier(w);
                LL_TSTAMP(20);
                ln_phase<H>(w, a, a.fc2_b + l * a.blk_stride, 4, l, 1, s, g);
                LL_TSTAMP(21);
                team_barrier(w);
                LL_TSTAMP(22);
            }
            // ================================================================ output layer: Linear(H, H) + GELU, Linear(H, F)
            stage_bias(w, a.out1_b, c * T1 * 16, T1 * 16);
            stage_panel<H>(w, a.xa, H, 0, g, B, N);
            wg_barrier();
            wg_barrier();
            copy_out<2>(w, a.ho, H, c * T1 * 16, T1 * 16, g, B, N);
            team_barrier(w);
            if (c < ftiles) stage_bias(w, a.out2_b, c * 16, 16);
            stage_panel<H>(w, a.ho, H, 0, g, B, N);
            wg_barrier();
            wg_barrier();
            if (c < ftiles) copy_out<4>(w, a.outF, a.F, c * 16, 16, g, B, N);
            team_barrier(w);
            if (!a.run_post) continue;
            // ================================================================ posterior, guidance, sampling, state update
            if (w.iw < 2) post_rows_body<true>(a.post, grow_of(2 * c + w.iw, g, B, N), lane, s);
            team_barrier(w);
            if (w.iw == 0 && c < N) post_pairs_body<true>(a.post, c, g, lane, s);
            team_barrier(w);
        }
    }
}

// output-layer GEMM of one MFMA wave, weights loaded at the phase (outside the FIFO sequence): tile `tile` (< 0: no tile for this CU),
// k-steps [ks0, ks0 + nj) of the staged panel, K split `kparts` ways between consecutive waves (this one is part `kpart`)
template <int H, int EPI>
__device__ __forceinline__ void direct_gemm(Who &w, const bf16_t *wp, int tile, int ks0, int nj, int kparts, int kpart, int icols, int col_tile) {
    using G = Geom<H>;
    const int lane = w.lane, mw = w.iw;
    const __amdgpu_buffer_rsrc_t wr = rsrc_of(wp);
    const uint32_t p = (uint32_t)(((tile < 0 ? 0 : tile) * G::KS + ks0) * 1024);
    af32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = af32x4{0.f, 0.f, 0.f, 0.f};
    const int fi = lane & 15, fq = lane >> 4;
    constexpr int ROWB = H * 2;
    uint32_t ax[4];
    const int oz = opaque_zero();
#pragma unroll
    for (int m = 0; m < 4; ++m) ax[m] = (uint32_t)(fi * ROWB + ((((m << 2) | fq) ^ fi) << 4) + oz);
    for (int j0 = 0; j0 < nj; j0 += 8) {                               // ks0 and j0 are multiples of 4
        u32x4 wf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wf[u] = load_frag_at(wr, p + (uint32_t)(j0 + u) * 1024, lane * 16);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned char *ap = w.lds + ax[u & 3] + ((ks0 + j0 + u) >> 2) * 256;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const abf16x8 av = *reinterpret_cast<const abf16x8 *>(ap + mt * 16 * ROWB);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(abf16x8, wf[u]), av, acc[mt], 0, 0, 0);
            }
        }
    }
    ++w.m_gen;
    lds_signal(&w.fl->m_done, lane);
    lds_wait(&w.fl->m_done, 4 * w.m_gen, w.ok, w.err);
    // K parts through LDS (behind the image), summed in order by part 0
    float *part = reinterpret_cast<float *>(w.lds + 48 * 1024) + (size_t)mw * (4 * 64 * 4);
    if (kpart != 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<af32x4 *>(part + (mt * 64 + lane) * 4) = acc[mt];
    }
    w.p_tgt += 4;
    lds_signal(&w.fl->m_part, lane);
    lds_wait(&w.fl->m_part, w.p_tgt, w.ok, w.err);
    if (kpart == 0 && tile >= 0) {
        for (int k = 1; k < kparts; ++k) {
            const float *op = part + (size_t)k * (4 * 64 * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const af32x4 o = *reinterpret_cast<const af32x4 *>(op + (mt * 64 + lane) * 4);
                acc[mt][0] += o[0]; acc[mt][1] += o[1]; acc[mt][2] += o[2]; acc[mt][3] += o[3];
            }
        }
        put_tile<EPI>(w.lds, icols * (EPI == EP_BIAS_GELU_BF16 ? 2 : 4) + 16, col_tile * 16, acc, w.fl->bias, lane);
    }
}

template <int H>
__device__ __forceinline__ void mfma_main(Who &w, const Args &a) {
    using G = Geom<H>;
    const int c = w.rank, B = a.B, lane = w.lane, mw = w.iw;
    const int ftiles = a.F / 16;
    // ---- weight stream bases and the first D fragments
    WBase wb;
    Fifo<H> ff;
    {
        const int cg = c & 7, kc = c >> 3;
        const uint32_t tile_b = (uint32_t)G::KS * 1024;               // bytes per 16-row tile of a [*, H] weight
        const int grp = mw >> 1, kq = mw & 1;
        // q|k|v: tiles c * QKV_T + grp * QKV_TW .., k-steps kq * QKV_NJ ..
        wb.r[0] = rsrc_of(a.wqkv);
        wb.off[0] = (uint32_t)(c * G::QKV_T + grp * G::QKV_TW) * tile_b + (uint32_t)kq * G::QKV_NJ * 1024;
        wb.ls[0] = (uint32_t)3 * H * H * 2;
        // proj: column group cg (PRJ_T tiles), K quarter kc; wave mw takes PRJ_TW tiles
        wb.r[1] = rsrc_of(a.wproj);
        wb.off[1] = (uint32_t)(cg * G::PRJ_T + mw * G::PRJ_TW) * tile_b + (uint32_t)kc * G::PRJ_NJ * 1024;
        wb.ls[1] = (uint32_t)H * H * 2;
        // fc1: tiles c * FC_T + mw * FC_TW
        wb.r[2] = rsrc_of(a.wfc1);
        wb.off[2] = (uint32_t)(c * G::FC_T + mw * G::FC_TW) * tile_b;
        wb.ls[2] = (uint32_t)G::HM * H * 2;
        // fc2: column group cg (FC_T tiles), K chunk kc of HM; tile pitch 4 * KS k-steps
        wb.r[3] = rsrc_of(a.wfc2);
        wb.off[3] = (uint32_t)(cg * G::FC_T + mw * G::FC_TW) * (4 * tile_b) + (uint32_t)kc * G::KS * 1024;
        wb.ls[3] = (uint32_t)G::HM * H * 2;
        static_for<G::D>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            ff.s[F] = load_frag<H, F>(wb, 0, lane * 16);
        });
    }
    constexpr int T1 = H / 16 / TEAM, KW1 = 4 / T1, NJ1 = G::KS / KW1;
    for (int step = 0; step < a.n_steps; ++step) {
        for (int g = w.xcc; g < B; g += NTEAMS) {
            for (int l = 0; l < a.L; ++l) {
                const int ln = (l + 1 < a.L) ? l + 1 : 0;             // the stream runs on into the next pass over the blocks
                wg_barrier();                                          // (A) panel staged
                gemm_phase_mfma<H, 0, 2, EP_BF16>(w, ff, wb, l, ln, mw & 1, mw >> 1, (mw & 1) * G::QKV_NJ, G::QKV_T * 16);
                wg_barrier();                                          // (B) output image written
                wg_barrier();
                gemm_phase_mfma<H, 1, 1, EP_RAW_F32>(w, ff, wb, l, ln, 0, mw, 0, G::PRJ_T * 16);
                wg_barrier();
                wg_barrier();
                gemm_phase_mfma<H, 2, 1, EP_BIAS_GELU_BF16>(w, ff, wb, l, ln, 0, mw, 0, G::FC_T * 16);
                wg_barrier();
                wg_barrier();
                gemm_phase_mfma<H, 3, 1, EP_RAW_F32>(w, ff, wb, l, ln, 0, mw, 0, G::FC_T * 16);
                wg_barrier();
            }
            // output layer: Linear(H, H) with T1 tiles per CU, wave mw -> tile mw / KW1, K part mw % KW1; Linear(H, F): one tile on the first
            // F / 16 CUs, K split over the four waves
            wg_barrier();
            direct_gemm<H, EP_BIAS_GELU_BF16>(w, a.wout1, c * T1 + mw / KW1, (mw % KW1) * NJ1, NJ1, KW1, mw % KW1, T1 * 16, mw / KW1);
            wg_barrier();
            wg_barrier();
            direct_gemm<H, EP_BIAS_F32>(w, a.wout2, c < ftiles ? c : -1, mw * (G::KS / 4), G::KS / 4, 4, mw, 16, 0);
            wg_barrier();
        }
    }
    // nothing of the stream may still be in flight into registers that die with the wave
    u32x4 x = (u32x4)(0);
    static_for<G::D>([&](auto fc) { x ^= ff.s[decltype(fc)::value]; });
    if (x[0] == 0x7fc01234u && x[1] == 0x12345678u && x[2] == 0x9abcdef0u) a.ctl->teams_done = 1;      // never true in practice: keeps the loads alive
}

// ---------------------------------------------------------------------------------------------------------------- the kernel
template <int H>
__global__ __launch_bounds__(THREADS) void dit_team_kernel(Args a) {
    using G = Geom<H>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_team[];
    Who w;
    w.lds = lds_team;
    w.fl = reinterpret_cast<Flags *>(lds_team + G::PANEL);
    w.ctl = a.ctl;
    w.err = &a.ctl->error;
    w.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.lane = threadIdx.x & 63;
    w.io = w.wave >= 4;
    w.iw = w.wave & 3;
    w.bar_gen = 0;
    w.io_gen = w.m_gen = w.p_tgt = 0;
    w.ok = true;
    w.xcc = (int)xcc_id();
    // ---- census: the workgroups that find themselves on XCC x form team x; rank = order of arrival
    __shared__ int s_rank;
    if (threadIdx.x == 0) {
        s_rank = (int)atomicAdd(&a.ctl->census[w.xcc][0], 1u);
        Flags *f = w.fl;
        f->io_drain = f->io_go = f->io_bar = f->m_done = f->m_part = 0;
    }
    __syncthreads();
    w.rank = s_rank;
    if (w.rank >= TEAM) {                              // cannot happen with one workgroup per CU; never deadlock the others
        if (threadIdx.x == 0) atomicOr(&a.ctl->error, 2u);
        return;
    }
    // graphs of this team: g = xcc, xcc + 8, ...
    if (w.xcc >= a.B) return;                          // nothing to do for this team (its members never enter a barrier)
    if (threadIdx.x == 0) {                            // wait for the whole team (bounded)
        unsigned int spins = 0;
        while (__hip_atomic_load(&a.ctl->census[w.xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < TEAM) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > 4000000u) { atomicOr(&a.ctl->error, 1u); break; }
        }
    }
    __syncthreads();
    if (w.io) io_main<H>(w, a);
    else mfma_main<H>(w, a);
}


// ================================================================================================ attention projection + AdaLN epilogue, one launch
// The launch chain's proj GEMM (split-K slabs) and its `ln_mod_res` launch are two dependent launches per block; LayerNorm needs whole rows, i.e.
// every column tile of the GEMM, which is what forces the boundary.  proj's weight is small (2 MB of bf16 per block), so it can be REPLICATED
// across the XCDs at no cost that matters (16 MB of L2 misses per launch at batch 8) -- and then one graph's 64 token rows x all 1024 columns can
// be produced by the 32 CUs of ONE XCD, whose hand-off goes through their own L2: GEMM tile -> plain stores -> 1.1 us team barrier -> LayerNorm of two
// rows per CU from sc1 loads.  Graphs of the batch go to the teams round robin.  Unlike dit_team_kernel this launch is short, runs inside the chain
// (after the attention launch, before fc1) and resets its own counters, so the chain needs no memset between launches.
struct Ctl2 {
    unsigned int census[8][32];
    unsigned int bar[8][32];
    unsigned int leave[8][32];       // workgroups of the XCC that are done: the last one zeroes the three counters of its XCC
    unsigned int error;
    unsigned int pad_;
    unsigned long long stamps[16];   // LL_TEAM_PROBE: 100 MHz clock of XCC 0's first workgroup at the phase boundaries of block 3
    unsigned long long span[2][256]; // LL_TEAM_PROBE: every workgroup's first and last clock of that launch
};
#ifdef LL_TEAM_PROBE
#define LL_PSTAMP(i) do { if (w.xcc == 0 && c == 0 && threadIdx.x == 0 && p.layer == 3) ctl->stamps[i] = wall_clock64(); } while (0)
#else
#define LL_PSTAMP(i) do { } while (0)
#endif

struct ProjLnArgs {
    int B, N, L, layer;
    const bf16_t *ao;                // [2 B N][H] attention output
    const bf16_t *wproj;             // packed weight of this layer
    const float *bias;
    const float *modrows;            // this step's modulation rows [(B + 1)][L][6 H]
    float *y;                        // [2 B N][H] f32 scratch (slab 0 of the chain's split-K buffer)
    float *x32;
    bf16_t *xa;
    Ctl2 *ctl;
};

// 512-thread panel staging from data a PREVIOUS launch wrote (plain policy)
template <int KCH>
__device__ __forceinline__ void stage_panel512(unsigned char *lds, int tid, const bf16_t *src, int ld, int g, int B, int N) {
    constexpr int PPR = KCH / 8, RPI = 512 / PPR, PT = ROWS / RPI;
    static_assert(512 % PPR == 0 && ROWS % RPI == 0 && NP % RPI == 0, "panel pieces");
    const int r0 = tid / PPR, col = tid % PPR;
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(src);
    const int voff = (r0 * ld + col * 8) * 2;
    const int seq0 = g * N * ld * 2, seq1 = (B * N + g * N) * ld * 2;
    u32x4 v[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int node0 = (i * RPI) & (NP - 1);
        const int soff = ((i * RPI) >> 5 ? seq1 : seq0) + (node0 < N ? node0 : 0) * ld * 2;
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, node0 < N ? voff : (int)0x80000000u, soff, 0);
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int row = r0 + i * RPI;
        *reinterpret_cast<u32x4 *>(lds + row * (KCH * 2) + ((col ^ (row & 15)) << 4)) = v[i];
    }
}

template <int H>
__global__ __launch_bounds__(THREADS) void proj_ln_team_kernel(ProjLnArgs p) {
    using G = Geom<H>;
    static_assert(H / 16 / TEAM == 2, "two 16-column tiles per CU");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_pl[];
    Who w;
    w.lds = lds_pl;
    w.fl = reinterpret_cast<Flags *>(lds_pl + G::PANEL);
    w.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.lane = threadIdx.x & 63;
    w.io = true;
    w.iw = w.wave;                                     // ln_phase / io_barrier see waves 0..3
    w.bar_gen = 0;
    w.io_gen = w.m_gen = w.p_tgt = 0;
    w.ok = true;
    w.xcc = (int)xcc_id();
    Ctl2 *ctl = p.ctl;
    w.ctl = nullptr;
    w.err = &ctl->error;
    __shared__ int s_rank2;
    if (threadIdx.x == 0) {
        s_rank2 = (int)atomicAdd(&ctl->census[w.xcc][0], 1u);
        Flags *f = w.fl;
        f->io_drain = f->io_go = f->io_bar = f->m_done = f->m_part = 0;
    }
    __syncthreads();
    w.rank = s_rank2;
    const int c = w.rank, B = p.B, N = p.N, lane = w.lane, wave = w.wave;
    unsigned int *err = &ctl->error;
    LL_PSTAMP(0);
#ifdef LL_TEAM_PROBE
    if (threadIdx.x == 0 && p.layer == 3) ctl->span[0][blockIdx.x] = wall_clock64();
#endif
    if (c < TEAM && w.xcc < B) {
        if (threadIdx.x == 0) {                        // the whole team is here (bounded)
            unsigned int spins = 0;
            while (__hip_atomic_load(&ctl->census[w.xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < TEAM) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 4000000u) { atomicOr(err, 1u); break; }
            }
        }
        __syncthreads();
        LL_PSTAMP(1);
        Args a;                                        // what ln_phase reads
        a.B = B; a.N = N; a.L = p.L;
        a.modtab = p.modrows; a.ybuf = p.y; a.slab_stride = 0; a.x32 = p.x32; a.xa = p.xa;
        const __amdgpu_buffer_rsrc_t wr = rsrc_of(p.wproj);
        const int tile = 2 * c + (wave & 1), kq = wave >> 1;           // this wave: one 16-column tile, one K quarter (8 k-steps)
        const int fi = lane & 15, fq = lane >> 4;
        constexpr int ROWB = H * 2;
        unsigned int bar_gen = 0;
        for (int g = w.xcc; g < B; g += NTEAMS) {
            u32x4 wf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[j] = load_frag_at(wr, (uint32_t)((tile * G::KS + kq * 8 + j) * 1024), lane * 16);
            stage_panel512<H>(w.lds, threadIdx.x, p.ao, H, g, B, N);
            __syncthreads();
            LL_PSTAMP(2);
            af32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = af32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ks = kq * 8 + j;
                const unsigned char *ap = w.lds + fi * ROWB + (((((ks & 3) << 2) | fq) ^ fi) << 4) + (ks >> 2) * 256;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(abf16x8, wf[j]), *reinterpret_cast<const abf16x8 *>(ap + mt * 16 * ROWB), acc[mt], 0, 0, 0);
            }
            __syncthreads();                           // the panel is consumed: its LDS takes the K-quarter partials
            float *part = reinterpret_cast<float *>(w.lds);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<af32x4 *>(part + (((wave * 4) + mt) * 64 + lane) * 4) = acc[mt];
            __syncthreads();
            {
                // wave v finishes block (tile v & 1, m-tile v >> 1): quarters summed in order 0..3; lane: row mt * 16 + fi, columns fq * 4 ..
                const int t = wave & 1, mt = wave >> 1;
                af32x4 o = *reinterpret_cast<const af32x4 *>(part + ((((0 * 2 + t) * 4) + mt) * 64 + lane) * 4);
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const af32x4 u = *reinterpret_cast<const af32x4 *>(part + ((((q * 2 + t) * 4) + mt) * 64 + lane) * 4);
                    o[0] += u[0]; o[1] += u[1]; o[2] += u[2]; o[3] += u[3];
                }
                const int gr = grow_of(mt * 16 + fi, g, B, N);
                if (gr >= 0) *reinterpret_cast<float4 *>(p.y + (int64_t)gr * H + (2 * c + t) * 16 + fq * 4) = make_float4(o[0], o[1], o[2], o[3]);
            }
            // ---- team barrier over all eight waves
            LL_PSTAMP(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            LL_PSTAMP(4);
            ++bar_gen;
            if (wave == 0) {
                unsigned int *ctr = &ctl->bar[w.xcc][0];
                if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int target = bar_gen * TEAM;
                unsigned int spins = 0;
                bool alive = true;
                while (alive && (unsigned int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 20000000u) { atomicOr(err, 1u); alive = false; }
                }
            }
            __syncthreads();
            LL_PSTAMP(5);
            // ---- AdaLN epilogue of rows 2 c, 2 c + 1 on waves 0..3
            if (wave < 4) ln_phase<H>(w, a, p.bias, 1, p.layer, 0, 0, g);
            __syncthreads();
            LL_PSTAMP(6);
        }
    } else if (c >= TEAM && threadIdx.x == 0) {
        atomicOr(err, 2u);
    }
    // ---- the last workgroup of the XCC to leave zeroes the XCC's counters: the next launch starts from a clean block without a memset
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int n = __hip_atomic_fetch_add(&ctl->leave[w.xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == TEAM - 1) {
            __hip_atomic_store(&ctl->census[w.xcc][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->bar[w.xcc][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->leave[w.xcc][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef LL_TEAM_PROBE
        if (p.layer == 3) ctl->span[1][blockIdx.x] = wall_clock64();
#endif
    }
}

}  // namespace team
}  // namespace ll
