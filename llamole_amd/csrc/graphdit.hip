// GraphDiT reverse-diffusion engine for MI355X (gfx950): host side of the C ABI in include/llamole_hip.h.
//
// Replaces reference GraphDiT.generate / sample_p_zs_given_zt / Transformer.forward
// (src/model/graph_decoder/diffusion_model.py:252-399, transformer.py:93-187).
//
// MI355X-first structure (not a translation of the reference's per-op ATen stream):
//   * integer graph state in HBM (int8 X[B,N], E[B,N,N]); one-hot floats and the three dense
//     [B,F,F] transition matrices of the reference are never built -- the posterior is evaluated in
//     its structured O(N*F) form inside one fused kernel together with CFG and sampling;
//   * conditional and unconditional passes run as ONE batch of 2*B*N tokens, so each weight matrix is
//     streamed from HBM once per step;
//   * everything that does not depend on the graph state is hoisted out of the T-step loop at
//     ll_dit_begin: c = c_t + c_y + c_txt for every step and all L+1 adaLN modulations
//     (7 of the 19 H^2 weights per block are then never read inside the loop);
//   * a step is ~7 launches per block; the step index lives in device memory so ONE captured
//     hipGraph is replayed T times (no host sync inside the trajectory; the reference syncs 3x/step).
#include <stdarg.h>
#include <stdlib.h>

#include <unordered_map>
#include <vector>

#include "dit_kernels.h"

namespace ll {

static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// ------------------------------------------------------------------------------------------ widths
// The reference classes take any sizes (transformer.py:24-37: hidden_size, num_heads, mlp_ratio come from the checkpoint's config.yaml).
// The kernels want row pitches that are multiples of 64 and a head pitch that is a multiple of the MFMA k-step, so the engine keeps an
// INTERNAL copy of the weights zero-padded to such pitches whenever the checkpoint's widths are not already of that form:
//   Hp  = hidden rounded up to 64                    (residual stream, every H-wide vector and weight dimension)
//   Hmp = mlp_hidden rounded up to 64
//   hdp = head_dim rounded up to 32, Ha = heads * hdp rounded up to 64   (q | k | v sections of a qkv row, attention output, K of proj)
// Zero weight rows / columns, zero biases and zero LayerNorm affine parameters make every padded column of every activation an exact
// zero (GELU(0) = SiLU(0) = Softsign(0) = 0: zero gates keep the residual stream's padding zero); the only places the true width
// matters are the row statistics -- LayerNorm over hidden / head_dim columns, the Softmax(dim=1) of ConditionEmbedder (conditions.py:68)
// and the attention scale 1 / sqrt(head_dim) -- and those kernels take it as an argument.  The public arena layout
// (ll_dit_param_info) is the checkpoint's own: callers never see the padding.
struct DitDims {
    int Ht, Hp, Hmt, Hmp, hd, hdp, Ha;
    bool padded;
};
static DitDims dit_dims(const LLDitConfig &c) {
    DitDims d;
    d.Ht = c.hidden; d.Hp = round_up(c.hidden, 64);
    d.Hmt = c.mlp_hidden; d.Hmp = round_up(c.mlp_hidden, 64);
    d.hd = c.hidden / c.heads; d.hdp = round_up(d.hd, 32);
    d.Ha = round_up(c.heads * d.hdp, 64);
    d.padded = d.Hp != d.Ht || d.Hmp != d.Hmt || d.hdp != d.hd || d.Ha != d.Ht;
    return d;
}

// ------------------------------------------------------------------------------------------ parameter layout
struct ParamInfo {
    std::string name;
    int64_t numel;   // checkpoint tensor
    int64_t offset;  // in the caller's arena (f32 elements)
    int rows, cols;  // cols == 0 for vectors
    int64_t inumel, ioffset;   // internal (padded) tensor and its offset in the engine's arena; == numel / offset when nothing is padded
    int irows, icols;
    PadMap map;
};

static std::vector<ParamInfo> dit_layout(const LLDitConfig &c) {
    std::vector<ParamInfo> v;
    int64_t off = 0, ioff = 0;
    const DitDims d = dit_dims(c);
    // (r, cc): checkpoint shape; (ir, ic): internal shape; vectors: cc = ic = 0
    auto add = [&](const std::string &n, int r, int cc, int ir, int ic, PadMap m = PadMap()) {
        const int64_t ne = (int64_t)r * (cc ? cc : 1), ine = (int64_t)ir * (ic ? ic : 1);
        v.push_back({n, ne, off, r, cc, ine, ioff, ir, ic, m});
        off += (ne + 63) / 64 * 64;  // 256-B aligned slots
        ioff += (ine + 63) / 64 * 64;
    };
    const int H = d.Ht, Hp = d.Hp, F = LL_XDIM + LL_EDIM * c.max_nodes, Hm = d.Hmt, Hmp = d.Hmp, hd = d.hd, hdp = d.hdp, Ha = d.Ha;
    auto vecH = [&](const std::string &n) { add(n, H, 0, Hp, 0); };
    auto matHH = [&](const std::string &n) { add(n, H, H, Hp, Hp); };
    PadMap rows6;  rows6.rg = H; rows6.rgp = Hp;                                         // k chunks of H rows -> chunks of Hp rows
    PadMap qkvm;   qkvm.rg2 = H; qkvm.rgp2 = Ha; qkvm.rg = hd; qkvm.rgp = hdp;          // (section, head, d) -> section * Ha + head * hdp + d
    PadMap projm;  projm.cg = hd; projm.cgp = hdp;                                       // columns (head, d) -> head * hdp + d
    add("x_embedder.0.weight", H, F, Hp, F);
    vecH("x_embedder.1.weight");
    vecH("x_embedder.1.bias");
    add("t_embedder.mlp.0.weight", H, 256, Hp, 256);
    vecH("t_embedder.mlp.0.bias");
    matHH("t_embedder.mlp.2.weight");
    vecH("t_embedder.mlp.2.bias");
    add("y_embedder.embedding_drop.weight", LL_YDIM, H, LL_YDIM, Hp);
    for (int i = 0; i < LL_YDIM; ++i) {
        const std::string p = "y_embedder.mlps." + std::to_string(i) + ".";
        add(p + "0.weight", H, 1, Hp, 1);
        vecH(p + "0.bias");
        matHH(p + "2.weight");
    }
    add("txt_embedder.embedding_drop.weight", 1, H, 1, Hp);
    add("txt_embedder.linear.weight", H, LL_TEXT_DIM, Hp, LL_TEXT_DIM);
    vecH("txt_embedder.linear.bias");
    for (int i = 0; i < c.depth; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        add(p + "attn.qkv.weight", 3 * H, H, 3 * Ha, Hp, qkvm);
        add(p + "attn.q_norm.weight", hd, 0, hdp, 0);
        add(p + "attn.q_norm.bias", hd, 0, hdp, 0);
        add(p + "attn.k_norm.weight", hd, 0, hdp, 0);
        add(p + "attn.k_norm.bias", hd, 0, hdp, 0);
        add(p + "attn.proj.weight", H, H, Hp, Ha, projm);
        vecH(p + "attn.proj.bias");
        add(p + "mlp.fc1.weight", Hm, H, Hmp, Hp);
        add(p + "mlp.fc1.bias", Hm, 0, Hmp, 0);
        add(p + "mlp.fc2.weight", H, Hm, Hp, Hmp);
        vecH(p + "mlp.fc2.bias");
        matHH(p + "adaLN_modulation.0.weight");
        vecH(p + "adaLN_modulation.0.bias");
        add(p + "adaLN_modulation.2.weight", 6 * H, H, 6 * Hp, Hp, rows6);
        add(p + "adaLN_modulation.2.bias", 6 * H, 0, 6 * Hp, 0, rows6);
    }
    matHH("output_layer.xedecoder.fc1.weight");
    vecH("output_layer.xedecoder.fc1.bias");
    add("output_layer.xedecoder.fc2.weight", F, H, F, Hp);
    add("output_layer.xedecoder.fc2.bias", F, 0, F, 0);
    matHH("output_layer.adaLN_modulation.0.weight");
    vecH("output_layer.adaLN_modulation.0.bias");
    add("output_layer.adaLN_modulation.2.weight", 2 * F, H, 2 * F, Hp);
    add("output_layer.adaLN_modulation.2.bias", 2 * F, 0, 2 * F, 0);
    return v;
}

// max_nodes <= 128 is the one bound the reference does not have (round 6: up to 64 nodes one 64-lane wave = one row of bond partners in
// the posterior and one 64-row attention tile; 65..128 take two waves / a 128-row tile); hidden <= 2048 and head_dim <= 128 bound the
// per-row register / LDS footprints of the row kernels.
// The Python wrapper reports all three as ValueError naming the limit (graph_decoder.py).
static int check_cfg(const LLDitConfig *c) {
    LL_CHECK(c != nullptr, "config is null");
    LL_CHECK(c->hidden >= 1 && c->hidden <= 2048, "hidden=%d must be in [1,2048]", c->hidden);
    LL_CHECK(c->heads > 0 && c->hidden % c->heads == 0, "hidden %d not divisible by heads %d", c->hidden, c->heads);
    LL_CHECK(c->hidden / c->heads <= 128, "head_dim %d > 128 unsupported", c->hidden / c->heads);
    LL_CHECK(c->mlp_hidden >= 1 && c->mlp_hidden <= 16384, "mlp_hidden=%d must be in [1,16384]", c->mlp_hidden);
    LL_CHECK(c->max_nodes >= 1 && c->max_nodes <= 128, "max_nodes=%d must be in [1,128]", c->max_nodes);
    LL_CHECK(c->depth >= 1 && c->T >= 1, "depth/T must be positive");
    LL_CHECK(c->dtype == LL_F32 || c->dtype == LL_BF16, "unknown dtype %d", c->dtype);
    return LL_OK;
}

// live guarded engine buffers of BOTH engines (LL_DEBUG_POISON=1 only; common.h)
static std::vector<std::pair<const void *, size_t>> g_guarded;
void debug_registry_add(const void *p, size_t n) { g_guarded.emplace_back(p, n); }
void debug_registry_remove(const void *p) {
    for (size_t i = 0; i < g_guarded.size(); ++i)
        if (g_guarded[i].first == p) {
            g_guarded.erase(g_guarded.begin() + i);
            return;
        }
}

// ------------------------------------------------------------------------------------------ engine
struct DevBuf {
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

struct DitEngine {
    LLDitConfig cfg;
    DitDims d;                   // checkpoint widths and the engine's padded pitches
    std::vector<ParamInfo> layout;
    int F, hd, esz;  // esz = operand element size
    // weights
    const float *w32 = nullptr;  // f32 master arena in the INTERNAL layout: the caller's arena itself (kept alive by the Python wrapper) when no
                                 // width needs padding, else `wpad`
    DevBuf wpad;                 // zero-padded f32 copy of the caller's arena (only when d.padded)
    DevBuf wop;                  // operand-dtype copy of the arena (bf16 mode)
    DevBuf wxT;                  // x_embedder weight transposed [F][H] f32
    DevBuf wycat;                // [H][10H] operand dtype
    DevBuf wqkvp;                // [depth][3H x H] q|k|v weights in MFMA A-operand order (pack_mfma16), bf16 mode
    DevBuf wfc1p, wfc2p;         // [depth][Hm x H], [depth][H x Hm]: the MLP weights packed the same way (gemm_m64_kernel)
    DevBuf wprojp;               // [depth][H x H]
    // in-situ kernel timing (ll_dit_class_probe): HIP events around every launch of ONE class of the block's kernels inside a real trajectory
    int time_class = -1;
    int time_mode = 0;           // 0 = the pair brackets the launch; LL_DIT_PROBE_EMPTY = an EMPTY pair is recorded at the launch site (what a pair
                                 // itself costs there); LL_DIT_PROBE_SKIP = no events, and the class is NOT launched (timing only: the trajectory's
                                 // results are meaningless; its run time against a normal one is the class's marginal cost inside the step)
    std::vector<hipEvent_t> tev;
    size_t tev_n = 0;
    std::vector<const void *> packed_keys;   // row-major weights registered with register_packed_weight
    DevBuf yw0, yb0;             // packed [10][H] f32
    DevBuf tables;               // x_marg16 e_marg8 u_xe80 u_ex80 betas[T+1] alphas_bar[T+1]
    // per-batch
    int B = 0, M2 = 0, M2p = 0, splits_h = 1, splits_m = 1;
    bool begun = false, state_set = false;
    DevBuf n_nodes, X, E, x32, xa, qkv, attn_o, ybuf, h1, ho, outF;
    DevBuf ct_in, ct_h, ct, zy, cy, txt_op, ctxt, ynan, tnan, c32, ca, m1, modtab, modo, rows, modcur;
    DevBuf scal;  // [0] int step, [8] u64 seed
    DevBuf predX, pxe;
    int state_half = 0;      // which half of X/E holds the current state
    bool state_both = false; // both halves identical (after set_state)
    // graph
    hipStream_t own = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    int graph_B = -1;
    // second captured step for trajectories that overlap another stream's work (ll_dit_set_overlap): panel GEMMs on the small-LDS ring
    hipGraph_t graph_ov = nullptr;
    hipGraphExec_t gexec_ov = nullptr;
    int graph_ov_B = -1;
    int overlap = 0;
    int last_steps = 0;
    bool timed = false;
    bool force_generic_attn = false;
    int fuse_qkv_attn = -1;          // q|k|v GEMM + attention in one launch: -1 = by workgroup count, 0 = never, 1 = whenever eligible

    std::unordered_map<std::string, size_t> index;      // parameter name -> layout entry (filled by index_params)
    void index_params() {
        index.clear();
        for (size_t i = 0; i < layout.size(); ++i) index.emplace(layout[i].name, i);
    }
    const float *pfs(const std::string &name) const {
        auto it = index.find(name);
        return it == index.end() ? nullptr : w32 + layout[it->second].ioffset;
    }
    const float *pf(const char *name) const { return pfs(std::string(name)); }
    const void *pw(const std::string &name) const {  // operand-dtype weight
        auto it = index.find(name);
        if (it == index.end()) return nullptr;
        const auto &p = layout[it->second];
        return cfg.dtype == LL_BF16 ? (const void *)(wop.as<bf16_t>() + p.ioffset) : (const void *)(w32 + p.ioffset);
    }
    // the block's parameters the per-step loop needs, resolved once (the loop launches ~200 kernels per millisecond: no string work there)
    struct BlockW {
        const void *qkv, *proj, *fc1, *fc2;
        const float *proj_b, *fc1_b, *fc2_b, *qn_w, *qn_b, *kn_w, *kn_b;
    };
    std::vector<BlockW> bw;
    const void *w_out1 = nullptr, *w_out2 = nullptr;
    const float *b_out1 = nullptr, *b_out2 = nullptr, *xe_w = nullptr, *xe_b = nullptr;
    void cache_block_params() {
        bw.resize(cfg.depth);
        for (int l = 0; l < cfg.depth; ++l) {
            const std::string p = "blocks." + std::to_string(l) + ".";
            BlockW &b = bw[l];
            b.qkv = pw(p + "attn.qkv.weight"); b.proj = pw(p + "attn.proj.weight");
            b.fc1 = pw(p + "mlp.fc1.weight"); b.fc2 = pw(p + "mlp.fc2.weight");
            b.proj_b = pfs(p + "attn.proj.bias"); b.fc1_b = pfs(p + "mlp.fc1.bias"); b.fc2_b = pfs(p + "mlp.fc2.bias");
            b.qn_w = pfs(p + "attn.q_norm.weight"); b.qn_b = pfs(p + "attn.q_norm.bias");
            b.kn_w = pfs(p + "attn.k_norm.weight"); b.kn_b = pfs(p + "attn.k_norm.bias");
        }
        w_out1 = pw("output_layer.xedecoder.fc1.weight"); w_out2 = pw("output_layer.xedecoder.fc2.weight");
        b_out1 = pf("output_layer.xedecoder.fc1.bias"); b_out2 = pf("output_layer.xedecoder.fc2.bias");
        xe_w = pf("x_embedder.1.weight"); xe_b = pf("x_embedder.1.bias");
    }
    // The step index the kernels read: the device scalar (captured graphs: advance_step_kernel counts it down), or -- when ll_dit_run
    // launches the kernels itself and knows the step -- entry `step_host` of a constant table, so that no launch waits for a value an
    // earlier launch wrote and the count-down launch disappears.
    DevBuf steps_tab;            // int [T + 1]: steps_tab[i] == i
    int step_host = -1;
    int *step_ptr() const { return step_host >= 0 ? steps_tab.as<int>() + step_host : scal.as<int>(); }
    int *step_scalar() const { return scal.as<int>(); }
    const int *rowvec = nullptr;   // per-graph table rows while ll_dit_denoise_rows runs, else null
    unsigned long long *seed_ptr() const { return reinterpret_cast<unsigned long long *>(scal.as<char>() + 8); }
    float *tab(int off) const { return tables.as<float>() + off; }
    float *t_xm() const { return tab(0); }
    float *t_em() const { return tab(16); }
    float *t_uxe() const { return tab(24); }
    float *t_uex() const { return tab(104); }
    float *t_beta() const { return tab(184); }
    float *t_ab() const { return tab(184 + cfg.T + 1); }
};

// Panel-GEMM choice of this engine for the calls made inside the scope (the flag gemm_dispatch consults is per host thread):
// overlap mode keeps <= 64-row panels on the 48 KB LDS-DMA ring, everything else on gemm_m64_kernel.
struct PanelScope {
    explicit PanelScope(const DitEngine *e) {
        static const bool keep = getenv("LL_OVERLAP_PANEL") && atoi(getenv("LL_OVERLAP_PANEL")) != 0;     // A/B switch: panel GEMMs in overlap mode too
        set_panel_gemm(!e->overlap || keep);
    }
    ~PanelScope() { set_panel_gemm(true); }
};

static void drop_graph(DitEngine *e) {
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    e->gexec = nullptr;
    e->graph = nullptr;
    e->graph_B = -1;
    if (e->gexec_ov) (void)hipGraphExecDestroy(e->gexec_ov);
    if (e->graph_ov) (void)hipGraphDestroy(e->graph_ov);
    e->gexec_ov = nullptr;
    e->graph_ov = nullptr;
    e->graph_ov_B = -1;
}

template <typename T> static void launch_embed(DitEngine *e, hipStream_t st) {
    hipLaunchKernelGGL((embed_kernel<T>), dim3(e->B * e->cfg.max_nodes), dim3(256), 0, st, e->X.as<int8_t>(),
                       e->E.as<int8_t>(), e->wxT.as<float>(), e->xe_w, e->xe_b,
                       e->x32.as<float>(), e->xa.as<T>(), e->step_ptr(), e->B, e->cfg.max_nodes, e->d.Hp, e->d.Ht);
}
static int g_attn_waves = 4;     // waves per (sequence, head) of attn_mfma_kernel (1 | 2 | 4; four only at head dimension 64)
static int g_fuse_qkv_pair_wgs = 320, g_fuse_qkv_pair_max_wgs = 768;   // between these many (sequence, head) pairs two sequences share a workgroup (batch 11..24 at 16 heads)
static int g_fuse_qkv_min_wgs = 64, g_fuse_qkv_max_wgs = 512;   // fuse_qkv_attn = -1: fuse when the launch has this many (sequence, head) workgroups

template <int NP, int HD>
static void launch_attn_mfma_t(DitEngine *e, const DitEngine::BlockW &w, hipStream_t st) {
#define LL_ATTN(W)                                                                                                     \
    hipLaunchKernelGGL((attn_mfma_kernel<NP, HD, W>), dim3(e->cfg.heads, 2 * e->B), dim3(64 * W), (attn_mfma_lds_bytes<NP, HD, W>()), st, \
                       e->qkv.as<bf16_t>(), e->attn_o.as<bf16_t>(), w.qn_w, w.qn_b,                                            \
                       w.kn_w, w.kn_b, e->n_nodes.as<int>(), e->B, e->cfg.max_nodes,                                           \
                       e->d.Ha, e->cfg.heads, e->d.hd)
    if constexpr (NP == 128) {      // 128 token rows always take four waves (two query tiles each): 32 accumulator tiles would not fit two
        LL_ATTN(4);
        return;
    } else if (g_attn_waves == 4 && HD == 64) LL_ATTN((HD == 64 ? 4 : 2));      // LayerNorm / transpose rows on four waves (eight rows each per pass)
    else if (g_attn_waves >= 2) LL_ATTN(2);
    else LL_ATTN(1);
#undef LL_ATTN
}
template <typename T> static void launch_attn(DitEngine *e, int layer, hipStream_t st) {
    const int N = e->cfg.max_nodes, hd = e->d.hd, hdp = e->d.hdp;
    const DitEngine::BlockW &w = e->bw[layer];
    if (sizeof(T) == 2 && !e->force_generic_attn) {      // bf16: the MFMA kernel at head pitch 32 | 64 | 96 | 128 (any head_dim <= 128)
        const int NP = N <= 32 ? 32 : N <= 64 ? 64 : 128;      // 65..128 nodes: two 64-key halves, eight 16-row tiles (four waves)
#define LL_ATTN_HD(HD) do { if (NP == 32) launch_attn_mfma_t<32, HD>(e, w, st); else if (NP == 64) launch_attn_mfma_t<64, HD>(e, w, st); \
                            else launch_attn_mfma_t<128, HD>(e, w, st); } while (0)
        if (hdp == 32) LL_ATTN_HD(32);
        else if (hdp == 64) LL_ATTN_HD(64);
        else if (hdp == 96) LL_ATTN_HD(96);
        else LL_ATTN_HD(128);
#undef LL_ATTN_HD
        return;
    }
    // K and V of the sequence + a chunk of QC query rows and their score rows in LDS: the largest of N, 64, 32, 16 that fits 160 KB
    int QC = N;
    auto lds_of = [&](int qc) { return (size_t)((2 * N + qc) * (hd + 1) + qc * (N + 1)) * 4; };
    while (QC > 16 && lds_of(QC) > 160 * 1024) QC = QC > 64 ? 64 : QC / 2;
    const size_t lds = lds_of(QC);
    hipLaunchKernelGGL((attn_generic_kernel<T>), dim3(e->cfg.heads, 2 * e->B), dim3(256), lds, st, e->qkv.as<T>(),
                       e->attn_o.as<T>(), w.qn_w, w.qn_b, w.kn_w, w.kn_b, e->n_nodes.as<int>(), e->B, N,
                       e->d.Ha, hd, hdp, QC);
}
// q|k|v projection + attention of block `layer` as ONE launch (qkv_attn_kernel); false = not eligible, run the two launches
static bool qkv_attn_eligible(const DitEngine *e) {
    const int H = e->cfg.hidden, N = e->cfg.max_nodes;
    if (e->cfg.dtype != LL_BF16 || e->cfg.heads * 64 != H || N > 64) return false;
    const int kc = std::min(H, N <= 32 ? 1024 : 512);                 // staged K chunk: 256 | 512 | 1024, dividing H
    return (kc == 256 || kc == 512 || kc == 1024) && H % kc == 0;
}
// two sequences per workgroup (qkv_attn_kernel<64, 512, true>): graphs of <= 32 nodes, hidden a multiple of 512
static bool qkv_attn_pair_eligible(const DitEngine *e) { return e->wqkvp.p != nullptr && e->cfg.max_nodes <= 32 && e->cfg.hidden % 512 == 0; }
// 0 = the two launches, 1 = one (sequence, head) per workgroup, 2 = two sequences per workgroup
static int qkv_attn_mode(const DitEngine *e) {
    if (e->wqkvp.p == nullptr || e->force_generic_attn || e->fuse_qkv_attn == 0) return 0;
    if (e->fuse_qkv_attn == 2) return qkv_attn_pair_eligible(e) ? 2 : 1;
    if (e->fuse_qkv_attn == 1) return 1;
    // measured: a workgroup takes its 448 KB in at the per-CU rate whatever the batch, so below ~64 workgroups the
    // two launches with 192+ workgroups each are faster, and beyond two rounds of workgroups per CU what the 128-row tiles of the GEMM save
    // exceeds what they save -- unless two sequences share a workgroup's weight stream (half the intake per sequence).
    // Next to another stream's kernels (overlap mode: the trajectory under the LLM decode) every launch costs
    // the other stream a dispatch slot as well: fused from batch 1 (e2e 369.4 -> 366.9 ms per molecule).
    const int wgs = 2 * e->B * e->cfg.heads;
    // ... up to the batch where the launch has more than ~four rounds of workgroups per CU: at 64 graphs (2048 workgroups, each streaming the
    // head's 384 KB of q|k|v weights) the fused launch took 66.6 us per block under the 64-sequence decode against 31 + 9 us for the 128-row
    // GEMM tiles + the attention launch (profiles/r6_llama64_kernel_stats.csv)
    if (e->overlap) return wgs <= 2 * g_fuse_qkv_max_wgs ? 1 : 0;
    // up to 128 token rows the q|k|v projection runs on the all-in-flight panel kernels (gemm_m64 / gemm_m128), which beat the fused launch
    // (graphs of more than 32 nodes stage their 64-row panel in two K chunks: there the fused launch needs twice the workgroups to pay)
    const int min_wgs = e->cfg.max_nodes <= 32 ? g_fuse_qkv_min_wgs : 2 * g_fuse_qkv_min_wgs;
    if (e->M2 <= 128 || wgs < min_wgs) return 0;
    // measured, ms per step unfused / one / two sequences per workgroup: B=8 1.61 / 1.50 / 1.65, B=12 2.12 / 2.07 / 2.01, B=16 2.22 / 2.22 / 2.15,
    // B=32 3.34 / 3.48 / 3.37
    if (wgs <= g_fuse_qkv_pair_wgs) return 1;
    if (qkv_attn_pair_eligible(e) && wgs <= g_fuse_qkv_pair_max_wgs) return 2;
    return wgs <= g_fuse_qkv_max_wgs ? 1 : 0;
}
static bool qkv_attn_wanted(const DitEngine *e) { return qkv_attn_mode(e) != 0; }
static void launch_qkv_attn(DitEngine *e, int layer, hipStream_t st) {
    const DitEngine::BlockW &w = e->bw[layer];
    const int N = e->cfg.max_nodes;
    const dim3 blk(768);
    if (qkv_attn_mode(e) == 2) {
        const dim3 grid(e->B * e->cfg.heads);
        if (e->cfg.hidden % 512 == 0)
            hipLaunchKernelGGL((qkv_attn_kernel<64, 512, true>), grid, blk, (QkvAttnGeom<64, 512>::lds_bytes()), st, e->xa.as<bf16_t>(),
                               e->wqkvp.as<bf16_t>() + (size_t)layer * 3 * e->cfg.hidden * e->cfg.hidden, e->attn_o.as<bf16_t>(), w.qn_w, w.qn_b,
                               w.kn_w, w.kn_b, e->n_nodes.as<int>(), e->B, N, e->cfg.hidden, e->cfg.heads);
        return;
    }
    const dim3 grid(2 * e->B * e->cfg.heads);
    const int kc = std::min(e->cfg.hidden, N <= 32 ? 1024 : 512);     // K chunk of the token panel staged in LDS
#define LL_QA(NP, KC)                                                                                                  \
    hipLaunchKernelGGL((qkv_attn_kernel<NP, KC>), grid, blk, (QkvAttnGeom<NP, KC>::lds_bytes()), st, e->xa.as<bf16_t>(),   \
                       e->wqkvp.as<bf16_t>() + (size_t)layer * 3 * e->cfg.hidden * e->cfg.hidden, e->attn_o.as<bf16_t>(), w.qn_w, \
                       w.qn_b, w.kn_w, w.kn_b, e->n_nodes.as<int>(), e->B, N, \
                       e->cfg.hidden, e->cfg.heads)
    if (N <= 32) { if (kc == 1024) LL_QA(32, 1024); else if (kc == 512) LL_QA(32, 512); else LL_QA(32, 256); }
    else { if (kc == 512) LL_QA(64, 512); else LL_QA(64, 256); }
#undef LL_QA
}
static int g_steps_per_graph = 5;      // reverse steps captured per hipGraph of ll_dit_run (when it divides T; env LL_STEPS_PER_GRAPH)
static int g_stage_mod = 1;           // stage the step's modulation rows at a fixed address (ll_set_stage_mod)
static int g_lnmod_multiwave = 1;     // one wave per 256-column chunk of a row (ln_mod_res_mw_kernel) instead of one wave per row

template <typename T>
static void launch_lnmod(DitEngine *e, int layer, int sel, int nslab, const float *bias, hipStream_t st) {
    const dim3 grid(e->M2), blk(64);
    const int H = e->d.Hp;      // row pitch; the kernels take the checkpoint's width for the statistics
    const int64_t ss = (int64_t)e->M2p * H;
    // the step's modulation rows at an address known at launch time: the slice of the hoisted table itself when the host knows the step
    // (ll_dit_run launching), else the copy stage_mod_kernel made (denoise_body)
    const float *mc = e->rowvec != nullptr ? nullptr
                      : e->step_host >= 0  ? e->modtab.as<float>() + (int64_t)e->step_host * (e->B + 1) * e->cfg.depth * 6 * H
                      : g_stage_mod        ? e->modcur.as<float>()
                                           : nullptr;
#define LL_LNMOD2(NS, ME)                                                                                              \
    do {                                                                                                               \
        if (ME > 1 && g_lnmod_multiwave)                                                                               \
            hipLaunchKernelGGL((ln_mod_res_mw_kernel<T, NS, ME>), grid, dim3(64 * ME), 0, st, e->ybuf.as<float>(), ss, bias, \
                               e->x32.as<float>(), e->xa.as<T>(), e->modtab.as<float>(), e->step_ptr(), e->rowvec, mc, layer, sel, \
                               e->B, e->cfg.max_nodes, H, e->cfg.depth, e->M2, e->d.Ht);                               \
        else                                                                                                           \
            hipLaunchKernelGGL((ln_mod_res_kernel<T, NS, ME>), grid, blk, 0, st, e->ybuf.as<float>(), ss, bias,         \
                               e->x32.as<float>(), e->xa.as<T>(), e->modtab.as<float>(), e->step_ptr(), e->rowvec, mc, layer, sel, \
                               e->B, e->cfg.max_nodes, H, e->cfg.depth, e->M2, e->d.Ht);                               \
    } while (0)
#define LL_LNMOD(NS)                                                                                                   \
    do {                                                                                                               \
        if (H <= 256) LL_LNMOD2(NS, 1);                                                                                \
        else if (H <= 512) LL_LNMOD2(NS, 2);                                                                           \
        else if (H <= 1024) LL_LNMOD2(NS, 4);                                                                          \
        else LL_LNMOD2(NS, 8);                                                                                         \
    } while (0)
    switch (nslab) {
        case 1: LL_LNMOD(1); break;
        case 2: LL_LNMOD(2); break;
        case 4: LL_LNMOD(4); break;
        default: LL_LNMOD(8); break;
    }
#undef LL_LNMOD2
#undef LL_LNMOD
}

// Events around one launch class inside the trajectory loop of ll_dit_run (launch mode only: events cannot be recorded into a captured step)
struct ClassTimer {
    DitEngine *e;
    hipStream_t st;
    bool on, skip;
    ClassTimer(DitEngine *e_, int cls, hipStream_t st_) : e(e_), st(st_) {
        const bool mine = e->time_class == cls && e->step_host >= 0;
        skip = mine && e->time_mode == LL_DIT_PROBE_SKIP;
        on = mine && !skip && e->tev_n + 2 <= e->tev.size();
        if (on) (void)hipEventRecord(e->tev[e->tev_n++], st);
        if (on && e->time_mode == LL_DIT_PROBE_EMPTY) {      // the pair closes before the launch: it times itself
            (void)hipEventRecord(e->tev[e->tev_n++], st);
            on = false;
        }
    }
    bool run() const { return !skip; }       // false: the probe asked for this class to be left out of the trajectory
    ~ClassTimer() {
        if (on) (void)hipEventRecord(e->tev[e->tev_n++], st);
    }
};

static int pick_splits(int M2, int H, int K) {
    if (M2 >= 2048 || (M2 >= 1024 && K >= 2048)) {
        // 128 x 128 tiles on sixteen-wave workgroups (gemm_dispatch): split K until the launch has ~256 of them
        const long t = (long)cdiv(M2, 128) * cdiv(H, 128);
        int s = 1;
        while (s < 4 && t * s < 256 && K / (s * 2) >= 512 && (K / (s * 2)) % 64 == 0) s *= 2;
        return s;
    }
    const long tiles = (long)cdiv(M2, 64) * cdiv(H, 64);
    int s = 1;
    // each split costs the consumer one more f32 slab to read back: cap at 4
    while (s < 4 && tiles * s < 256 && (K / (s * 2)) % 64 == 0 && K / (s * 2) >= 128) s *= 2;
    return s;
}

// denoiser on the current state for both passes -> e->outF [2][B][N][F] (decoder output before LN0/modulate)
static int denoise_body(DitEngine *e, hipStream_t st, float *hidden_tap, int tap_layer) {
    const LLDitConfig &c = e->cfg;
    const int H = e->d.Hp, Hm = e->d.Hmp, Ha = e->d.Ha, M2 = e->M2, dt = c.dtype;      // row pitches (multiples of 64)
    const bool bf = dt == LL_BF16;
    if (e->rowvec == nullptr && g_stage_mod && e->step_host < 0) {
        const int64_t row_floats = (int64_t)(e->B + 1) * c.depth * 6 * H;
        hipLaunchKernelGGL(stage_mod_kernel, dim3(256), dim3(256), 0, st, e->modtab.as<float>(), e->modcur.as<float>(), e->step_ptr(), row_floats);
    }
    if (bf) launch_embed<bf16_t>(e, st); else launch_embed<float>(e, st);
    LL_LAUNCH_CHECK();
    if (hidden_tap && tap_layer == 0)
        LL_HIP(hipMemcpy2DAsync(hidden_tap, (size_t)e->d.Ht * 4, e->x32.p, (size_t)H * 4, (size_t)e->d.Ht * 4, M2, hipMemcpyDeviceToDevice, st));
    const int64_t slab = (int64_t)e->M2p * H;
    const bool fused_qkv = qkv_attn_wanted(e);
    for (int l = 0; l < c.depth; ++l) {
        const DitEngine::BlockW &w = e->bw[l];
        if (fused_qkv) {
            ClassTimer tm(e, LL_DIT_CLS_QKV, st);
            launch_qkv_attn(e, l, st);
        } else {
            {
                ClassTimer tm(e, LL_DIT_CLS_QKV, st);
                LL_TRY(linear_launch(dt, e->xa.p, H, w.qkv, H, nullptr, e->qkv.p, 3 * Ha, M2, 3 * Ha, H, 0, 0, st));
            }
            ClassTimer tm(e, LL_DIT_CLS_ATTN, st);
            if (bf) launch_attn<bf16_t>(e, l, st); else launch_attn<float>(e, l, st);
        }
        LL_LAUNCH_CHECK();
        {
            ClassTimer tm(e, LL_DIT_CLS_PROJ, st);
            if (e->splits_h > 1)
                LL_TRY(linear_splitk_launch(dt, e->attn_o.p, Ha, w.proj, Ha, e->ybuf.as<float>(), H, slab, M2, H, Ha, e->splits_h, st));
            else
                LL_TRY(linear_launch(dt, e->attn_o.p, Ha, w.proj, Ha, nullptr, e->ybuf.p, H, M2, H, Ha, 0, 1, st));
        }
        {
            ClassTimer tm(e, LL_DIT_CLS_LNMOD, st);
            if (bf) launch_lnmod<bf16_t>(e, l, 0, e->splits_h, w.proj_b, st);
            else launch_lnmod<float>(e, l, 0, e->splits_h, w.proj_b, st);
        }
        LL_LAUNCH_CHECK();
        const int nslab_m = e->splits_m;
        {
            ClassTimer tm(e, LL_DIT_CLS_FC1, st);
            if (tm.run()) LL_TRY(linear_launch(dt, e->xa.p, H, w.fc1, H, w.fc1_b, e->h1.p, Hm, M2, Hm, H, 1, 0, st));
        }
        {
            ClassTimer tm(e, LL_DIT_CLS_FC2, st);
            if (e->splits_m > 1)
                LL_TRY(linear_splitk_launch(dt, e->h1.p, Hm, w.fc2, Hm, e->ybuf.as<float>(), H, slab, M2, H, Hm, e->splits_m, st));
            else
                LL_TRY(linear_launch(dt, e->h1.p, Hm, w.fc2, Hm, nullptr, e->ybuf.p, H, M2, H, Hm, 0, 1, st));
        }
        {
            ClassTimer tm(e, LL_DIT_CLS_LNMOD, st);
            if (bf) launch_lnmod<bf16_t>(e, l, 1, nslab_m, w.fc2_b, st);
            else launch_lnmod<float>(e, l, 1, nslab_m, w.fc2_b, st);
        }
        LL_LAUNCH_CHECK();
        if (hidden_tap && tap_layer == l + 1)
            LL_HIP(hipMemcpy2DAsync(hidden_tap, (size_t)e->d.Ht * 4, e->x32.p, (size_t)H * 4, (size_t)e->d.Ht * 4, M2, hipMemcpyDeviceToDevice, st));
    }
    LL_TRY(linear_launch(dt, e->xa.p, H, e->w_out1, H, e->b_out1, e->ho.p, H, M2, H, H, 1, 0, st));
    LL_TRY(linear_launch(dt, e->ho.p, H, e->w_out2, H, e->b_out2, e->outF.p, e->F, M2, e->F, H, 0, 1, st));
    return LL_OK;
}

static int posterior_launch(DitEngine *e, const float *qx, const float *qe, int update, float *pX, float *pE,
                            float *logX, float *logE, hipStream_t st) {
    PostArgs a;
    a.out = e->outF.as<float>();
    a.modo = e->modo.as<float>();
    a.predX = e->predX.as<float>();
    a.pxe = e->pxe.as<float>();
    a.X = e->X.as<int8_t>();
    a.E = e->E.as<int8_t>();
    a.n_nodes = e->n_nodes.as<int>();
    a.x_marg = e->t_xm(); a.e_marg = e->t_em(); a.u_xe = e->t_uxe(); a.u_ex = e->t_uex();
    a.betas = e->t_beta(); a.alphas_bar = e->t_ab();
    a.qx = qx; a.qe = qe;
    a.seed_ptr = e->seed_ptr();
    a.step_ptr = e->step_ptr();
    a.rowvec = e->rowvec;
    a.B = e->B; a.N = e->cfg.max_nodes; a.F = e->F; a.T = e->cfg.T;
    a.guide = e->cfg.guide_scale;
    a.pX_out = pX; a.pE_out = pE; a.logX = logX; a.logE = logE;
    a.update_state = update;
    if (pX) LL_HIP(hipMemsetAsync(pX, 0, (size_t)a.B * a.N * XD * 4, st));
    if (pE) LL_HIP(hipMemsetAsync(pE, 0, (size_t)a.B * a.N * a.N * ED * 4, st));
    if (a.N <= 64) {
        hipLaunchKernelGGL(post_rows_kernel<6>, dim3(cdiv(2 * a.B * a.N, 4)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(post_pairs_kernel<1>, dim3(a.N, a.B), dim3(64), 0, st, a);
    } else {        // 65..128 nodes: eleven 64-column chunks per decoder row, two wavefronts per row of bond partners
        hipLaunchKernelGGL(post_rows_kernel<11>, dim3(cdiv(2 * a.B * a.N, 4)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(post_pairs_kernel<2>, dim3(a.N, a.B), dim3(128), 0, st, a);
    }
    LL_LAUNCH_CHECK();
    return LL_OK;
}

// make half `want` of the double-buffered state hold the current state (copy from the live half if needed)
static int ensure_state_half(DitEngine *e, int want, hipStream_t st) {
    if (e->state_both || e->state_half == want) return LL_OK;
    const size_t nx = (size_t)e->B * e->cfg.max_nodes, ne = nx * e->cfg.max_nodes;
    LL_HIP(hipMemcpyAsync(e->X.as<int8_t>() + want * nx, e->X.as<int8_t>() + e->state_half * nx, nx, hipMemcpyDeviceToDevice, st));
    LL_HIP(hipMemcpyAsync(e->E.as<int8_t>() + want * ne, e->E.as<int8_t>() + e->state_half * ne, ne, hipMemcpyDeviceToDevice, st));
    e->state_both = true;
    return LL_OK;
}

static int check_ready(DitEngine *e, bool need_state) {
    LL_CHECK(e != nullptr, "null handle");
    if (!e->begun) { set_error("ll_dit_begin has not been called"); return LL_ESTATE; }
    if (need_state && !e->state_set) { set_error("no state: call ll_dit_init_state or ll_dit_set_state first"); return LL_ESTATE; }
    return LL_OK;
}

}  // namespace ll

using namespace ll;

// ============================================================================================ C ABI
extern "C" {

int ll_version(void) { return 100; }
const char *ll_last_error(void) { return g_err.c_str(); }

int ll_dit_param_count(const LLDitConfig *cfg) {
    if (check_cfg(cfg) != LL_OK) return LL_EINVAL;
    return (int)dit_layout(*cfg).size();
}
int ll_dit_param_info(const LLDitConfig *cfg, int idx, char *name, int name_cap, int64_t *numel, int64_t *offset) {
    LL_TRY(check_cfg(cfg));
    auto v = dit_layout(*cfg);
    LL_CHECK(idx >= 0 && idx < (int)v.size(), "param index %d out of range", idx);
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", v[idx].name.c_str());
    if (numel) *numel = v[idx].numel;
    if (offset) *offset = v[idx].offset;
    return LL_OK;
}
int64_t ll_dit_arena_elems(const LLDitConfig *cfg) {
    if (check_cfg(cfg) != LL_OK) return LL_EINVAL;
    auto v = dit_layout(*cfg);
    return v.back().offset + (v.back().numel + 63) / 64 * 64;
}

int ll_dit_create(const LLDitConfig *cfg, const LLDitTables *t, const float *d_weights_f32, void **handle) {
    LL_TRY(check_cfg(cfg));
    LL_CHECK(t && d_weights_f32 && handle, "null argument");
    DitEngine *e = new DitEngine();
    e->cfg = *cfg;
    e->layout = dit_layout(*cfg);
    e->index_params();
    e->F = LL_XDIM + LL_EDIM * cfg->max_nodes;
    e->d = dit_dims(*cfg);
    e->hd = e->d.hd;
    e->esz = cfg->dtype == LL_BF16 ? 2 : 4;
    const int H = e->d.Hp, T = cfg->T;      // H: row pitch of every hidden-wide vector from here on
    const int64_t elems = e->layout.back().ioffset + (e->layout.back().inumel + 63) / 64 * 64;
    auto fail = [&](int rc) { ll_dit_destroy(e); return rc; };
#define CR(x) do { int rc_ = (x); if (rc_ != LL_OK) return fail(rc_); } while (0)
#define CRH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e_)); return fail(LL_EHIP); } } while (0)
    if (e->d.padded) {
        // the checkpoint's tensors into the zero-padded internal arena (rows / columns / heads spread out as their PadMap says)
        CR(e->wpad.ensure((size_t)elems * 4));
        CRH(hipMemset(e->wpad.p, 0, (size_t)elems * 4));
        for (const ParamInfo &pi : e->layout) {
            const int R = pi.rows, Cc = pi.cols ? pi.cols : 1, ldd = pi.icols ? pi.icols : 1;
            const PadMap &m = pi.map;
            hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)std::min<int64_t>((pi.numel + 255) / 256, 4096)), dim3(256), 0, 0,
                               d_weights_f32 + pi.offset, e->wpad.as<float>() + pi.ioffset, R, Cc, ldd, m.rg2 ? m.rg2 : R, m.rgp2 ? m.rgp2 : R,
                               m.rg ? m.rg : R, m.rgp ? m.rgp : R, m.cg ? m.cg : Cc, m.cgp ? m.cgp : Cc);
        }
        CRH(hipGetLastError());
        CRH(hipDeviceSynchronize());
        e->w32 = e->wpad.as<float>();
    } else {
        e->w32 = d_weights_f32;
    }
    if (cfg->dtype == LL_BF16) {
        CR(e->wop.ensure((size_t)elems * 2));
        CR(convert_f32_to_bf16(e->w32, e->wop.as<bf16_t>(), elems, 0));
    }
    e->cache_block_params();          // after wop exists: pw() resolves into it
    // x_embedder weight transposed to [F][H] (gather-sum form)
    CR(e->wxT.ensure((size_t)e->F * H * 4));
    hipLaunchKernelGGL(transpose_kernel, dim3(256), dim3(256), 0, 0, e->pf("x_embedder.0.weight"), e->wxT.as<float>(), H, e->F);
    // property MLPs: packed first layers + K-concatenated second layers
    CR(e->yw0.ensure((size_t)LL_YDIM * H * 4));
    CR(e->yb0.ensure((size_t)LL_YDIM * H * 4));
    {
        std::vector<const float *> w2(LL_YDIM);
        for (int d = 0; d < LL_YDIM; ++d) {
            const std::string p = "y_embedder.mlps." + std::to_string(d) + ".";
            CRH(hipMemcpy(e->yw0.as<float>() + (size_t)d * H, e->pfs(p + "0.weight"), (size_t)H * 4, hipMemcpyDeviceToDevice));
            CRH(hipMemcpy(e->yb0.as<float>() + (size_t)d * H, e->pfs(p + "0.bias"), (size_t)H * 4, hipMemcpyDeviceToDevice));
            w2[d] = e->pfs(p + "2.weight");
        }
        DevBuf ptrs, cat32;
        CR(ptrs.ensure(sizeof(float *) * LL_YDIM));
        CRH(hipMemcpy(ptrs.p, w2.data(), sizeof(float *) * LL_YDIM, hipMemcpyHostToDevice));
        CR(cat32.ensure((size_t)H * LL_YDIM * H * 4));
        hipLaunchKernelGGL(ycat_kernel, dim3(64, LL_YDIM), dim3(256), 0, 0, (const float *const *)ptrs.p, cat32.as<float>(), H);
        if (cfg->dtype == LL_BF16) {
            CR(e->wycat.ensure((size_t)H * LL_YDIM * H * 2));
            CR(convert_f32_to_bf16(cat32.as<float>(), e->wycat.as<bf16_t>(), (int64_t)H * LL_YDIM * H, 0));
            CRH(hipDeviceSynchronize());
            cat32.release();
        } else {
            CRH(hipDeviceSynchronize());
            e->wycat = cat32;  // ownership moves
            cat32.p = nullptr;
        }
        CRH(hipDeviceSynchronize());
        ptrs.release();
    }
    // tables
    {
        std::vector<float> h(184 + 2 * (T + 1), 0.f);
        memcpy(&h[0], t->h_x_marg, 16 * 4);
        memcpy(&h[16], t->h_e_marg, 5 * 4);
        memcpy(&h[24], t->h_u_xe, 80 * 4);
        memcpy(&h[104], t->h_u_ex, 80 * 4);
        memcpy(&h[184], t->h_betas, (T + 1) * 4);
        memcpy(&h[184 + T + 1], t->h_alphas_bar, (T + 1) * 4);
        CR(e->tables.ensure(h.size() * 4));
        CRH(hipMemcpy(e->tables.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    CR(e->scal.ensure(64));
    {
        std::vector<int> ids(T + 1);
        for (int i = 0; i <= T; ++i) ids[i] = i;
        CR(e->steps_tab.ensure(ids.size() * 4));
        CRH(hipMemcpy(e->steps_tab.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    }
    // the generic attention kernel may need > 64 KiB of dynamic LDS (N=64, hd>=64)
    CRH(hipFuncSetAttribute((const void *)attn_generic_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_generic_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<128, 32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<128, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<128, 96, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<128, 128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#define LL_QA_ATTR(NP, KC) CRH(hipFuncSetAttribute((const void *)(qkv_attn_kernel<NP, KC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(QkvAttnGeom<NP, KC>::lds_bytes())))
    LL_QA_ATTR(32, 1024); LL_QA_ATTR(32, 512); LL_QA_ATTR(32, 256); LL_QA_ATTR(64, 512); LL_QA_ATTR(64, 256);
    CRH(hipFuncSetAttribute((const void *)(qkv_attn_kernel<64, 512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(QkvAttnGeom<64, 512>::lds_bytes())));
#undef LL_QA_ATTR
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<32, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<32, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<32, 64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<32, 64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<64, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<64, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<64, 64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<64, 64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<32, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CRH(hipFuncSetAttribute((const void *)attn_mfma_kernel<64, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    if (qkv_attn_eligible(e)) {
        const size_t per = (size_t)3 * H * H;
        CR(e->wqkvp.ensure(per * cfg->depth * 2));
        for (int l = 0; l < cfg->depth; ++l)
            CR(pack_mfma16(reinterpret_cast<const bf16_t *>(e->pw("blocks." + std::to_string(l) + ".attn.qkv.weight")),
                           e->wqkvp.as<bf16_t>() + per * l, 3 * H, H, 0));
    }
    if (cfg->dtype == LL_BF16) {
        // MFMA-operand-order copies of the MLP and proj weights: the batch-1 panel GEMM (gemm_m64_kernel) reads fragments from them
        const int Hm = e->d.Hmp, Ha = e->d.Ha;
        const size_t per = (size_t)Hm * H, perh = (size_t)H * Ha;
        CR(e->wfc1p.ensure(per * cfg->depth * 2));
        CR(e->wfc2p.ensure(per * cfg->depth * 2));
        CR(e->wprojp.ensure(perh * cfg->depth * 2));
        auto reg = [&](const std::string &name, const bf16_t *packed) {
            register_packed_weight(e->pw(name), packed);
            e->packed_keys.push_back(e->pw(name));
        };
        for (int l = 0; l < cfg->depth; ++l) {
            const std::string p = "blocks." + std::to_string(l) + ".";
            CR(pack_mfma16(reinterpret_cast<const bf16_t *>(e->pw(p + "mlp.fc1.weight")), e->wfc1p.as<bf16_t>() + per * l, Hm, H, 0));
            CR(pack_mfma16(reinterpret_cast<const bf16_t *>(e->pw(p + "mlp.fc2.weight")), e->wfc2p.as<bf16_t>() + per * l, H, Hm, 0));
            CR(pack_mfma16(reinterpret_cast<const bf16_t *>(e->pw(p + "attn.proj.weight")), e->wprojp.as<bf16_t>() + perh * l, H, Ha, 0));
            reg(p + "mlp.fc1.weight", e->wfc1p.as<bf16_t>() + per * l);
            reg(p + "mlp.fc2.weight", e->wfc2p.as<bf16_t>() + per * l);
            reg(p + "attn.proj.weight", e->wprojp.as<bf16_t>() + perh * l);
            if (e->wqkvp.p) reg(p + "attn.qkv.weight", e->wqkvp.as<bf16_t>() + (size_t)3 * H * H * l);
        }
    }
    e->force_generic_attn = getenv("LL_GENERIC_ATTN") != nullptr;
    if (const char *v = getenv("LL_FUSE_QKV_ATTN")) e->fuse_qkv_attn = atoi(v);
    if (const char *v = getenv("LL_FUSE_PAIR_WGS")) g_fuse_qkv_pair_wgs = atoi(v);
    if (const char *v = getenv("LL_STEPS_PER_GRAPH")) g_steps_per_graph = std::max(1, atoi(v));
    if (const char *ev = getenv("LL_STAGE_MOD")) g_stage_mod = atoi(ev) ? 1 : 0;      // A/B switch for bench runs
    CRH(hipStreamCreateWithFlags(&e->own, hipStreamNonBlocking));
    CRH(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
    CRH(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));
    CRH(hipEventCreate(&e->ev_t0));
    CRH(hipEventCreate(&e->ev_t1));
    CRH(hipDeviceSynchronize());
#undef CR
#undef CRH
    *handle = e;
    return LL_OK;
}

int ll_dit_destroy(void *handle) {
    DitEngine *e = (DitEngine *)handle;
    if (!e) return LL_OK;
    (void)hipDeviceSynchronize();
    drop_graph(e);
    DevBuf *bufs[] = {&e->wop, &e->wxT, &e->wycat, &e->yw0, &e->yb0, &e->tables, &e->n_nodes, &e->X, &e->E, &e->x32,
                      &e->xa, &e->qkv, &e->attn_o, &e->ybuf, &e->h1, &e->ho, &e->outF, &e->ct_in, &e->ct_h, &e->ct,
                      &e->zy, &e->cy, &e->txt_op, &e->ctxt, &e->ynan, &e->tnan, &e->c32, &e->ca, &e->m1, &e->modtab,
                      &e->modo, &e->scal, &e->predX, &e->pxe, &e->rows, &e->modcur, &e->wpad, &e->wqkvp, &e->wfc1p, &e->wfc2p, &e->wprojp, &e->steps_tab};
    for (const void *k : e->packed_keys) register_packed_weight(k, nullptr);
    for (DevBuf *b : bufs) b->release();
    if (e->own) (void)hipStreamDestroy(e->own);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    if (e->ev_t0) (void)hipEventDestroy(e->ev_t0);
    if (e->ev_t1) (void)hipEventDestroy(e->ev_t1);
    for (hipEvent_t ev : e->tev) (void)hipEventDestroy(ev);
    delete e;
    return LL_OK;
}

int ll_dit_begin(void *handle, int B, const float *props, const float *text, const int32_t *n_nodes, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_CHECK(e && props && text && n_nodes, "null argument");
    LL_CHECK(B >= 1 && B <= 4096, "batch %d out of range", B);
    hipStream_t st = (hipStream_t)stream;
    const LLDitConfig &c = e->cfg;
    const int H = e->d.Hp, Hm = e->d.Hmp, Ha = e->d.Ha, N = c.max_nodes, T = c.T, L = c.depth, F = e->F, dt = c.dtype, es = e->esz;
    const bool bf = dt == LL_BF16;
    if (B != e->B) drop_graph(e);
    e->B = B;
    e->M2 = 2 * B * N;
    e->M2p = round_up(e->M2, 128);
    e->splits_h = pick_splits(e->M2, H, Ha);
    e->splits_m = pick_splits(e->M2, H, Hm);
    const int smax = std::max(e->splits_h, e->splits_m);
    const int Mc = (T + 1) * (B + 1), Mcp = round_up(Mc, 128);   // rows 0..T-1: reverse steps (t = s+1); row T: t = 0 (training)
    const int Tp = round_up(T + 1, 128), Bp = round_up(B, 128);
    const size_t M2p = e->M2p;
    void *oldp[] = {e->x32.p, e->xa.p, e->qkv.p, e->attn_o.p, e->ybuf.p, e->h1.p, e->ho.p, e->outF.p, e->modtab.p, e->modo.p, e->X.p, e->E.p, e->n_nodes.p, e->modcur.p};
    LL_TRY(e->n_nodes.ensure((size_t)B * 4));
    LL_TRY(e->X.ensure((size_t)2 * B * N));
    LL_TRY(e->E.ensure((size_t)2 * B * N * N));
    LL_TRY(e->predX.ensure((size_t)2 * B * N * XD * 4));
    LL_TRY(e->pxe.ensure((size_t)2 * B * N * 8 * 4));
    LL_TRY(e->x32.ensure(M2p * H * 4));
    LL_TRY(e->xa.ensure(M2p * H * es));
    LL_TRY(e->qkv.ensure(M2p * 3 * Ha * es));
    LL_TRY(e->attn_o.ensure(M2p * Ha * es));
    // the generic attention kernel writes only the true head columns: the padded ones (zero columns of proj) must not hold NaN patterns
    if (e->d.padded) LL_HIP(hipMemsetAsync(e->attn_o.p, 0, M2p * Ha * es, st));
    LL_TRY(e->ybuf.ensure((size_t)smax * M2p * H * 4));
    LL_TRY(e->h1.ensure(M2p * Hm * es));
    LL_TRY(e->ho.ensure(M2p * H * es));
    LL_TRY(e->outF.ensure(M2p * F * 4));
    LL_TRY(e->ct_in.ensure((size_t)Tp * 256 * es));
    LL_TRY(e->ct_h.ensure((size_t)Tp * H * es));
    LL_TRY(e->ct.ensure((size_t)Tp * H * 4));
    LL_TRY(e->zy.ensure((size_t)Bp * LL_YDIM * H * es));
    LL_TRY(e->cy.ensure((size_t)Bp * H * 4));
    LL_TRY(e->txt_op.ensure((size_t)Bp * LL_TEXT_DIM * es));
    LL_TRY(e->ctxt.ensure((size_t)Bp * H * 4));
    LL_TRY(e->ynan.ensure((size_t)B * LL_YDIM));
    LL_TRY(e->tnan.ensure((size_t)B));
    LL_TRY(e->c32.ensure((size_t)Mcp * H * 4));
    LL_TRY(e->ca.ensure((size_t)Mcp * H * es));
    LL_TRY(e->m1.ensure((size_t)Mcp * H * es));
    LL_TRY(e->modtab.ensure((size_t)Mc * L * 6 * H * 4));
    LL_TRY(e->modo.ensure((size_t)Mc * 2 * F * 4));
    LL_TRY(e->modcur.ensure((size_t)(B + 1) * L * 6 * H * 4));
    void *newp[] = {e->x32.p, e->xa.p, e->qkv.p, e->attn_o.p, e->ybuf.p, e->h1.p, e->ho.p, e->outF.p, e->modtab.p, e->modo.p, e->X.p, e->E.p, e->n_nodes.p, e->modcur.p};
    for (size_t i = 0; i < sizeof(oldp) / sizeof(oldp[0]); ++i)
        if (oldp[i] != newp[i]) { drop_graph(e); break; }

    LL_HIP(hipMemcpyAsync(e->n_nodes.p, n_nodes, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    // ---- c_t for every step: sinusoid -> Linear(256,H)+SiLU -> Linear(H,H)          (conditions.py:53-58)
    if (bf) hipLaunchKernelGGL((tfreq_kernel<bf16_t>), dim3(T + 1), dim3(128), 0, st, e->ct_in.as<bf16_t>(), T);
    else hipLaunchKernelGGL((tfreq_kernel<float>), dim3(T + 1), dim3(128), 0, st, e->ct_in.as<float>(), T);
    LL_LAUNCH_CHECK();
    LL_TRY(linear_launch(dt, e->ct_in.p, 256, e->pw("t_embedder.mlp.0.weight"), 256, e->pf("t_embedder.mlp.0.bias"), e->ct_h.p, H, T + 1, H, 256, 2, 0, st));
    LL_TRY(linear_launch(dt, e->ct_h.p, H, e->pw("t_embedder.mlp.2.weight"), H, e->pf("t_embedder.mlp.2.bias"), e->ct.p, H, T + 1, H, H, 0, 1, st));
    // ---- c_y: softmax features -> one GEMM over the K-concatenated property MLPs      (conditions.py:60-98)
    if (bf) hipLaunchKernelGGL((yfeat_kernel<bf16_t>), dim3(B, LL_YDIM), dim3(256), 0, st, props, e->yw0.as<float>(), e->yb0.as<float>(), e->zy.as<bf16_t>(), e->ynan.as<int8_t>(), H, e->d.Ht);
    else hipLaunchKernelGGL((yfeat_kernel<float>), dim3(B, LL_YDIM), dim3(256), 0, st, props, e->yw0.as<float>(), e->yb0.as<float>(), e->zy.as<float>(), e->ynan.as<int8_t>(), H, e->d.Ht);
    LL_LAUNCH_CHECK();
    LL_TRY(linear_launch(dt, e->zy.p, LL_YDIM * H, e->wycat.p, LL_YDIM * H, nullptr, e->cy.p, H, B, H, LL_YDIM * H, 0, 1, st));
    // ---- c_txt                                                                        (conditions.py:100-123)
    if (bf) hipLaunchKernelGGL((txt_prep_kernel<bf16_t>), dim3(B), dim3(256), 0, st, text, e->txt_op.as<bf16_t>(), e->tnan.as<int8_t>(), LL_TEXT_DIM);
    else hipLaunchKernelGGL((txt_prep_kernel<float>), dim3(B), dim3(256), 0, st, text, e->txt_op.as<float>(), e->tnan.as<int8_t>(), LL_TEXT_DIM);
    LL_LAUNCH_CHECK();
    LL_TRY(linear_launch(dt, e->txt_op.p, LL_TEXT_DIM, e->pw("txt_embedder.linear.weight"), LL_TEXT_DIM, e->pf("txt_embedder.linear.bias"), e->ctxt.p, H, B, H, LL_TEXT_DIM, 0, 1, st));
    // ---- c[s][ci]
    if (bf) hipLaunchKernelGGL((combine_c_kernel<bf16_t>), dim3(T + 1, B + 1), dim3(256), 0, st, e->ct.as<float>(), e->cy.as<float>(), e->ctxt.as<float>(), e->pf("y_embedder.embedding_drop.weight"), e->pf("txt_embedder.embedding_drop.weight"), e->ynan.as<int8_t>(), e->tnan.as<int8_t>(), e->c32.as<float>(), e->ca.as<bf16_t>(), B, H);
    else hipLaunchKernelGGL((combine_c_kernel<float>), dim3(T + 1, B + 1), dim3(256), 0, st, e->ct.as<float>(), e->cy.as<float>(), e->ctxt.as<float>(), e->pf("y_embedder.embedding_drop.weight"), e->pf("txt_embedder.embedding_drop.weight"), e->ynan.as<int8_t>(), e->tnan.as<int8_t>(), e->c32.as<float>(), e->ca.as<float>(), B, H);
    LL_LAUNCH_CHECK();
    // ---- all adaLN modulations for all steps: Linear(H,H)+SiLU -> Linear(H,6H)+Softsign  (transformer.py:125-130)
    for (int l = 0; l < L; ++l) {
        const std::string p = "blocks." + std::to_string(l) + ".adaLN_modulation.";
        LL_TRY(linear_launch(dt, e->ca.p, H, e->pw(p + "0.weight"), H, e->pfs(p + "0.bias"), e->m1.p, H, Mc, H, H, 2, 0, st));
        LL_TRY(linear_launch(dt, e->m1.p, H, e->pw(p + "2.weight"), H, e->pfs(p + "2.bias"), e->modtab.as<float>() + (size_t)l * 6 * H, L * 6 * H, Mc, 6 * H, H, 3, 1, st));
    }
    LL_TRY(linear_launch(dt, e->ca.p, H, e->pw("output_layer.adaLN_modulation.0.weight"), H, e->pf("output_layer.adaLN_modulation.0.bias"), e->m1.p, H, Mc, H, H, 2, 0, st));
    LL_TRY(linear_launch(dt, e->m1.p, H, e->pw("output_layer.adaLN_modulation.2.weight"), H, e->pf("output_layer.adaLN_modulation.2.bias"), e->modo.p, 2 * F, Mc, 2 * F, H, 0, 1, st));
    e->begun = true;
    e->state_set = false;
    return LL_OK;
}

int ll_dit_init_state(void *handle, const float *qx, const float *qe, uint64_t seed, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, false));
    LL_CHECK((qx == nullptr) == (qe == nullptr), "qx and qe must both be given or both be null");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, e->step_scalar(), e->cfg.T - 1, e->seed_ptr(), (unsigned long long)seed);
    {
        const int half = e->cfg.T & 1;
        const size_t nx = (size_t)e->B * e->cfg.max_nodes, ne = nx * e->cfg.max_nodes;
        hipLaunchKernelGGL(init_state_kernel, dim3(e->B), dim3(256), 0, st, e->X.as<int8_t>() + half * nx, e->E.as<int8_t>() + half * ne,
                           e->n_nodes.as<int>(), e->t_xm(), e->t_em(), qx, qe, e->seed_ptr(), e->B, e->cfg.max_nodes, e->cfg.T);
        e->state_half = half;
        e->state_both = false;
    }
    LL_LAUNCH_CHECK();
    e->state_set = true;
    return LL_OK;
}

int ll_dit_set_state(void *handle, const int8_t *X, const int8_t *E, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, false));
    LL_CHECK(X && E, "null state");
    const int N = e->cfg.max_nodes;
    for (int h = 0; h < 2; ++h) {
        LL_HIP(hipMemcpyAsync(e->X.as<int8_t>() + (size_t)h * e->B * N, X, (size_t)e->B * N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        LL_HIP(hipMemcpyAsync(e->E.as<int8_t>() + (size_t)h * e->B * N * N, E, (size_t)e->B * N * N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    e->state_half = 0;
    e->state_both = true;
    e->state_set = true;
    return LL_OK;
}

int ll_dit_get_state(void *handle, int8_t *X, int8_t *E, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, true));
    const int N = e->cfg.max_nodes;
    const size_t hx = (size_t)e->state_half * e->B * N, he = hx * N;
    if (X) LL_HIP(hipMemcpyAsync(X, e->X.as<int8_t>() + hx, (size_t)e->B * N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (E) LL_HIP(hipMemcpyAsync(E, e->E.as<int8_t>() + he, (size_t)e->B * N * N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return LL_OK;
}

int ll_dit_step(void *handle, int s, const float *qx, const float *qe, uint64_t seed, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, true));
    LL_CHECK(s >= 0 && s < e->cfg.T, "step %d out of range [0,%d)", s, e->cfg.T);
    LL_CHECK((qx == nullptr) == (qe == nullptr), "qx and qe must both be given or both be null");
    hipStream_t st = (hipStream_t)stream;
    PanelScope panel(e);
    LL_TRY(ensure_state_half(e, (s + 1) & 1, st));
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, e->step_scalar(), s, e->seed_ptr(), (unsigned long long)seed);
    LL_LAUNCH_CHECK();
    LL_TRY(denoise_body(e, st, nullptr, -1));
    LL_TRY(posterior_launch(e, qx, qe, 1, nullptr, nullptr, nullptr, nullptr, st));
    e->state_half = s & 1;
    e->state_both = false;
    return LL_OK;
}

int ll_dit_denoise(void *handle, int s, float *logX, float *logE, float *hidden, int tap_layer, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, true));
    LL_CHECK(s >= 0 && s < e->cfg.T, "step %d out of range", s);
    hipStream_t st = (hipStream_t)stream;
    PanelScope panel(e);
    LL_TRY(ensure_state_half(e, (s + 1) & 1, st));
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, e->step_scalar(), s, e->seed_ptr(), 0ull);
    LL_TRY(denoise_body(e, st, hidden, tap_layer));
    return posterior_launch(e, nullptr, nullptr, 0, nullptr, nullptr, logX, logE, st);
}

int ll_dit_denoise_rows(void *handle, const int32_t *t_int, float *logX, float *logE, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, true));
    LL_CHECK(t_int && logX && logE, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const int T = e->cfg.T;
    PanelScope panel(e);
    // the state the denoiser reads is the one a reverse step s = T-1 would read (set by ll_dit_set_state)
    LL_TRY(ensure_state_half(e, T & 1, st));
    LL_TRY(e->rows.ensure((size_t)e->B * 4));
    hipLaunchKernelGGL(t_to_row_kernel, dim3(1), dim3(256), 0, st, t_int, e->rows.as<int>(), e->B, T);
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, e->step_scalar(), T - 1, e->seed_ptr(), 0ull);
    LL_LAUNCH_CHECK();
    e->rowvec = e->rows.as<int>();
    int rc = denoise_body(e, st, nullptr, -1);
    if (rc == LL_OK) rc = posterior_launch(e, nullptr, nullptr, 0, nullptr, nullptr, logX, logE, st);
    e->rowvec = nullptr;
    return rc;
}

int ll_dit_step_probs(void *handle, int s, float *pX, float *pE, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, true));
    LL_CHECK(s >= 0 && s < e->cfg.T, "step %d out of range", s);
    hipStream_t st = (hipStream_t)stream;
    PanelScope panel(e);
    LL_TRY(ensure_state_half(e, (s + 1) & 1, st));
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, e->step_scalar(), s, e->seed_ptr(), 0ull);
    LL_TRY(denoise_body(e, st, nullptr, -1));
    return posterior_launch(e, nullptr, nullptr, 0, pX, pE, nullptr, nullptr, st);
}

int ll_dit_cvec(void *handle, int s, float *c, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, false));
    LL_CHECK(s >= 0 && s < e->cfg.T && c, "bad argument");
    // c [B + 1][hidden] in the checkpoint's width (the engine's rows have pitch Hp)
    const size_t Ht = (size_t)e->d.Ht, Hp = (size_t)e->d.Hp;
    LL_HIP(hipMemcpy2DAsync(c, Ht * 4, e->c32.as<float>() + (size_t)s * (e->B + 1) * Hp, Hp * 4, Ht * 4, (size_t)e->B + 1, hipMemcpyDeviceToDevice,
                            (hipStream_t)stream));
    return LL_OK;
}

int ll_dit_run(void *handle, uint64_t seed, int use_graph, void *stream) {
    DitEngine *e = (DitEngine *)handle;
    LL_TRY(check_ready(e, true));
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t st = e->own;
    const int T = e->cfg.T;
    LL_HIP(hipEventRecord(e->ev_in, caller));
    LL_HIP(hipStreamWaitEvent(st, e->ev_in, 0));
    LL_TRY(ensure_state_half(e, T & 1, st));
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, e->step_scalar(), T - 1, e->seed_ptr(), (unsigned long long)seed);
    LL_LAUNCH_CHECK();
    // overlap mode: the trajectory runs next to another stream's kernels (the LLM decode of the next prompt); gemm_m64_kernel's
    // workgroups need a whole CU's LDS and keep that stream's workgroups off the CU (and wait for a drained CU themselves), so the
    // panel GEMMs take the 48 KB LDS-DMA ring there: 1.36 instead of 1.17 ms per step alone, but +1.2 % molecules/s end to end
    PanelScope panel(e);      // restored on every exit path: other engines / the GIN path keep the panel kernel
    if (use_graph == LL_DIT_RUN_AUTO) {      // env LL_DIT_RUN_MODE = graph | launches overrides the library's choice (e.g. a host too busy to feed launches)
        static const int forced = [] {
            const char *v = getenv("LL_DIT_RUN_MODE");
            return !v ? -1 : (strcmp(v, "graph") == 0 ? LL_DIT_RUN_GRAPH : strcmp(v, "launches") == 0 ? LL_DIT_RUN_LAUNCHES : -1);
        }();
        use_graph = forced >= 0 ? forced : (e->overlap ? LL_DIT_RUN_GRAPH : LL_DIT_RUN_LAUNCHES);
    }
    if (use_graph == LL_DIT_RUN_GRAPH) {
        hipGraph_t &graph = e->overlap ? e->graph_ov : e->graph;
        hipGraphExec_t &gexec = e->overlap ? e->gexec_ov : e->gexec;
        int &graph_B = e->overlap ? e->graph_ov_B : e->graph_B;
        // consecutive graph launches leave ~9 us between them on the device; five steps per graph share one such gap
        const int spg = g_steps_per_graph > 1 && T % g_steps_per_graph == 0 ? g_steps_per_graph : 1;
        if (!gexec || graph_B != e->B) {
            if (gexec) (void)hipGraphExecDestroy(gexec);
            if (graph) (void)hipGraphDestroy(graph);
            gexec = nullptr;
            graph = nullptr;
            LL_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = LL_OK;
            for (int g = 0; g < spg && rc == LL_OK; ++g) {        // `spg` reverse steps per captured graph: the step index lives in device memory
                rc = denoise_body(e, st, nullptr, -1);
                if (rc == LL_OK) rc = posterior_launch(e, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, st);
                if (rc == LL_OK) hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, st, e->step_ptr());
            }
            hipError_t ce = hipStreamEndCapture(st, &graph);
            if (rc != LL_OK) return rc;
            LL_HIP(ce);
            LL_HIP(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
            graph_B = e->B;
        }
        LL_HIP(hipEventRecord(e->ev_t0, st));
        for (int i = 0; i < T / spg; ++i) LL_HIP(hipGraphLaunch(gexec, st));
        LL_HIP(hipEventRecord(e->ev_t1, st));
    } else {
        LL_HIP(hipEventRecord(e->ev_t0, st));
        int rc = LL_OK;
        for (int i = 0; i < T && rc == LL_OK; ++i) {
            e->step_host = T - 1 - i;        // the kernels read this step from the constant table: no count-down launch, no staged rows
            rc = denoise_body(e, st, nullptr, -1);
            if (rc == LL_OK) rc = posterior_launch(e, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, st);
        }
        e->step_host = -1;
        LL_TRY(rc);
        hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, e->step_scalar(), -1, e->seed_ptr(), (unsigned long long)seed);   // as the count-down leaves it
        LL_HIP(hipEventRecord(e->ev_t1, st));
    }
    e->last_steps = T;
    e->timed = true;
    e->state_half = 0;      // z_0 lands in half 0
    e->state_both = false;
    LL_HIP(hipEventRecord(e->ev_out, st));
    LL_HIP(hipStreamWaitEvent(caller, e->ev_out, 0));
    return LL_OK;
}

int ll_dit_set_overlap(void *handle, int on) {
    DitEngine *e = (DitEngine *)handle;
    LL_CHECK(e, "ll_dit_set_overlap: null handle");
    e->overlap = on ? 1 : 0;
    return LL_OK;
}

int ll_dit_set_option(void *handle, int option, int value) {
    DitEngine *e = (DitEngine *)handle;
    LL_CHECK(e, "ll_dit_set_option: null handle");
    switch (option) {
        case LL_DIT_OPT_OVERLAP: e->overlap = value ? 1 : 0; break;
        case LL_DIT_OPT_GENERIC_ATTN:
            if (e->force_generic_attn != (value != 0)) drop_graph(e);      // captured steps hold the other kernel
            e->force_generic_attn = value != 0;
            break;
        case LL_DIT_OPT_FUSED_QKV_ATTN:
            if (e->fuse_qkv_attn != value) drop_graph(e);
            e->fuse_qkv_attn = value < 0 ? -1 : (value >= 2 ? 2 : value ? 1 : 0);
            break;
        default: LL_CHECK(false, "ll_dit_set_option: unknown option %d", option);
    }
    return LL_OK;
}

// ---- test probes of the on-device noise source (include/llamole_hip_tuning.h)
static __global__ void philox_probe_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 r = philox4x32(make_uint4(in[i * 6 + 0], in[i * 6 + 1], in[i * 6 + 2], in[i * 6 + 3]), make_uint2(in[i * 6 + 4], in[i * 6 + 5]));
    out[i * 4 + 0] = r.x, out[i * 4 + 1] = r.y, out[i * 4 + 2] = r.z, out[i * 4 + 3] = r.w;
}

// the Exp(1) variates a reverse step s (z_T: s = T) draws for every (graph, node, atom class) and (graph, i, j, bond class), through the
// same dit_noise_x4 / dit_noise_e4 / exp1_from_bits the sampling kernels call
static __global__ void noise_probe_kernel(unsigned long long seed, int s, int B, int N, float *__restrict__ qx, float *__restrict__ qe) {
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    const int64_t nx = (int64_t)B * N * 4, ne = (int64_t)B * N * N * 2;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nx + ne; t += (int64_t)gridDim.x * 256) {
        if (t < nx) {
            const int node = (int)(t >> 2), g = (int)(t & 3);
            const uint4 r = dit_noise_x4(key, s, node, g);
            float *o = qx + (int64_t)node * XD + g * 4;
            o[0] = exp1_from_bits(r.x), o[1] = exp1_from_bits(r.y), o[2] = exp1_from_bits(r.z), o[3] = exp1_from_bits(r.w);
        } else {
            const int64_t u = t - nx;
            const int pair = (int)(u >> 1), g = (int)(u & 1);
            const uint4 r = dit_noise_e4(key, s, pair, g);
            const float v[4] = {exp1_from_bits(r.x), exp1_from_bits(r.y), exp1_from_bits(r.z), exp1_from_bits(r.w)};
            for (int k = 0; k < 4; ++k)
                if (g * 4 + k < ED) qe[(int64_t)pair * ED + g * 4 + k] = v[k];
        }
    }
}

#if LL_TUNING
int ll_philox_probe(const uint32_t *ctr_key, uint32_t *out, int n, void *stream) {
    LL_CHECK(ctr_key && out && n >= 1, "ll_philox_probe: bad argument");
    hipLaunchKernelGGL(philox_probe_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ctr_key, out, n);
    LL_LAUNCH_CHECK();
    return LL_OK;
}
#endif

#if LL_TUNING
int ll_dit_noise_probe(uint64_t seed, int s, int B, int N, float *qx, float *qe, void *stream) {
    LL_CHECK(qx && qe && B >= 1 && N >= 1 && s >= 0 && (int64_t)B * N * N < (1ll << 31), "ll_dit_noise_probe: bad argument");
    hipLaunchKernelGGL(noise_probe_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed, s, B, N, qx, qe);
    LL_LAUNCH_CHECK();
    return LL_OK;
}
#endif

#if LL_TUNING
int ll_dit_class_probe(void *handle, int cls) {
    DitEngine *e = (DitEngine *)handle;
    LL_CHECK(e, "ll_dit_class_probe: null handle");
    const int mode = cls < 0 ? 0 : (cls & (LL_DIT_PROBE_EMPTY | LL_DIT_PROBE_SKIP));
    if (cls >= 0) cls &= ~(LL_DIT_PROBE_EMPTY | LL_DIT_PROBE_SKIP);
    LL_CHECK(cls >= -1 && cls < LL_DIT_CLS_COUNT, "ll_dit_class_probe: unknown class %d", cls);
    LL_CHECK(mode != LL_DIT_PROBE_SKIP || cls == LL_DIT_CLS_FC1, "ll_dit_class_probe: only the fc1 class can be left out");
    LL_CHECK(mode != (LL_DIT_PROBE_EMPTY | LL_DIT_PROBE_SKIP), "ll_dit_class_probe: EMPTY and SKIP exclude each other");
    e->time_class = cls;
    e->time_mode = mode;
    e->tev_n = 0;
    if (cls >= 0 && e->tev.empty()) {
        const size_t n = (size_t)2 * 2 * e->cfg.depth * e->cfg.T;        // at most two launches of a class per block
        e->tev.resize(n);
        for (size_t i = 0; i < n; ++i) LL_HIP(hipEventCreate(&e->tev[i]));
    }
    return LL_OK;
}
#endif

#if LL_TUNING
int ll_dit_class_probe_read(void *handle, float *total_us, int *launches) {
    DitEngine *e = (DitEngine *)handle;
    LL_CHECK(e && total_us && launches, "ll_dit_class_probe_read: null argument");
    LL_HIP(hipDeviceSynchronize());
    double us = 0.0;
    for (size_t i = 0; i + 1 < e->tev_n; i += 2) {
        float ms = 0.f;
        LL_HIP(hipEventElapsedTime(&ms, e->tev[i], e->tev[i + 1]));
        us += 1e3 * ms;
    }
    *total_us = (float)us;
    *launches = (int)(e->tev_n / 2);
    e->tev_n = 0;
    return LL_OK;
}
#endif

#if LL_TUNING
int ll_debug_check_guards(void) {
    if (!ll::debug_poison()) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    int bad = 0;
    for (const auto &b : ll::g_guarded) bad += ll::debug_guard_damage(b.first, b.second, "ll_debug_check_guards");
    return bad;
}
#endif

#if LL_TUNING
int ll_debug_guard_selftest(void) {
    if (!ll::debug_poison()) return -1;
    void *p = nullptr;
    if (ll::debug_alloc(&p, 1000) != LL_OK) return -2;
    const int clean = ll_debug_check_guards();
    (void)hipMemset((char *)p + 1000 + 17, 0, 1);      // what an overrunning kernel would do
    const int after = ll_debug_check_guards();
    ll::debug_registry_remove(p);
    (void)hipFree(p);
    return (clean == 0 && after == 1) ? 1 : 0;
}
#endif

#if LL_TUNING
int ll_set_attn_waves(int waves) {
    const int old = g_attn_waves;
    if (waves == 1 || waves == 2 || waves == 4) g_attn_waves = waves;
    return old;
}
#endif

#if LL_TUNING
int ll_set_stage_mod(int on) {
    const int old = g_stage_mod;
    g_stage_mod = on ? 1 : 0;
    return old;
}
#endif

#if LL_TUNING
int ll_set_lnmod_multiwave(int on) {
    const int old = g_lnmod_multiwave;
    g_lnmod_multiwave = on ? 1 : 0;
    return old;
}
#endif

int ll_dit_last_run_ms(void *handle, float *ms, int *steps) {
    DitEngine *e = (DitEngine *)handle;
    LL_CHECK(e && ms && steps, "null argument");
    if (!e->timed) { set_error("no ll_dit_run has been issued"); return LL_ESTATE; }
    LL_HIP(hipEventSynchronize(e->ev_t1));
    LL_HIP(hipEventElapsedTime(ms, e->ev_t0, e->ev_t1));
    *steps = e->last_steps;
    return LL_OK;
}

}  // extern "C"
