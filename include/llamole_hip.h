/*
 * llamole_hip.h -- C ABI of the MI355X (gfx950) implementation of Llamole's graph hot path.
 *
 * The reference (liugangcode/Llamole) has NO native/FFI interface: its seam is the Python
 * class level.  This header is therefore the boundary a maintainer
 * would bind from Python with ctypes; every entry point cites the
 * reference Python interface it replaces.  Conventions:
 *   - plain pointers and sizes only; all data pointers are DEVICE pointers unless named h_*;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls are asynchronous
 *     on that stream unless stated otherwise;
 *   - every function returns 0 on success or a negative LL_E* code; the message is available
 *     through ll_last_error() (thread-local).  No C++ exception crosses the boundary;
 *   - one in-flight call per handle; handles are independent of each other.
 */
#ifndef LLAMOLE_HIP_H
#define LLAMOLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LL_OK 0
#define LL_EINVAL (-1)   /* bad argument / unsupported shape */
#define LL_EHIP (-2)     /* a HIP runtime call failed        */
#define LL_ESTATE (-3)   /* call order violated              */

#define LL_F32 0
#define LL_BF16 1

#define LL_XDIM 16       /* atom classes   (reference diffusion_utils.py:58-59) */
#define LL_EDIM 5        /* bond classes                                          */
#define LL_YDIM 10       /* property slots                                        */
#define LL_TEXT_DIM 768  /* text-embedding width (diffusion_model.py:62)          */

int ll_version(void);
const char *ll_last_error(void);

/* ------------------------------------------------------------------ dense linear (building block)
 * C[M,N] = epi(A[M,K] * W[N,K]^T + bias)   -- torch.nn.Linear layout, used by every Linear on the
 * path (reference layers.py:47,53,106-109; transformer.py:116-130).  dtype selects the operand type
 * of A and W (LL_F32: exact f32 FMA chains; LL_BF16: MFMA 16x16x32 with f32 accumulation).
 * epi: 0 none, 1 GELU(erf), 2 SiLU, 3 Softsign.  out_f32 != 0 writes f32, else the operand dtype.
 * bf16: M <= 4 runs a weight-streaming GEMV, larger M the LDS-DMA pipelined MFMA tiles (A rows are clamped, no
 * padding needed); f32: A must be readable for round_up(M,64) rows. */
int ll_linear(int dtype, const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc,
              int M, int N, int K, int epi, int out_f32, void *stream);

/* bf16 Linear with the K dimension split over `splits` workgroup groups (2..16, K % (64*splits) == 0): for a few rows times a
 * short, wide weight matrix (batch-8/16 decode through down_proj: 3584 x 18944) an unsplit launch has too few workgroups to
 * pull HBM bandwidth.  workspace: device f32 [splits * M * N]; partial slabs are summed in order (deterministic). */
int ll_linear_splitk_bf16(const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, int M, int N,
                          int K, int epi, int splits, float *workspace, void *stream);


/* ------------------------------------------------------------------ GraphDiT sampler
 * Replaces reference GraphDiT.generate / sample_p_zs_given_zt / Transformer.forward
 * (src/model/graph_decoder/diffusion_model.py:252-399, transformer.py:93-187,
 *  diffusion_utils.py:273-349,376-413,476-518). */
/* Any sizes the reference Transformer constructs are valid (transformer.py:24-37): hidden, heads (dividing hidden) and mlp_hidden need
 * not be multiples of anything -- the engine zero-pads its internal copy of the weights to row pitches of 64 and a head pitch of 32 and
 * takes the true widths for every row statistic; the arena layout below is the checkpoint's own.  Bounds (LL_EINVAL beyond them):
 * hidden <= 2048, hidden / heads <= 128, max_nodes <= 128 (65..128 nodes: two wavefronts per row of bond partners, a 128-row attention tile). */
typedef struct LLDitConfig {
    int hidden;        /* H            (config.yaml hidden_size)      */
    int depth;         /* L            (depth)                        */
    int heads;         /* num_heads                                    */
    int mlp_hidden;    /* int(H * mlp_ratio)                           */
    int max_nodes;     /* N            (data.meta.json max_node)       */
    int T;             /* diffusion_steps                              */
    float guide_scale; /* classifier-free guidance scale (1 = off)     */
    int dtype;         /* LL_F32 | LL_BF16: operand type of the Linears */
} LLDitConfig;

/* Host-side tables built from data.meta.json exactly as the reference does
 * (diffusion_model.py:78-103, diffusion_utils.py:172-187). */
typedef struct LLDitTables {
    const float *h_x_marg;     /* [16]     */
    const float *h_e_marg;     /* [5]      */
    const float *h_u_xe;       /* [16*5]   row-normalised S            */
    const float *h_u_ex;       /* [5*16]   row-normalised S^T          */
    const float *h_betas;      /* [T+1]    */
    const float *h_alphas_bar; /* [T+1]    */
} LLDitTables;

/* Weight arena layout: parameters in the order of the reference Transformer.state_dict()
 * (SURVEY.md section 8b).  idx in [0, ll_dit_param_count).  offset/numel in f32 elements. */
int ll_dit_param_count(const LLDitConfig *cfg);
int ll_dit_param_info(const LLDitConfig *cfg, int idx, char *name, int name_cap, int64_t *numel, int64_t *offset);
int64_t ll_dit_arena_elems(const LLDitConfig *cfg);

/* d_weights_f32: device arena laid out as above (f32 masters; converted once to cfg->dtype). Synchronous. */
int ll_dit_create(const LLDitConfig *cfg, const LLDitTables *tables, const float *d_weights_f32, void **handle);
int ll_dit_destroy(void *handle);

/* Start a batch (GraphDiT.generate up to the loop, diffusion_model.py:259-268; conditions.py:19-123):
 * properties [B,10] (NaN = absent), text [B,768] (a NaN in a row drops the row), n_nodes [B] int32.
 * Hoists every step-invariant piece out of the T-step loop: c = c_t + c_y + c_txt for all T steps and
 * all L+1 adaLN modulations (transformer.py:125-130,156-160). */
int ll_dit_begin(void *handle, int B, const float *props, const float *text, const int32_t *n_nodes, void *stream);

/* z_T ~ limit marginals (diffusion_utils.py:495-518).  qx [B*N,16], qe [B*N*N,5] = Exp(1) race noise in the
 * reference's draw order, or NULL/NULL to draw on device from Philox4x32-10(seed). */
int ll_dit_init_state(void *handle, const float *qx, const float *qe, uint64_t seed, void *stream);

/* One reverse step z_{s+1} -> z_s (sample_p_zs_given_zt, diffusion_model.py:309-399), s in [0,T). */
int ll_dit_step(void *handle, int s, const float *qx, const float *qe, uint64_t seed, void *stream);

/* Whole trajectory T-1 .. 0 with on-device noise.  use_graph: 1 = the step is captured once as a hipGraph (five steps per graph when
 * that divides T) and replayed -- the host call returns at once; 0 = every kernel launched from this call's loop (no Python in it) --
 * measured 4-14 % faster per step than the replay when the trajectory has the GPU to itself (consecutive kernel nodes of a replayed
 * graph start ~1.8 us apart at best, consecutive queued launches ~1.3 us), but the host is busy for most of the trajectory's duration
 * (~4 us per launch); 2 = the library's choice: launches when the engine is not in overlap mode, the replay when it is (the host then
 * has the next prompt's LLM decode to feed); the environment variable LL_DIT_RUN_MODE = graph | launches overrides that choice. */
enum { LL_DIT_RUN_LAUNCHES = 0, LL_DIT_RUN_GRAPH = 1, LL_DIT_RUN_AUTO = 2 };
int ll_dit_run(void *handle, uint64_t seed, int use_graph, void *stream);

/* State I/O: X int8 [B,N] (-1 = masked), E int8 [B,N,N] (-1 = all-zero one-hot: masked pair or z_T diagonal). */
int ll_dit_set_state(void *handle, const int8_t *X, const int8_t *E, void *stream);
int ll_dit_get_state(void *handle, int8_t *X, int8_t *E, void *stream);

/* Parity taps (tests only): run the denoiser on the current state at step s.
 * logX [2,B,N,16], logE [2,B,N,N,5] (pass 0 = conditional, 1 = unconditional), masked like
 * Transformer.forward's PlaceHolder.mask; hidden (nullable) [2,B,N,H] = residual stream after `tap_layer`
 * blocks (0 = after x_embedder). */
int ll_dit_denoise(void *handle, int s, float *logX, float *logE, float *hidden, int tap_layer, void *stream);
/* Guided probabilities of step s as handed to sample_discrete_features: pX [B,N,16], pE [B,N,N,5]
 * (only rows of valid nodes / pairs i<j of valid nodes are defined; others are written as 0). */
/* ll_dit_denoise_rows : the denoiser's conditional / unconditional logits for the current state (ll_dit_set_state) with a
 * PER-GRAPH timestep t_int[b] in 0..T (device int32 [B]) instead of one shared reverse step: the training forward
 * (GraphDiT.forward / apply_noise, diffusion_model.py:148-250) draws t independently per graph, t = 0 included.
 * logX [2][B][N][16], logE [2][B][N][N][5] as ll_dit_denoise. */
int ll_dit_denoise_rows(void *handle, const int32_t *t_int, float *logX, float *logE, void *stream);
int ll_dit_step_probs(void *handle, int s, float *pX, float *pE, void *stream);
/* Conditioning vectors c [B+1,H] at step s (rows 0..B-1 conditional, row B unconditional). */
int ll_dit_cvec(void *handle, int s, float *c, void *stream);

/* Timing of the most recent ll_dit_run measured with HIP events on `stream`: total ms and steps. */
/* ll_dit_set_overlap : on = the next ll_dit_run trajectories run NEXT TO another stream's kernels (GraphDiT.generate_graphs_async
 *     under the LLM decode of the next prompt): their <= 64-row panel GEMMs take the 48 KB LDS-DMA ring instead of the panel kernel,
 *     whose workgroups need a whole CU's LDS (slower alone, faster for the pair); a separate captured step is kept per mode. */
int ll_dit_set_overlap(void *handle, int on);
/* ll_dit_set_option : per-engine switches, effective for every later denoiser call of this handle (step / run / denoise /
 * step_probs).  LL_DIT_OPT_OVERLAP = ll_dit_set_overlap; LL_DIT_OPT_GENERIC_ATTN = run the f32-LDS attention kernel under the
 * bf16 engine instead of the MFMA one (parity hook: the two are compared on identical q|k|v by the tests);
 * LL_DIT_OPT_FUSED_QKV_ATTN = the block's q|k|v projection and attention as ONE launch per (sequence, head) (bf16, head
 * dimension 64, hidden 256 | 512 | a multiple of 1024): -1 = above 128 token rows when the launch has 64..512 such workgroups, i.e. batch 3..16
 * at 16 heads and 32 nodes -- two sequences per workgroup from batch 11 to 24 -- and always in overlap mode (default), 0 = never,
 * 1 = whenever eligible, 2 = two sequences per workgroup whenever eligible (graphs of <= 32 nodes).
 * The launch chain is the one GraphDiT step this library has: the per-XCD persistent trajectory kernel, the per-XCD projection + AdaLN
 * launch and the packed-weight panel MLP GEMM of rounds 2-4 were measured slower at every batch and have been removed (HISTORY.md). */
enum { LL_DIT_OPT_OVERLAP = 0, LL_DIT_OPT_GENERIC_ATTN = 1, LL_DIT_OPT_FUSED_QKV_ATTN = 2 };
int ll_dit_set_option(void *handle, int option, int value);
int ll_dit_last_run_ms(void *handle, float *ms, int *steps);
/* ------------------------------------------------------------------ GIN encoder / predictor
 * Replaces GNNEncoder.forward + ProjectionHead (src/model/graph_encoder/model.py:124-205) and
 * GNNRetrosynthsizer.forward (src/model/graph_predictor/model.py:306-353). */
/* hidden (<= 2048) and text_dim need not be multiples of 64 either: zero-padded inside the engine (graph_encoder/model.py:87-112,
 * graph_predictor/model.py:231-278 take them from the checkpoint's config); out_dim is any positive number of templates. */
typedef struct LLGinConfig {
    int num_layer;
    int hidden;     /* H_gin                                   */
    int kind;       /* 0 = encoder (+projection head), 1 = predictor (+decoder) */
    int out_dim;    /* predictor: #templates (num_task); encoder: ignored          */
    int text_dim;   /* predictor: 768                                              */
    int dtype;      /* LL_F32 | LL_BF16                                            */
} LLGinConfig;

int ll_gin_param_count(const LLGinConfig *cfg);
int ll_gin_param_info(const LLGinConfig *cfg, int idx, char *name, int name_cap, int64_t *numel, int64_t *offset);
int64_t ll_gin_arena_elems(const LLGinConfig *cfg);
int ll_gin_create(const LLGinConfig *cfg, const float *d_weights_f32, void **handle);
int ll_gin_destroy(void *handle);

/* x [n] atom ids in [0,118).  Edges in CSR-by-destination form: rowptr [n+1], and for e in
 * [rowptr[v], rowptr[v+1]) the message source src[e] and bond class attr[e] in 0..4 (the PyG
 * edge_index[0] / edge_attr of the edges whose edge_index[1] == v, in stable order).
 * batch [n] sorted graph ids, gptr [G+1] node offsets of each graph.
 * c [G,text_dim] or NULL (predictor: NULL -> text_dropping row, model.py:315-316).
 * out: encoder [G,H] L2-normalised embedding; predictor [G,out_dim] template logits.
 * pooled (nullable) [G,H]: add-pooled node states before the head. */
/* ll_graph_csr : the reference's graph batch (int64 x [n], edge_index [2,E], edge_attr [E], sorted batch [n]; modeling_llamole.py:
 * 328-333, 611-616, 829-834) -> the int32 arrays of ll_gin_forward in one launch: x32 [n], rowptr [n+1] / src [E] / attr [E] =
 * CSR by destination with every node's in-edges in edge-list order (the reference's summation order), batch32 [n], gptr [G+1].
 * scratch: n int32.  *err (device memory, or pinned host memory so that the caller's stream never carries a read-back) = 0, 1 (batch
 * unsorted / graph id out of range), 2 (edge endpoint out of range) or 3 (atom type outside [0,118) / bond type outside [0,5)).  Whatever
 * the flag says the output arrays are in bounds for ll_gin_forward (ids clamped, bad edges dropped), so the caller may check it late. */
int ll_graph_csr(const int64_t *x, const int64_t *edge_index, const int64_t *edge_attr, const int64_t *batch, int n_nodes, int n_edges,
                 int n_graphs, int32_t *x32, int32_t *rowptr, int32_t *src, int32_t *attr, int32_t *batch32, int32_t *gptr,
                 int32_t *scratch, int32_t *err, void *stream);
int ll_gin_forward(void *handle, const int32_t *x, const int32_t *rowptr, const int32_t *src, const int32_t *attr,
                   const int32_t *batch, const int32_t *gptr, int n_nodes, int n_edges, int n_graphs, const float *c,
                   float *out, float *pooled, void *stream);

/* Training path of the predictor (reference modeling_llamole.py:385-419: the retro cross-entropy
 * reaches the LLM through c = lm_to_graph_predictor(hidden); the predictor's own weights are frozen):
 * ll_gin_forward_train = ll_gin_forward (predictor, c != NULL) that also keeps the per-layer activations;
 * ll_gin_backward_c    = d loss / d c [G,text_dim] from d loss / d logits [G,out_dim] for the batch of the last
 *                        ll_gin_forward_train call (LL_ESTATE otherwise).  Edges here in CSR-by-SOURCE form: rowptr_src [n+1],
 *                        and for e in [rowptr_src[v], rowptr_src[v+1]) the message destination dst[e] and bond class attr[e].
 * Replaces torch.autograd through GNNRetrosynthsizer.forward (graph_predictor/model.py:306-353). */
int ll_gin_forward_train(void *handle, const int32_t *x, const int32_t *rowptr, const int32_t *src, const int32_t *attr,
                         const int32_t *batch, const int32_t *gptr, int n_nodes, int n_edges, int n_graphs, const float *c,
                         float *out, void *stream);
int ll_gin_backward_c(void *handle, const int32_t *rowptr_src, const int32_t *dst, const int32_t *attr, const int32_t *batch,
                      const int32_t *gptr, int n_nodes, int n_edges, int n_graphs, const float *c, const float *dlogits, float *dc,
                      void *stream);

/* softmax over out_dim then top-k (GraphPredictor.sample_templates, graph_predictor/model.py:174-179).
 * probs [rows,k] descending, idx [rows,k]. k <= 64. */
int ll_softmax_topk(const float *logits, int rows, int out_dim, int k, float *probs, int32_t *idx, void *stream);

/* CostMLP (graph_predictor/model.py:356-391): softplus(W3 relu(W0 fp + b0) + b3); fps [n,2048] f32 0/1.
 * weights: device f32 arena = [layers.0.weight 128x2048 | layers.0.bias 128 | layers.3.weight 1x128 | layers.3.bias 1]. */
int ll_cost_mlp(const float *weights, const float *fps, int n, float *out, void *stream);

/* ------------------------------------------------------------------ fused LLM decode helpers (optional, section 8 f2)
 * Same arithmetic as the HuggingFace modules they stand in for, including the intermediate bf16 roundings of PyTorch's
 * op-by-op evaluation (results are bit-identical); all tensors bf16.
 * ll_rmsnorm_bf16 : Qwen2RMSNorm/LlamaRMSNorm.forward on [rows,H].
 * ll_rope_bf16    : apply_rotary_pos_emb on q [B,nh,S,D] / k [B,nkv,S,D] given by element strides of dims 0..2 (last dim
 *                   contiguous) and cos/sin [B or 1,S,D] strides of dims 0..1; outputs contiguous.
 * ll_silu_mul_bf16: silu(gate) * up on [rows, cols] views with input row stride ld_in (elements), contiguous output. */
int ll_rmsnorm_bf16(const void *x, const void *w, void *out, int rows, int H, float eps, void *stream);
int ll_rope_bf16(const void *q, const void *k, const void *cos, const void *sin, void *qo, void *ko, int B, int nh, int nkv,
                 int S, int D, const int64_t *qstr, const int64_t *kstr, const int64_t *cstr, void *stream);
int ll_silu_mul_bf16(const void *gate, const void *up, void *out, int rows, int cols, int64_t ld_in, void *stream);
/* ll_kv_append_bf16 : StaticCache layer update at decode: write k_new/v_new [B,nkv,S,D] (element strides of dims 0..2) into
 *                     keys/values [B,nkv,maxlen,D] at positions *pos .. *pos+S-1 (pos: device int64).
 * ll_decode_attn_bf16: softmax(q K^T * scale + mask) V over the static cache with grouped-query heads; q [B,nh,S,D] strided,
 *                     mask bool [B,1,S,maxlen] (strides of dims 0 and 2), out [B,S,nh,D] contiguous; D in {64,128}. */
int ll_kv_append_bf16(void *K, void *V, const void *k_new, const void *v_new, const int64_t *pos, int B, int nkv, int S,
                      int maxlen, int D, const int64_t *kstr, const int64_t *vstr, void *stream);
int ll_decode_attn_bf16(const void *q, const void *K, const void *V, const void *mask, void *out, int B, int nh, int nkv, int S,
                        int maxlen, int D, float scale, const int64_t *qstr, const int64_t *mstr, void *stream);

/* ---- decode-step layer fusion (one decoder layer of the HF LLM at batch <= 4 = five launches) ----------------------------
 * ll_gemv_fused_bf16 : out[M,N] = epilogue( rmsnorm?(x)[M,K] . W^T + bias ), M in 1..4, bf16 operands, f32 accumulation.
 *     norm_w != NULL : x is first normalised like Qwen2RMSNorm/LlamaRMSNorm (K <= 8192, M*K <= 32768), eps as given;
 *     epi LL_GEMV_PLAIN    : bf16(acc + bias)                                 (nn.Linear; fused q/k/v; lm_head)
 *         LL_GEMV_RESIDUAL : bf16(residual + bf16(acc + bias))                (o_proj / down_proj + the layer's residual add)
 *         LL_GEMV_SILU_MUL : W has 2N rows (gate rows [0,N), up rows [N,2N)): bf16(bf16(silu(gate)) * up)   (gated MLP)
 *     Replaces <Model>RMSNorm.forward + nn.Linear.forward + the residual add / act_fn(gate)*up of
 *     transformers modeling_qwen2.py Qwen2DecoderLayer.forward / Qwen2MLP.forward (bit-identical arithmetic).
 * ll_decode_attn_rope_bf16 : apply_rotary_pos_emb + StaticLayer.update + attention for ONE new position per sequence.
 *     qkv [B, (nh+2*nkv)*D] rows (stride ld_qkv) = fused q|k|v projection; cos/sin [B or 1, D] (batch stride cs_stride, 0 to
 *     broadcast); Kc/Vc [B,nkv,maxlen,D] static cache, appended at *pos (device int64); mask bool [B,maxlen] rows
 *     (stride mask_stride) for the new query (causal decode: cache slots behind *pos hold no keys yet and must be masked -- with more than 16
 *     sequences, where one workgroup serves the whole KV group, they are not even fetched); out [B, nh*D].  D in {64,128}.
 * (timing / tuning hooks of these kernels: include/llamole_hip_tuning.h) */
#define LL_GEMV_PLAIN 0
#define LL_GEMV_RESIDUAL 1
#define LL_GEMV_SILU_MUL 2
int ll_gemv_fused_bf16(const void *x, int ldx, const void *W, int ldw, const float *bias, const void *norm_w, float eps,
                       const void *residual, int ldr, void *out, int ldc, int M, int N, int K, int epi, void *stream);
int ll_decode_attn_rope_bf16(const void *qkv, int64_t ld_qkv, const void *cos, const void *sin, int64_t cs_stride, void *Kc,
                             void *Vc, const int64_t *pos, const void *mask, int64_t mask_stride, void *out, int B, int nh,
                             int nkv, int maxlen, int D, float scale, void *stream);
/* ll_decode_prologue : per-token prologue of a decode step in one launch: cos/sin [B,D] bf16 of Qwen2RotaryEmbedding.forward
 *     (transformers modeling_qwen2.py: inv_freq * position in f32, cos/sin, * attention_scaling, cast) for position_ids [B],
 *     and the boolean key mask [B,maxlen] of create_causal_mask (masking_utils.py) for one new query at cache slot *pos:
 *     key j visible iff j <= *pos and mask2d[b][j] != 0 (mask2d int64, row stride mask_stride). */
int ll_decode_prologue(const int64_t *position_ids, const float *inv_freq, float attention_scaling, const int64_t *mask2d,
                       int64_t mask_stride, const int64_t *pos, void *cos, void *sin, void *mask_out, int B, int D, int maxlen,
                       void *stream);
/* ll_suffix_prologue / ll_suffix_attn_rope_bf16 : the same pair for S consecutive new positions per sequence -- the reference's query-token
 *     re-forward (modeling_llamole.py:641-646: <design_start> + the body tokens) run on top of the decode's KV cache at slots
 *     *pos .. *pos+S-1.  Rows r = b*S + s.  Prologue: cos/sin [B*S,D] for position_ids [B*S], key mask [B*S,maxlen]: key j visible to row
 *     (b,s) iff j <= *pos + s and mask2d[b][j] != 0.  Attention: qkv [B*S, (nh+2*nkv)*D] (row stride ld_qkv); rotary on q and k, the
 *     new keys / values stored to Kc/Vc [B,nkv,maxlen,D], softmax(q K^T * scale + mask) V over the cache -> out [B*S, nh*D]; bit for
 *     bit ll_rope_bf16 + ll_kv_append_bf16 + ll_decode_attn_bf16.  S <= 16, D in {64,128}. */
int ll_suffix_prologue(const int64_t *position_ids, const float *inv_freq, float attention_scaling, const int64_t *mask2d,
                       int64_t mask_stride, const int64_t *pos, void *cos, void *sin, void *mask_out, int B, int S, int D, int maxlen,
                       void *stream);
int ll_suffix_attn_rope_bf16(const void *qkv, int64_t ld_qkv, const void *cos, const void *sin, void *Kc, void *Vc, const int64_t *pos,
                             const void *mask, void *out, int B, int S, int nh, int nkv, int maxlen, int D, float scale, void *stream);

/* ll_linear_rows16_bf16 : out[M,N] = epilogue(x[M,K] . W^T + bias), M in 1..16, K % 32 == 0, bf16 operands, f32 accumulation on MFMA:
 *     the nn.Linear of a batched decode step (5..16 sequences: lock-step A* searches, several prompts per GPU) as a weight
 *     stream -- every wave streams its own 16 weight rows in line-contiguous segments through a private LDS image, no barrier in
 *     the main loop.  Arguments as ll_gemv_fused_bf16: norm_w != NULL = RMSNorm prologue (x is multiplied by the norm weight while it
 *     is staged and rsqrt(mean(x^2) + eps) of the row scales the accumulator: Qwen2RMSNorm + nn.Linear up to the place of one bf16
 *     rounding); epilogues LL_GEMV_PLAIN / LL_GEMV_RESIDUAL / LL_GEMV_SILU_MUL (W has 2N rows for SILU_MUL); replaces nn.Linear.forward (+ the residual add / act_fn(gate)*up of Qwen2DecoderLayer / Qwen2MLP.forward,
 *     transformers modeling_qwen2.py) under the reference's language_model.generate (modeling_llamole.py:599, :849).
 * (geometry tuning and the timing utility: include/llamole_hip_tuning.h) */
int ll_linear_rows16_bf16(const void *x, int ldx, const void *W, int ldw, const float *bias, const void *norm_w, float eps,
                          const void *residual, int ldr, void *out, int ldc, int M, int N, int K, int epi, void *stream);

/* ll_linear_rows64_bf16 : the same Linear for up to 64 token rows (17..64 sequences per GPU decoded together: the reference hands
 *     language_model.generate whatever per_device_eval_batch_size the DataLoader yields, eval/workflow.py:89-91,110-124,
 *     modeling_llamole.py:599-603; BASELINE configs[3] is 64 prompts).  Wp is a copy of the [N, K] weight ([2N, K] gate rows then up rows
 *     for LL_GEMV_SILU_MUL, N % 16 == 0) in MFMA operand order made ONCE by ll_rows64_pack_bf16 into ll_rows64_packed_elems(rows, K)
 *     bf16 elements (rows padded to a multiple of 16): weight fragments go from HBM straight into operand registers, x through an LDS
 *     ring; a persistent workgroup owns 64 weight rows x all token rows at a time, so weights are read once per launch and x once per
 *     workgroup.  K % 32 == 0.  Epilogues as ll_gemv_fused_bf16.  Matrices with few row groups (o_proj, down_proj) also split K over
 *     workgroups when `workspace` holds ll_linear_rows64_workspace_bytes(M, N) bytes of device scratch (f32 partial slabs, summed in
 *     slice order by a second launch on the same stream: deterministic); workspace = NULL keeps one launch.
 *     The RMSNorm between two Linears of a decoder layer (Qwen2RMSNorm / LlamaRMSNorm, transformers modeling_qwen2.py) is split over
 *     the two calls, so that no normalisation arithmetic sits in a GEMM's main loop:
 *       producer (plain / residual epilogue, needs the workspace): next_norm_w != NULL -> the second launch also writes
 *           scaled_out[M, N] (row stride ldn) = bf16(out * next_norm_w) and ssq_out[M, ll_rows64_ssq_chunks(N)] = per-1024-column sums
 *           of out^2;
 *       consumer: row_ssq != NULL -> x is such a scaled_out and the accumulator of token row m is multiplied by
 *           rsqrt(sum(row_ssq[m, 0..ssq_chunks)) / K + eps) before bias / epilogue (the ll_linear_rows16_bf16 form of the prologue).
 *     ll_rows64_prenorm_bf16 : scaled_out / ssq_out of a given x[M, N] alone (the embedding rows ahead of the first layer). */
int64_t ll_rows64_packed_elems(int N, int K);
int ll_rows64_pack_bf16(const void *W, int ldw, int N, int K, void *packed, void *stream);
int ll_linear_rows64_bf16(const void *x, int ldx, const void *Wp, const float *bias, const void *residual, int ldr, void *out, int ldc, int M,
                          int N, int K, int epi, const float *row_ssq, int ssq_chunks, float eps, const void *next_norm_w, void *scaled_out,
                          int ldn, float *ssq_out, void *workspace, int64_t workspace_bytes, void *stream);
int64_t ll_linear_rows64_workspace_bytes(int M, int N);
int ll_rows64_ssq_chunks(int N);
int ll_rows64_prenorm_bf16(const void *x, int ldx, const void *norm_w, void *scaled_out, int ldn, float *ssq_out, int M, int N, void *stream);

/* ll_sample_token_bf16 : one decode-loop sampling step per row in ONE launch -- HF TemperatureLogitsWarper + TopPLogitsWarper
 *     + softmax + multinomial (transformers generation/logits_process.py, generation/utils.py _sample; the reference reaches
 *     them through language_model.generate(do_sample, temperature, top_p), modeling_llamole.py:599/:849) and the loop's
 *     bookkeeping.  logits bf16 [B,V] (row stride ld, V % 8 == 0, V <= 163840); inv_temp = 1/temperature; greedy != 0
 *     = argmax (lowest index on ties).  A value is kept iff the probability mass strictly above it is < top_p; ties with
 *     the boundary value are all kept.  Randomness: Philox4x32-10 keyed by *seed (device int64), counter (step[b], b).
 *     Per row b: t = step[b]; token = done[b] ? pad : sampled; out_tokens[b*ld_out + t] = tok[b] = token;
 *     done[b] |= token in eos[0..n_eos); step[b] = t+1; if advance: posid[b] += 1, pos[0] += 1 (either may be NULL).
 *     dbg (optional, [B,4] uint64): Z, kept mass (2^-40 fixed point), boundary key, sampled key -- parity taps. */
int ll_sample_token_bf16(const void *logits, int64_t ld, int B, int V, float inv_temp, float top_p, int greedy,
                         const int64_t *seed, const int64_t *eos, int n_eos, int64_t pad, void *done, int64_t *tok,
                         int64_t *out_tokens, int64_t ld_out, int max_new, int64_t *step, int64_t *posid, int64_t *pos,
                         int advance, uint64_t *dbg, void *stream);
/* ... with HF's TopKLogitsWarper ahead of the nucleus (the reference samples with GeneratingArguments' default top_k = 50,
 * src/hparams/generating_args.py:39-42): only the top_k highest logits survive (ties with the k-th are kept, as HF's
 * `logits < kth` test does) and top_p is applied to their renormalised distribution.  top_k <= 0 = off. */
int ll_sample_token_topk_bf16(const void *logits, int64_t ld, int B, int V, float inv_temp, float top_p, int top_k, int greedy,
                              const int64_t *seed, const int64_t *eos, int n_eos, int64_t pad, void *done, int64_t *tok,
                              int64_t *out_tokens, int64_t ld_out, int max_new, int64_t *step, int64_t *posid, int64_t *pos,
                              int advance, uint64_t *dbg, void *stream);
/* ll_sample_token_topk_ws_bf16 : the same with a workspace (ll_sample_workspace_bytes(B) bytes, 16-byte aligned, zero-filled ONCE by the
 *     caller, one per stream in flight).  With B <= 4 rows, 1 <= top_k <= 128, sampling and no dbg tap the work is split: a first launch of V/2048
 *     workgroups per row hands on the tokens that can be among the row's k largest, one workgroup per row then finishes on those -- the
 *     same token as without a workspace for every seed (the one-workgroup path also takes over, on the device, when a row has more than
 *     16384 candidates).  workspace NULL: ll_sample_token_topk_bf16. */
int64_t ll_sample_workspace_bytes(int B);
int ll_sample_token_topk_ws_bf16(const void *logits, int64_t ld, int B, int V, float inv_temp, float top_p, int top_k, int greedy,
                                 const int64_t *seed, const int64_t *eos, int n_eos, int64_t pad, void *done, int64_t *tok,
                                 int64_t *out_tokens, int64_t ld_out, int max_new, int64_t *step, int64_t *posid, int64_t *pos, int advance,
                                 uint64_t *dbg, void *workspace, int64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LLAMOLE_HIP_H */
