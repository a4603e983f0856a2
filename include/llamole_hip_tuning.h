/* libllamole_hip_tuning -- tuning, timing and test hooks.  NOT part of the drop-in boundary, NOT in the product library.
 *
 * include/llamole_hip.h declares what a maintainer of the reference binds (engines, forward passes, samplers); libllamole_hip.so exports
 * exactly that.  Everything in THIS file -- process-global A/B switches, micro-benchmarks (HIP events around one kernel), probes and
 * single-kernel test entry points used by tests/, tools/ and bench.py's roofline objects -- is compiled only with -DLL_TUNING=1, into
 * libllamole_hip_tuning.so (llamole_amd/build.py builds both from the same sources; LLAMOLE_TUNING=1 makes llamole_amd/_lib.py load the
 * tuning build).  Nothing here changes results beyond documented summation-order effects, and no product code path calls any of it
 * (tests/test_abi_cpu.py checks the exports of the product library and the product package's sources).
 */
#ifndef LLAMOLE_HIP_TUNING_H
#define LLAMOLE_HIP_TUNING_H

#include "llamole_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning utility: average ms of `iters` launches of pipelined-GEMM configuration `cfg` (-1 = the production
 * dispatch) on an [M,K]x[N,K]^T bf16 problem, cycling over `nweights` weight matrices (HBM-resident stream). */
int ll_gemm_bench(int M, int N, int K, int cfg, int splits, int out_f32, int iters, int nweights, float *ms);
/* ll_host_launch_probe : HOST time per enqueued launch (wall time of the issuing loop): kind 0 = empty kernel, 1 = two-argument kernel,
 * 2 = linear_launch onto the <= 64-row panel GEMM, 3 = linear_launch onto the LDS-DMA ring. */
int ll_host_launch_probe(int kind, int n, float *us_per_launch);
/* ll_linear_cfg : ll_linear (bf16 operands) through ONE kernel configuration of the tuning table (gemm.hip: g_pipe_cfgs), so that
 * every variant can be checked against a reference.  splits > 1: C receives `splits` raw f32 slabs (stride M * ldc). */
int ll_linear_cfg(int cfg, const void *A, int lda, const void *W, int ldw, const float *bias, void *C, int ldc, int M, int N, int K,
                  int splits, int epi, int out_f32, void *stream);
/* In-situ kernel timing of the GraphDiT launch chain: ll_dit_class_probe(handle, cls) makes the NEXT ll_dit_run in launch mode record a HIP
 * event pair around every launch of one class of the block's kernels (cls = -1: off), i.e. the kernel is timed where it runs -- behind
 * the launch that feeds it, on the weights of its own layer -- not back to back in a micro-benchmark; ll_dit_class_probe_read returns the sum
 * of the pairs' elapsed times and the number of launches (and synchronises the device).  bench.py's roofline_graphdit comes from it. */
enum { LL_DIT_CLS_QKV = 0, LL_DIT_CLS_ATTN = 1, LL_DIT_CLS_PROJ = 2, LL_DIT_CLS_LNMOD = 3, LL_DIT_CLS_FC1 = 4, LL_DIT_CLS_FC2 = 5, LL_DIT_CLS_COUNT = 6 };
/* Two calibrations of that bracket, OR-ed into `cls` (round 5: bench.py prices roofline_graphdit with what it measures in its own run):
 *   LL_DIT_PROBE_EMPTY : the pair is recorded back to back AT the launch site (before the launch), i.e. it times what an event pair itself
 *                        adds there -- subtracted from the bracketed figure;
 *   LL_DIT_PROBE_SKIP  : no events; the class (fc1 only) is simply NOT launched, so the trajectory's run time against a normal one is the
 *                        class's marginal cost inside the dependent chain (kernel + its launch boundary).  The trajectory's results are
 *                        meaningless in that mode: timing only. */
enum { LL_DIT_PROBE_EMPTY = 0x100, LL_DIT_PROBE_SKIP = 0x200 };
int ll_dit_class_probe(void *handle, int cls);
int ll_dit_class_probe_read(void *handle, float *total_us, int *launches);

/* Test probes of the on-device sampling noise (tests/test_noise_gpu.py):
 * ll_philox_probe : out[i][0..4) = Philox4x32-10 (Salmon et al. 2011; Random123) of counter ctr_key[i][0..4), key ctr_key[i][4..6) through the
 *     device function the samplers call (known-answer vectors).
 * ll_dit_noise_probe : the Exp(1) variates a GraphDiT reverse step s (z_T: s = T) draws on the device for `seed`: qx [B,N,16] atom noise,
 *     qe [B,N,N,5] bond noise (every (i, j); the step uses j > i) -- injecting them through ll_dit_step / ll_dit_init_state must reproduce the
 *     seed's own result bit for bit. */
int ll_philox_probe(const uint32_t *ctr_key, uint32_t *out, int n, void *stream);
int ll_dit_noise_probe(uint64_t seed, int s, int B, int N, float *qx, float *qe, void *stream);

/* Tuning: waves per workgroup of the <= 64-row panel GEMM (4 | 8; default 8); returns the previous value. */
int ll_set_m64_waves(int waves);
/* Tuning: 1 (default) = the <= 64-row panel GEMM reads the GraphDiT engine's MFMA-operand-order weight copies where they exist (one
 * full-line wave instruction per fragment), 0 = the row-major weights; same products and order, bit-identical; returns the previous value.
 * Takes effect at the next graph capture. */
int ll_set_m64_packed(int on);
/* Tuning: 1 (default) = Linears with 65..224 rows take the multi-panel form of the panel GEMM (gemm_m128_kernel: 32 columns per workgroup
 * up to 128 rows, 64 beyond), 2 = up to 256 rows, 0 = the LDS-DMA ring; returns the previous value.  Takes effect at the next launch /
 * graph capture. */
int ll_set_m128_panel(int on);
/* ll_set_gemm_krot : the LDS-DMA GEMM sweeps its k-tiles starting at ((m_tile * (krot & 255) + n_tile * (krot >> 8 or 1)) mod
 * n_ktiles) instead of 0, so that workgroups sharing an operand tile do not miss L2 on the same lines at the same time.  0 = off.
 * Changes the f32 summation order per tile (deterministic).  Returns the previous setting. */
int ll_set_gemm_krot(int krot);

/* Launch-latency probe (tuning utility): average us per kernel over n launches of a trivial kernel
 * (kind 0 empty, 1 load+store, 2 dependent loads, 3 1-MB copy), eager stream (graph=0) or one hipGraph (graph=1). */
int ll_launch_bench(int kind, int n, int graph, float *us);
int ll_launch_bench_set_buffers(void *a, void *b);   /* optional caller-provided 4 MB buffers (NULL = own) */

/* Tuning: ln_mod_res with one wave per 256-column chunk of a row (default) or one wave per row; bit-identical results; returns the
 * previous setting. */
int ll_set_lnmod_multiwave(int on);
/* ll_set_stage_mod : 1 (default) = every step first copies its (B+1) x L x 6H modulation rows to a fixed buffer, so that the 2L
 * AdaLN epilogue launches of the step address them without waiting for the step index in device memory (one memory round trip per
 * launch instead of two); 0 = every launch walks the hoisted table.  Bit-identical.  Takes effect at the next graph capture. */
int ll_set_stage_mod(int on);
/* Tuning: waves per (sequence, head) of the MFMA graph attention (1 | 2 | 4; default 4 = LayerNorm / transpose rows on four waves at
 * head dimension 64, two elsewhere; bit-identical); returns the previous value. */
int ll_set_attn_waves(int waves);

/* ll_debug_check_guards : with LL_DEBUG_POISON=1 in the environment the engines' device buffers carry 4 KB of guard bytes behind their
 * payload; returns how many of the live buffers a kernel has written past the end of (0 = none; always 0 without the variable).
 * Synchronises the device.  tests/conftest.py calls it after every GPU test in that mode. */
int ll_debug_check_guards(void);
/* ll_debug_guard_selftest : 1 = an overrun planted behind a scratch buffer was reported and nothing else was (-1 without LL_DEBUG_POISON). */
int ll_debug_guard_selftest(void);

/* ll_set_topk_single : 1 = one workgroup per row for any out_dim (the round-1 form); 0 (default) = rows longer than 4096 templates
 * are reduced by (out_dim / 4096) x rows workgroups to per-chunk candidates and merged by one workgroup per row -- same result
 * (set, order, ties to the lowest template index).  Returns the previous setting. */
int ll_set_topk_single(int on);

/* ll_rows16_bench : timing utility of ll_linear_rows16_bf16 (HIP events, `nweights` distinct weight matrices); epi | 0x100 = f32 output
 * (the GIN template head).  ll_set_rows16_geometry : bytes of a row per block (128 | 256 | 512), waves per workgroup (4 | 8) and how
 * many consecutive waves split K of one tile; (0, 0, 0) = chosen by tile count.  ll_gemv_fused_bench : the same for ll_gemv_fused_bf16;
 * ll_set_gemv_nt : toggle its non-temporal weight loads (returns the previous setting). */
int ll_gemv_fused_bench(int M, int N, int K, int epi, int norm, int nt, int iters, int nweights, float *ms);
int ll_set_rows16_geometry(int seg, int waves, int ksplit);
int ll_rows16_bench(int M, int N, int K, int epi, int norm, int iters, int nweights, float *ms);
int ll_set_gemv_nt(int on);
/* ll_rows64_bench : timing utility of ll_linear_rows64_bf16 (HIP events, `nweights` distinct packed weight matrices, own workspace;
 * norm & 1: with the output pre-norm, norm & 2: with the input row scale).  ll_set_rows64_ksplit : K slices over workgroups (1..8; 0 = chosen by shape); returns the
 * previous setting. */
int ll_set_rows64_ksplit(int ksg);
int ll_rows64_bench(int M, int N, int K, int epi, int norm, int iters, int nweights, float *ms);
/* Tuning: one-row GEMVs without RMSNorm and K >= 8192 (down_proj) stage x in LDS once per workgroup (default on; bit-identical). */
int ll_set_gemv_stage(int on);

#ifdef __cplusplus
}
#endif
#endif /* LLAMOLE_HIP_TUNING_H */
