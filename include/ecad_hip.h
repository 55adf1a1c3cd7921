/*
 * ecad_hip.h -- C ABI of libecad_hip.so: the MI355X (gfx950) native engine for the
 * denoising hot path of ECAD's PixArt cached-diffusion transformer.
 *
 * Plain pointers and sizes only; no torch / HIP C++ types.  Streams cross the boundary as
 * `void*` (a hipStream_t; NULL = the null stream).  Every "dev" pointer is device memory
 * owned by the caller (e.g. a torch tensor's data_ptr) that must outlive the stream work.
 *
 * Each entry point names the reference interface it replaces (paths under /root/reference).
 * All functions return 0 on success or a negative ECAD_E_* code; the message is available
 * from ecad_last_error().  One engine per (process, GPU); not thread-safe by contract.
 */
#ifndef ECAD_HIP_H
#define ECAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECAD_ABI_VERSION 7

enum {
  ECAD_OK = 0,
  ECAD_E_INVALID = -1,   /* bad argument / shape / state   (Python shim: ValueError)   */
  ECAD_E_HIP = -2,       /* HIP runtime failure             (Python shim: RuntimeError) */
  ECAD_E_NOT_READY = -3, /* weights / schedule / batch not set up                       */
  ECAD_E_UNSUPPORTED = -4
};

enum { ECAD_DTYPE_F16 = 0, ECAD_DTYPE_BF16 = 1, ECAD_DTYPE_F32 = 2 };

/* Model geometry: the constructor arguments of PixArtTransformer2DEdited
 * (ecad/transformer_2d_models/pixart_transformer_2d_edited.py:23-49). */
typedef struct EcadModelCfg {
  int32_t num_layers;        /* 28 */
  int32_t num_heads;         /* 16 */
  int32_t head_dim;          /* 72 */
  int32_t in_channels;       /* 4  */
  int32_t out_channels;      /* 8  */
  int32_t patch_size;        /* 2  */
  int32_t sample_size;       /* latent side the checkpoint was trained at: 32 for 256x256 */
  int32_t caption_channels;  /* 4096 */
  int32_t max_batch;         /* largest transformer batch Bt (CFG-doubled) to size workspaces for */
  int32_t max_tokens;        /* largest image-token count N = (h/p)*(w/p) */
  int32_t max_text_tokens;   /* largest T (120 alpha, 300 sigma) */
  int32_t dtype;             /* ECAD_DTYPE_F16 (reference: .half(), :98) or ECAD_DTYPE_BF16 */
  float norm_eps;            /* 1e-6 */
  float interpolation_scale; /* <= 0: max(sample_size / 64, 1) as diffusers does */
  int32_t use_additional_conditions; /* 1 for the 1024-MS alpha checkpoints: resolution / aspect-ratio embedders (pixart_transformer_2d_edited.py:43,69) */
} EcadModelCfg;

typedef struct EcadStats {
  int64_t components_executed;   /* of steps*blocks*3, after the "empty cache => recompute" rule */
  int64_t components_cached;
  int64_t linear_macs_reference; /* Linear/Conv MACs the reference would execute (compute_macs.py model) */
  int64_t linear_macs_executed;  /* after hoisting step-invariant work out of the step loop */
  int64_t attn_core_macs;        /* QK^T + PV */
  int64_t cache_bytes_read;      /* per image batch, algorithmic */
  int64_t cache_bytes_written;
  int64_t cache_arena_bytes;
  int64_t weight_bytes;
  int64_t workspace_bytes;
  int32_t graph_nodes;           /* kernels captured in the whole-schedule hipGraph (0 = not built) */
  int32_t launches_per_batch;
  int32_t plan_builds;           /* schedule -> op-list lowerings since ecad_create (a (schedule, shapes) seen recently is served from the plan-set cache) */
  int32_t graph_captures;        /* hipGraph captures + instantiations since ecad_create */
  int32_t dynamic_step_plans;    /* ecad_step calls that ran an ad-hoc plan because the usable caches differed from the static resolution (reset_cache mid-run) */
  int32_t plan_sets_cached;      /* plan sets (with their graphs) parked in the LRU right now */
  int32_t ln_folded;             /* LayerNorm + modulate launches of the whole-loop plan folded into their consumer GEMMs */
  int32_t fold_preps;            /* times the pre-scaled weights of a plan set were (re)built: once per (schedule, timestep grid) */
  int64_t fold_bytes;            /* bytes of pre-scaled weights + c1 / c2 vectors the live plan set uses */
  int64_t fold_arena_bytes;      /* device memory this engine holds for such copies right now: live + parked plan sets + recycling pool (bounded by ECAD_LN_FOLD_BUDGET_MB, default 24576) */
  int32_t fold_refused;          /* plan sets lowered WITHOUT the fold because their copies did not fit that budget / the device */
  int32_t replays_folded;        /* cached components of the whole-loop plan replayed in the epilogue of the read-modify-write GEMM in front of them (no pending list, no launch) */
  /* ABI 6: read-modify-write GEMM launches (both out-projections, FF2) issued through the host path since ecad_create, by the kernel that took them -- a captured
   * hipGraph counts its launches once, when it is captured; replays issue nothing through the host.  Which kernel a launch gets depends on the batches in flight,
   * the tile count and the replayed components it carries: these say what actually ran. */
  int64_t rmw_launches_wide;     /* gemm2w4.h, 256 x 288 tiles (whole rounds, or half-chip beside another batch) */
  int64_t rmw_launches_narrow4;  /* gemm3s4.h, 256 x 144 tiles on four waves */
  int64_t rmw_launches_gemm3s;   /* gemm3s.h, 256 x 144 tiles on eight waves (launches that took the few-rows forms -- latency tiles, split-K -- are in none of the three) */
  int32_t split_k;               /* ecad_set_split_k / ECAD_SPLIT_K as it stands */
  int32_t batches_in_flight;     /* ecad_set_batches_in_flight as it stands */
} EcadStats;

typedef struct EcadEngine EcadEngine;

/* ---- lifetime ------------------------------------------------------------------------- */

/* Replaces PixArtTransformer2DEdited.__init__/_post_init (pixart_transformer_2d_edited.py:23-102). */
int ecad_create(const EcadModelCfg* cfg, int device, EcadEngine** out);
void ecad_destroy(EcadEngine* e);
const char* ecad_last_error(const EcadEngine* e); /* e may be NULL: last create() error */
int ecad_abi_version(void);

/* Replaces from_pretrained + init_cached_transformer_blocks weight copy (:104-153).  `name` is the
 * diffusers state-dict key ("transformer_blocks.3.attn1.to_q.weight", "pos_embed.proj.bias", ...).
 * `ptr` may be host or device memory; the engine converts to its dtype and re-lays-out (fused QKV,
 * padded proj_out); the caller keeps ownership. */
int ecad_load_weights(EcadEngine* e, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim);
/* Call once after the last ecad_load_weights: checks completeness, builds fused layouts. */
int ecad_finalize_weights(EcadEngine* e);
/* A further execution context on the same GPU (round 3): `e` -- created with the same model configuration and dtype as `donor`, whose weights are
 * finalized -- drops its own weight arena and reads the donor's from now on; it keeps its own workspaces, cache arena, plans and hipGraph.  Image batches
 * submitted to different contexts on different HIP streams overlap on the device (one batch's kernel boundaries and HBM-bound kernels hide behind the other's
 * GEMMs), which is how a caller keeps several batches in flight -- the reference has one pipeline object per process and runs batches back to back
 * (image_generator.py:161-186, `generate_from_saved_prompts`).  The donor must outlive `e`.  ECAD_E_INVALID for mismatched configurations,
 * fp8 mode travels with the weights (ABI 6): a donor whose weights were finalized in fp8 mode hands its e4m3 copies on as well (they live with the arena), and
 * ecad_set_fp8 stays refused on the context itself (ECAD_E_UNSUPPORTED).  Thread safety: the per-batch calls of different contexts may run on different host
 * threads; ecad_create / ecad_share_weights / ecad_load_weights / ecad_finalize_weights / ecad_destroy of engines that share an arena must be serialized by the
 * caller (the arena's reference count and weight generation are plain fields). */
int ecad_share_weights(EcadEngine* e, EcadEngine* donor);
/* LayerNorm + modulate folded into the consumer GEMMs of the whole-loop plan (default on, env ECAD_LN_FOLD): the per-(step, block) pre-scaled weights are
 * built on the first ecad_run_schedule of a (schedule, timestep grid) -- a few milliseconds that a caller running each schedule for one batch only (the genetic
 * search's candidate evaluation under ecad/genetic) does not get back.  Results with and without the fold agree to the rounding that moved (3e-3 on final latents). */
int ecad_set_ln_fold(EcadEngine* e, int on);
/* How many image batches the caller keeps in flight on this GPU (this context and the ones sharing its weights; default 1).  With few rows per launch (batch 1-8) the
 * engine runs its GEMMs on 64-row latency tiles while the launch's 256-row tiles would cover at most half the chip; the tiles of the other batches in
 * flight count towards that half, so that a caller that fills the GPU with several small batches keeps the throughput kernels.  From two batches in flight on, the
 * read-modify-write GEMMs whose wide tiles cover half the chip (batch 16: both out-projections and FF2) go out as half-chip launches that run beside another
 * stream's kernels: faster per launch pair, slower than the full-chip kernel when nothing else runs -- say 1 when the batches are not really
 * concurrent.  Results do not depend on it (every tile shape is the same arithmetic, bit for bit).  ECAD_E_INVALID outside 1..64. */
int ecad_set_batches_in_flight(EcadEngine* e, int n);
/* Split-K for the few-rows FF2 (default on, env ECAD_SPLIT_K sets the default): batch 1-3 run FF2 as K-slices + a reduce launch (-14 % per image at
 * batch 1); the fp32 sum is then taken in another order than in the sequential-K kernels, so a sample run alone at batch 1-3 equals its row inside a larger batch
 * to rounding (1.6e-3 on final latents), not bit for bit.  0 restores bit-identity across batch sizes (golden files, cached references, a genetic evaluation whose
 * last prompt batch has fewer than four samples).  Plan sets and graphs are keyed by it. */
int ecad_set_split_k(EcadEngine* e, int on);
/* Opt-in OCP fp8 (e4m3) for the block GEMMs (QKV, attention out-projections, cross-attention Q, FF1, FF2): e4m3 weight copies with
 * per-output-channel scales at finalize, activations quantised per row in front of each GEMM (csrc/gemm3s8.h).  Call before
 * ecad_finalize_weights.  Default off (env ECAD_FP8=1 sets the default): the reference computes in fp16, which stays the precision of
 * the headline metric; DESIGN.md states the measured accuracy of this mode. */
int ecad_set_fp8(EcadEngine* e, int on);
/* Replaces the `added_cond_kwargs={"resolution", "aspect_ratio"}` argument of PixArtTransformer2DEdited.forward
 * (pixart_transformer_2d_edited.py:169,208-211; filled by the pipeline at pass_through.py:268-290 for sample_size 128 checkpoints).
 * Only for engines created with use_additional_conditions = 1, where it must precede ecad_begin_image_batch / ecad_run_schedule for a
 * batch of that size (ECAD_E_NOT_READY otherwise -- the reference's ValueError).  resolution_host [Bt][2] = (height, width),
 * aspect_ratio_host [Bt]: host arrays, copied.  ecad_step honours per-sample values; ecad_run_schedule requires one value for the
 * whole batch (what the pipeline passes) and returns ECAD_E_INVALID otherwise. */
int ecad_set_added_cond(EcadEngine* e, const float* resolution_host, const float* aspect_ratio_host, int Bt);
/* Replaces the `attention_mask` argument of PixArtTransformer2DEdited.forward and an `encoder_attention_mask` that arrives as a bias
 * (pixart_transformer_2d_edited.py:166-167,213-215,255-291): the reference converts a 2-D 0/1 mask into (1 - m) * -10000 and adds a 3-D
 * [B, 1, keys] bias to the attention scores as given, broadcast over heads and queries.  The caller does the first conversion (same rounding as the
 * reference: in the model dtype) and hands over one float per (sample, key): self_bias_dev [Bt][N] for the self-attention of every block,
 * enc_bias_dev [Bt][T] in place of the bias ecad_begin_image_batch / ecad_run_schedule would derive from their 0/1 mask.  Device pointers, copied on
 * `stream`; NULL switches that bias off (the default).  Must precede ecad_begin_image_batch / ecad_run_schedule of the batch it applies to and stays in
 * force until changed.  A bias that depends on the query row is not representable here (ECAD has no caller that passes one). */
int ecad_set_attention_bias(EcadEngine* e, const float* self_bias_dev, const float* enc_bias_dev, int Bt, int N, int T, void* stream);

/* ---- schedule --------------------------------------------------------------------------- */

/* Replaces the per-component host lookups CacheSchedule.get_recompute
 * (ecad/schedulers/cache_scheduler/cache_schedule.py:68-73).  bits: [steps][blocks][3] bytes
 * (0/1), layout of PixArtCacheSchedule.to_numpy (pixart_cache_schedule.py:15-27).  The rule
 * "flag false + empty cache => recompute" (cached_transformer_block.py:340-347) is resolved here,
 * statically, into the per-step op lists. */
int ecad_set_schedule(EcadEngine* e, const uint8_t* bits, int steps, int blocks);

/* ---- per image batch ---------------------------------------------------------------------- */

/* Start a batch: resets caches + step state (reset_cache, :155-158; _reset_schedules_callback,
 * ecad/image_generators/image_generator.py:193-202) and hoists the step-invariant work of
 * _process_input/_create_attention_mask (:255-330): caption projection and, per block, the
 * cross-attention K/V of the text.  enc_dev: [Bt, T, caption_channels] engine dtype;
 * mask_dev: [Bt, T] uint8 (1 keep / 0 discard), NULL = all keep.  H, W: latent height/width. */
int ecad_begin_image_batch(EcadEngine* e, int Bt, int H, int W, const void* enc_dev, const uint8_t* mask_dev, int T, void* stream);
/* The same step-invariant work (caption projection, text K/V, mask bias) recomputed from new input tensors WITHOUT touching the block caches: the reference's
 * forward takes encoder_hidden_states / encoder_attention_mask at every step and never looks at the caches when they change (they are dropped only by
 * reset_cache, image_generator.py:193-202).  For a batch that has begun with the same geometry (ECAD_E_NOT_READY otherwise). */
int ecad_rebind_batch_inputs(EcadEngine* e, int Bt, int H, int W, const void* enc_dev, const uint8_t* mask_dev, int T, void* stream);

/* One noise prediction == PixArtTransformer2DEdited.forward (:160-253) at schedule row `step_idx`.
 * latents_dev: [Bt, C, H, W] engine dtype; timestep_dev: [Bt] float32; out: [Bt, out_channels, H, W]
 * engine dtype.  Cache reads/writes follow the schedule row; the caller advances step_idx the way the
 * reference's callbacks advance curr_step. */
int ecad_step(EcadEngine* e, int step_idx, const void* latents_dev, const float* timestep_dev, void* noise_pred_out_dev, void* stream);

/* Whole denoising loop == the loop mirrored at ecad/pipelines/pass_through.py:297-380 for one image
 * batch: CFG concat, `steps` transformer calls, guidance combine, learned-sigma drop and the
 * DPM-Solver++(2M) update, captured once into a hipGraph per (schedule, shapes) and replayed.
 * latents_dev: [B, C, H, W] float32, B = Bt/2, updated in place.  enc/mask as in
 * ecad_begin_image_batch (uncond half first).  timesteps_host: [steps] float.
 * solver_coef_host: [steps][5] float = {alpha_s, sigma_s, c_x, c_m0, c_m1} with
 * x0 = (x - sigma_s*eps)/alpha_s; x <- c_x*x + c_m0*x0 + c_m1*x0_prev. */
int ecad_run_schedule(EcadEngine* e, int Bt, int H, int W, const void* enc_dev, const uint8_t* mask_dev, int T,
                      float* latents_dev, const float* timesteps_host, const float* solver_coef_host,
                      float guidance_scale, int use_graph, void* stream);

/* reset_cache (:155-158; block level cached_transformer_block.py:120-123): every cached attn1 / attn2 / ff output is dropped.  A batch that has begun
 * stays begun (its hoisted caption projection / text K/V are recomputations, not caches); the next ecad_step recomputes every component it finds
 * no usable cache for ("flag false + empty cache => recompute", :340-347), exactly as the reference's blocks do after a mid-run reset. */
int ecad_reset_cache(EcadEngine* e);

/* ---- user compute plug-ins ------------------------------------------------------------------------------------------------
 * Replaces CachedTransformerBlock.compute_attn / compute_ff dispatch (cached_transformer_block.py:125-165): a schedule may name, per block, a
 * registered function `(block, attn, hidden, enc, mask, **kw) -> Tensor` / `(block, norm_hidden, **kw) -> Tensor` instead of the cached defaults.
 * mask [blocks][3] (attn1, attn2, ff): 1 = this component is computed by `fn` in ecad_step mode.  For such a component the engine produces the
 * sub-layer input exactly as the block does (LayerNorm + adaLN modulate for attn1 / ff; the residual stream itself for attn2, :264-267) into
 * io_in_dev, calls fn on the launch stream's host thread, and then applies gate * out + hidden (:243-246,289,317-320) to what fn left in
 * io_out_dev.  Caching is the plug-in's business, as in the reference (the defaults are plug-ins there too).  io_in_dev / io_out_dev:
 * caller-owned [max_batch * max_tokens, D] buffers of the engine dtype.  fn returns 0 or non-zero to abort the step (ECAD_E_INVALID).
 * ecad_run_schedule (whole-loop hipGraph) refuses with ECAD_E_UNSUPPORTED while any component is routed to a HOST plug-in (native variants,
 * ecad_set_native_variants below, stay in the graph).  mask NULL / all zero: off. */
typedef int (*EcadComponentFn)(void* user, int step_idx, int block, int component, const void* in_dev, void* out_dev, int Bt, int N, int D, void* stream);
int ecad_set_custom_components(EcadEngine* e, const uint8_t* mask, int blocks, EcadComponentFn fn, void* user, void* io_in_dev, void* io_out_dev);
/* Native variants of the compute plug-ins (ABI 6): gate rules other than the default `out = f(x) if flag or cache is None else cache; cache = out` that the engine
 * lowers into its own kernels, so that a schedule naming them (`custom_compute_attn` / `custom_compute_ff`, pixart_cache_schedule.py:29-37) runs in ecad_step AND in
 * the whole-loop plan / hipGraph of ecad_run_schedule -- a host callback cannot be captured.  kinds [steps][blocks][3] (attn1, attn2, ff), one of:
 *   ECAD_NATIVE_DEFAULT          compute_attn_cached / compute_ff_cached (cached_transformer_block.py:326-391)
 *   ECAD_NATIVE_RECOMPUTE        the schedule's flag is ignored: the sub-layer is evaluated at this step and cached (the reference idiom "never reuse here")
 *   ECAD_NATIVE_CFG_MEAN_CACHE   attn2 only: evaluated per the schedule's flag as the default; what is CACHED is the mean of the unconditional and the text half of
 *                                the CFG-doubled batch, written to both halves, while the hidden states take the un-averaged output -- the caching rule of
 *                                compute_attn_tgate at its gate step (cached_transformer_block.py:443-452); needs an even batch
 * Call after ecad_set_schedule (which resets every component to the default), with that schedule's steps and blocks; NULL: all default.  The executed-component
 * counts and MAC figures of EcadStats follow.  ECAD_E_INVALID for another geometry / an unknown id, ECAD_E_UNSUPPORTED for CFG_MEAN_CACHE on attn1 / ff. */
enum { ECAD_NATIVE_DEFAULT = 0, ECAD_NATIVE_RECOMPUTE = 1, ECAD_NATIVE_CFG_MEAN_CACHE = 2 };
int ecad_set_native_variants(EcadEngine* e, const uint8_t* kinds, int steps, int blocks);
/* The block's native sub-layers for use inside a plug-in = the reference's `self.attn1(h)`, `self.attn2(h, enc, mask)` (which: 1 | 2; attn2 uses the
 * text K/V and mask bias of the batch that has begun) and `self.ff(h)`: x_dev, out_dev [Bt * N, D] engine dtype; bias only, no gate, no cache. */
int ecad_block_attn(EcadEngine* e, int block, int which, const void* x_dev, void* out_dev, void* stream);
int ecad_block_ff(EcadEngine* e, int block, const void* x_dev, void* out_dev, void* stream);
/* What the block passes as `encoder_hidden_states` / `encoder_attention_mask` (pixart_transformer_2d_edited.py:255-291,315-321): the caption-projected
 * text [Bt * T, D] (engine dtype) and the additive bias [Bt * T] float32 (0 keep / -10000 discard) of the batch that has begun; either may be NULL. */
int ecad_copy_batch_context(EcadEngine* e, void* ctx_out_dev, float* bias_out_dev, int* Bt, int* N, int* T, void* stream);
int ecad_cache_stats(EcadEngine* e, EcadStats* s); /* feeds the roofline accounting */

/* Per-kernel-class device time of the most recent NON-graph ecad_run_schedule/ecad_step when
 * profiling is on (hipEvents on the launch stream around every launch).  names: out array of
 * const char* (static storage); ms / launches / flops / bytes: per class.  Returns the class count. */
int ecad_set_profiling(EcadEngine* e, int on);
int ecad_get_profile(EcadEngine* e, int max_classes, const char** names, double* ms, int64_t* launches, double* flops, double* bytes);

/* ---- single-op entry points (parity tests and micro-benchmarks; same kernels the engine launches) */

/* out[M,N] = epilogue(A[M,K] @ W[N,K]^T + bias[N]); epi: 0 bias, 1 bias+GELU(tanh), 2 bias+SiLU,
 * 3 bias -> cache write -> resid += gate * out  (gate_table_f32 [N] + gate_mod [rows/rows_per_batch][..] may be NULL => gate 1). */
int ecad_op_gemm(EcadEngine* e, int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                 void* resid, const float* gate_table, const void* gate_mod, int64_t gate_mod_stride, int rows_per_batch, void* stream);
/* ecad_op_gemm with the kernel named by the caller, so the parity tests reach every GEMM kernel the engine launches with the exact
 * template instantiation of its call site (tests/test_gpu_ops.py).  kernel: ECAD_GEMM_K_*; call_site: ECAD_SITE_* = the engine's
 * call-site class (it only selects the kernel symbol rocprof reports).  ECAD_E_UNSUPPORTED if that kernel cannot run the shape
 * (never a silent substitution).  Bits 8 and up of `kernel`: 0 = the engine's K-loop schedule, 1 + v = force schedule variant v (A/B runs in one process). */
enum { ECAD_GEMM_K_TILE144 = 0 /* gemm3s.h 256x144 */, ECAD_GEMM_K_WIDE288 = 2 /* gemm2w.h 256x288 */, ECAD_GEMM_K_TILE128 = 3 /* gemm.h 128x128, N % 128 == 0 */,
       /* ids 1 and 4 belonged to kernels that left the tree in round 3 (persistent producer / consumer, 128x144 half tile): ECAD_E_UNSUPPORTED */
       ECAD_GEMM_K_LATENCY64 = 5 /* gemm3s.h 64x144, four waves of 16x144: what the engine launches when a GEMM has few rows (batch 1-4) */,
       ECAD_GEMM_K_LATENCY64_SPLITK = 6 /* the same tiles with K in four slices + splitk_rmw_kernel: few rows and K >= 2304 (FF2), epilogue 3 only */,
       ECAD_GEMM_K_NARROW4 = 7 /* gemm3s4.h: the 256x144 tile on four waves with the hand-placed K loop; epilogue 3 at the OUT1 / OUT2 / FF2 sites, whole 256-row tiles, K % 192 == 0 */ };
enum { ECAD_SITE_QKV = 0, ECAD_SITE_OUT1 = 1, ECAD_SITE_Q2 = 2, ECAD_SITE_OUT2 = 3, ECAD_SITE_FF1 = 4, ECAD_SITE_FF2 = 5, ECAD_SITE_TEXTKV = 6,
       ECAD_SITE_CAPTION = 7, ECAD_SITE_SMALL = 8, ECAD_SITE_KV1 = 9 };
int ecad_op_gemm_ex(EcadEngine* e, int kernel, int call_site, int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                    int64_t lda, int64_t ldc, void* resid, const float* gate_table, const void* gate_mod, int64_t gate_mod_stride, int rows_per_batch, void* stream);
/* y = LN(x) * (1 + scale) + shift with scale = table[1*D..] + mod[b][1], shift = table[0] + mod[b][0] (adaLN-single). */
int ecad_op_ln_modulate(EcadEngine* e, const void* x, void* y, int M, int D, const float* table_shift, const float* table_scale,
                        const void* mod_shift, const void* mod_scale, int64_t mod_stride, int rows_per_batch, void* stream);
/* h += sum_i gate_i * C_i  (n_src cached tensors; gate tables may be NULL => 1). */
int ecad_op_cache_accumulate(EcadEngine* e, void* h, int M, int D, int n_src, const void* const* srcs, const float* const* gate_tables,
                             const void* const* gate_mods, int64_t gate_mod_stride, int rows_per_batch, void* stream);
/* Multi-head attention core: out[b,q,h,:] = softmax(Q K^T / sqrt(dh) + bias[b,k]) V.
 * q: rows of stride ldq, k/v likewise; bias_f32 [B, Nk] or NULL. */
int ecad_op_attention(EcadEngine* e, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo,
                      int B, int H, int Nq, int Nk, int dh, const float* bias, void* stream);

/* ecad_op_attention with the kernel named: 0 = what the engine would pick, 1 = attention.h (any head_dim <= 72, % 8 == 0), 2 =
 * attention128.h (head_dim 72 only; ECAD_E_UNSUPPORTED otherwise), 3 = attention_dma.h with four query fragments per wave (head_dim 72, Nq % 512 == 0, Nk % 128 == 0, no bias),
 * 4 = attention_w472.h (head_dim 72, Nq % 256 == 0, Nk % 256 == 0, no bias): one wave per SIMD, generated instruction stream -- what the plans launch for
 * self-attention over >= 1024 tokens (PixArt-Sigma at 1024 x 1024). */
int ecad_op_attention_ex(EcadEngine* e, int kernel, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo,
                         int B, int H, int Nq, int Nk, int dh, const float* bias, void* stream);

/* The production launch of attention on the 256 / 1024 / 4096-token geometries (csrc/gemm3s_attn.h): the query projection of a block with the
 * attention of its rows run in the GEMM tail.   out[b*n_tokens + i, h*72 .. +71] = softmax_j((x Wq^T + bq)[i, h] . k[b, j, h] / sqrt(72) + key_bias[b, j]) v[b, j, h]
 * = attn.to_q + scaled_dot_product_attention of Attention.forward [ext] behind cached_transformer_block.py:231-239 (self; call_site ECAD_SITE_QKV, key_bias
 * ignored, n_keys = n_tokens) and :282-288 (cross; call_site ECAD_SITE_Q2, key_bias = additive mask or null, any n_keys).  x [Bt*n_tokens, D]; wq [D, D];
 * bq fp32 [D]; k / v: row (b*n_keys + j), head h at column h*72, row stride ld_kv elements; out [Bt*n_tokens, D].  D = heads x 72, n_tokens % 256 == 0 (self:
 * also a multiple of 128 keys), else ECAD_E_UNSUPPORTED -- the engine then runs the projection and ecad_op_attention as separate launches, and so can the caller. */
int ecad_op_qproj_attention(EcadEngine* e, int call_site, const void* x, const void* wq, const float* bq, const void* k, const void* v, int64_t ld_kv,
                            const float* key_bias, void* out, int Bt, int n_tokens, int n_keys, void* stream);

/* ---- decode tail: AutoencoderKL decoder + postprocess ------------------------------------------------
 * Replaces `self.vae.decode(latents / scaling_factor)` and `image_processor.postprocess(..., "pil")` inside the reference's
 * timed region (ecad/pipelines/pass_through.py:382-396).  SD / sd-vae-ft-ema geometry; weight names are diffusers'
 * AutoencoderKL state-dict keys ("decoder.up_blocks.1.resnets.0.conv1.weight", ...); encoder.* / quant_conv.* are ignored. */
typedef struct EcadVae EcadVae;
int ecad_vae_create(int device, int dtype, EcadVae** out);
void ecad_vae_destroy(EcadVae* v);
const char* ecad_vae_last_error(const EcadVae* v);
int ecad_vae_load_weights(EcadVae* v, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim);
int ecad_vae_finalize_weights(EcadVae* v);
/* latents_dev: [B, 4, h, w] float32 (un-scaled, as the pipeline holds them); out_u8_dev: [B, 8h, 8w, 3] uint8 RGB = the bytes
 * PIL.Image.fromarray wants, after (x / 2 + 0.5).clamp(0, 1) * 255 rounding. */
int ecad_vae_decode(EcadVae* v, const float* latents_dev, int B, int h, int w, float scaling_factor, uint8_t* out_u8_dev, void* stream);
/* FLUX VAE (16 latent channels, no post_quant_conv, z = latents / scaling_factor + shift_factor; FluxPipeline.__call__ [ext] as called
 * from ecad/image_generators/flux_image_generator.py:330-360): latent_channels 4 or 16. */
int ecad_vae_create_ex(int device, int dtype, int latent_channels, int use_post_quant_conv, EcadVae** out);
int ecad_vae_decode_ex(EcadVae* v, const float* latents_dev, int B, int h, int w, float scaling_factor, float shift_factor, uint8_t* out_u8_dev, void* stream);
/* Roofline accounting of the last ecad_vae_decode*: flops of its convolution GEMMs (2 M Cout taps Cin per launch: the implicit-GEMM view of the 3x3 / 1x1
 * convolutions behind vae.decode, pass_through.py:382-396), their launch count, and how many ran on the 256 x 128 / 3-stage kernel (any pointer may be NULL). */
int ecad_vae_decode_stats(const EcadVae* v, double* conv_flops, int* conv_launches, int* conv3s_launches);
/* ABI 7: the same, plus how many of the launches ran on the one-wave-per-SIMD generated K loop over the zero-bordered layout (conv4_kernel, csrc/conv4.h), and
 * the flops the launches actually multiplied: conv_flops stays the reference's arithmetic (nine taps per 3x3 convolution); the upsampler convolutions run as four
 * 2 x 2 convolutions with pre-summed weights (4/9 of their nine-tap multiplications), so conv_flops_executed <= conv_flops. */
int ecad_vae_decode_stats_ex(const EcadVae* v, double* conv_flops, int* conv_launches, int* conv3s_launches, int* conv4_launches, double* conv_flops_executed);

#ifdef __cplusplus
}
#endif
#endif /* ECAD_HIP_H */
