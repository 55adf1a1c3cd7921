/*
 * ecad_flux.h -- C ABI of the FLUX.1 part of libecad_hip.so: the cached MMDiT denoising loop
 * (19 joint + 38 single-stream blocks, 6 cacheable component kinds) on MI355X (gfx950).
 *
 * Same conventions as ecad_hip.h: plain pointers and sizes, streams as void* (hipStream_t), 0 / negative ECAD_E_* return
 * codes, message from ecad_flux_last_error().  Each entry point names the reference interface it replaces
 * (paths under /root/reference).
 */
#ifndef ECAD_FLUX_H
#define ECAD_FLUX_H

#include "ecad_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Constructor arguments of FluxTransformer2DEdited (ecad/transformer_2d_models/flux_transformer_2d_edited.py:36-66). */
typedef struct EcadFluxCfg {
  int32_t num_layers;             /* 19 joint blocks  */
  int32_t num_single_layers;      /* 38 single blocks */
  int32_t num_heads;              /* 24  */
  int32_t head_dim;               /* 128 (the only supported value) */
  int32_t in_channels;            /* 64 = 16 latent channels x 2 x 2 packing */
  int32_t joint_attention_dim;    /* 4096 */
  int32_t pooled_projection_dim;  /* 768  */
  int32_t guidance_embeds;        /* 1 for FLUX.1-dev */
  int32_t axes_dims_rope[3];      /* 16, 56, 56 (sum = head_dim) */
  int32_t max_batch;              /* largest B */
  int32_t max_image_tokens;       /* largest N = (H/16) * (W/16) */
  int32_t max_text_tokens;        /* largest T (512) */
  int32_t dtype;                  /* ECAD_DTYPE_BF16 (reference: torch.bfloat16, :97-102) or ECAD_DTYPE_F16 */
  float norm_eps;                 /* 1e-6 */
} EcadFluxCfg;

typedef struct EcadFluxStats {
  int64_t components_executed;   /* of steps * (num_layers + num_single_layers) * 3, after "empty cache => recompute" */
  int64_t components_cached;
  int64_t components_dead;       /* executed by the reference but never consumed (single_attn / single_proj_mlp under a cached proj_out): skipped */
  int64_t linear_macs_reference; /* per sample: Linear MACs the reference executes (compute_macs.py model / 2) */
  int64_t linear_macs_executed;  /* per sample: after hoisting the step-invariant context embedder and skipping dead components */
  int64_t attn_core_macs;        /* per sample: QK^T + PV */
  int64_t cache_arena_bytes;
  int64_t weight_bytes;
  int64_t workspace_bytes;
  int32_t graph_nodes;
  int32_t launches_per_batch;
} EcadFluxStats;

typedef struct EcadFlux EcadFlux;

/* Replaces FluxTransformer2DEdited.__init__/_post_init (flux_transformer_2d_edited.py:36-108). */
int ecad_flux_create(const EcadFluxCfg* cfg, int device, EcadFlux** out);
void ecad_flux_destroy(EcadFlux* e);
const char* ecad_flux_last_error(const EcadFlux* e); /* e may be NULL: last create() error */

/* Replaces from_pretrained + init_cached_transformer_blocks (:110-186).  `name` is the diffusers FluxTransformer2DModel
 * state-dict key ("transformer_blocks.3.attn.add_q_proj.weight", "single_transformer_blocks.7.proj_out.bias", ...).
 * ptr: host or device memory; the engine converts, fuses (QKV per stream, all adaLN linears into one matrix) and keeps its own copy. */
int ecad_flux_load_weights(EcadFlux* e, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim);
int ecad_flux_finalize_weights(EcadFlux* e);
/* Opt-in OCP fp8 (e4m3) for the block GEMMs (QKV, attention out, FF, proj_mlp, proj_out): weights get an fp8 copy with per-output-channel
 * scales at finalize, activations are quantised per row right before each GEMM (csrc/fgemm8.h).  Call before ecad_flux_finalize_weights
 * (a later call requires finalizing again).  Default off: the reference computes in bf16.  Env ECAD_FLUX_FP8=1 sets the default. */
int ecad_flux_set_fp8(EcadFlux* e, int on);

/* Replaces the per-component FluxCacheSchedule.get_recompute lookups (ecad/schedulers/cache_scheduler/cache_schedule.py:68-73 via
 * cached_flux_transformer_block.py:53-55,76-78,168-170,205-207).  bits_full: [steps][num_layers][3] = (full_attn, full_ff,
 * full_ff_context); bits_single: [steps][num_single_layers][3] = (single_attn, single_proj_mlp, single_proj_out) -- the two halves of
 * each step row of FluxCacheSchedule.to_numpy (flux_cache_schedule.py:63-88).  "flag false + empty cache => recompute" is
 * resolved here, statically. */
int ecad_flux_set_schedule(EcadFlux* e, const uint8_t* bits_full, const uint8_t* bits_single, int steps);

/* Start a batch: reset_cache (:188-193) + the step-invariant context_embedder / pooled-text embedding / RoPE table
 * (forward :268-285).  enc_dev [B, T, joint_attention_dim], pooled_dev [B, pooled_projection_dim]: engine dtype.
 * grid_h x grid_w: packed latent grid (image H/16 x W/16); image ids are (0, row, col), text ids 0 (FluxPipeline [ext]). */
int ecad_flux_begin_image_batch(EcadFlux* e, int B, int grid_h, int grid_w, const void* enc_dev, const void* pooled_dev, int T, void* stream);

/* One velocity prediction == FluxTransformer2DEdited.forward (:216-326) at schedule row step_idx.
 * latents_dev [B, N, in_channels] engine dtype (packed); timestep_x1000 / guidance_x1000: the values the model sees after its
 * `* 1000` (:270-274), shared by the batch; out_dev [B, N, in_channels] engine dtype. */
int ecad_flux_step(EcadFlux* e, int step_idx, const void* latents_dev, float timestep_x1000, float guidance_x1000, void* out_dev, void* stream);

/* Whole denoising loop of FluxPipeline.__call__ [ext] as driven by ecad/image_generators/flux_image_generator.py:330-360:
 * `steps` transformer calls + the FlowMatchEuler update x <- x + (sigma[i+1] - sigma[i]) * v (latents kept in the model
 * dtype between steps), captured once into a hipGraph per (schedule, shapes) and replayed.  latents_dev [B, N, in_channels]
 * float32, updated in place.  timesteps_x1000_host [steps], sigmas_host [steps + 1]. */
int ecad_flux_run_schedule(EcadFlux* e, int B, int grid_h, int grid_w, const void* enc_dev, const void* pooled_dev, int T, float* latents_dev,
                           const float* timesteps_x1000_host, float guidance_x1000, const float* sigmas_host, int use_graph, void* stream);

int ecad_flux_reset_cache(EcadFlux* e); /* reset_cache (:188-193) */
int ecad_flux_stats(EcadFlux* e, EcadFluxStats* s);
int ecad_flux_set_profiling(EcadFlux* e, int on); /* as ecad_set_profiling */
int ecad_flux_get_profile(EcadFlux* e, int max_classes, const char** names, double* ms, int64_t* launches, double* flops, double* bytes);

/* ---- single-op entry points (parity tests, micro-benchmarks; same kernels the engine launches) */
/* out = epi([A | A2] @ W^T + bias); epi 0 bias, 1 bias+GELU(tanh), 2 bias -> optional copy to out -> resid += gate[b] * value, 3 bias+SiLU.
 * bn: 128 or 256 (tile width; N % bn == 0).  A2 (may be NULL) supplies K columns >= k_split.  Rows [0, seg0) belong to samples of
 * rows_per_batch0 rows, the rest to samples of rows_per_batch1 rows. */
int ecad_flux_op_gemm(EcadFlux* e, int epi, int bn, const void* A, const void* A2, int k_split, const void* W, const float* bias, void* out, int M, int N,
                      int K, void* resid, const void* gate, int64_t gate_stride, int seg0, int rows_per_batch0, int rows_per_batch1, void* stream);
/* fp8 (OCP e4m3) variant of the plain GEMM, for parity / speed probes: A [M,K] and W [N,K] (engine dtype) are quantised on the device
 * (per-row / per-output-channel amax scales), out = epi((A8 @ W8^T) * sa * sw + bias), epi 0 or 1.  The GEMM alone is then repeated
 * `iters` times and its average duration returned in *avg_us (may be NULL). */
int ecad_flux_op_gemm_fp8(EcadFlux* e, int epi, int bn, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int iters, float* avg_us,
                          void* stream);
/* In-place per-head RMSNorm + RoPE on the q and k thirds of qkv [B*(T+N), 3*heads*128] (text rows first); rope_f32 [T+N][64][2] (cos, sin). */
int ecad_flux_op_qk_norm_rope(EcadFlux* e, void* qkv, int B, int T, int N, const float* wq, const float* wk, const float* wq_txt, const float* wk_txt,
                              const float* rope_f32, void* stream);
/* Joint attention over [text | image] rows: out [B*(T+N), heads*128] from qkv [B*(T+N), 3*heads*128]. */
int ecad_flux_op_attention(EcadFlux* e, const void* qkv, void* out, int B, int T, int N, void* stream);
/* y = LN(x) * (1 + scale[b]) + shift[b] over M = B*(T+N) joint rows (pass T = 0 for plain [B, N] rows). */
int ecad_flux_op_ln_modulate(EcadFlux* e, const void* x, void* y, int B, int T, int N, const void* shift, const void* scale, int64_t mod_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ECAD_FLUX_H */
