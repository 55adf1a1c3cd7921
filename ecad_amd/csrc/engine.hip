// engine.hip -- the engine behind include/ecad_hip.h: owns weights (re-laid-out), the cache arena
// [L][3][Bt*N][D], workspaces, the per-step op lists compiled from the schedule bits, and the captured
// hipGraph of a whole image batch.  No torch, no other GPU library: HIP runtime + the kernels in this directory.
//
// Reference semantics implemented here (paths under /root/reference/ecad):
//   transformer_2d_models/pixart_transformer_2d_edited.py:160-376  forward / _process_input / _create_output
//   transformer_blocks/cached_transformer_block.py:167-391          block data flow + cache gate
//   schedulers/cache_scheduler/cache_schedule.py:58-73              schedule[step][block][component]
//   pipelines/pass_through.py:297-380                               CFG / learned-sigma drop / scheduler step
#include "../../include/ecad_hip.h"

#if (defined(G_STAMP) || defined(G2_BREAK_TILE) || defined(ATT_ABL) || defined(ATT_STAMP)) && !defined(ECAD_DIAG)
#error "in-kernel stamps and wrong-result ablations exist in diagnostic builds only: make -C ecad_amd/csrc diag (-DECAD_DIAG, output build_tmp/libecad_diag.so)"
#endif
#if defined(ECAD_DIAG) && defined(ECAD_PRODUCT_BUILD)
#error "the product library (ecad_amd/libecad_hip.so) is never built with -DECAD_DIAG"
#endif

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <list>
#include <map>
#include <string>
#include <vector>

#include "attention.h"
#include "attention128.h"
#include "attention_dma.h"
#include "common.h"
#include "elementwise.h"
#include "gemm.h"
#include "final_layer.h"
#include "gemm2w.h"
#include "gemm2w4.h"
#include "gemm3s4.h"
#include "attention_w472.h"
#include "gemm3s.h"
#include "gemm3s8.h"

namespace ecad {

static std::string g_create_error;
constexpr int PROJ_PAD = 144;  // proj_out rows padded to one GEMM tile column

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) return fail(ECAD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

enum KClass {
  KC_GEMM_QKV = 0, KC_GEMM_OUT1, KC_GEMM_Q2, KC_GEMM_OUT2, KC_GEMM_FF1, KC_GEMM_FF2, KC_GEMM_TEXTKV, KC_GEMM_CAPTION,
  KC_GEMM_SMALL, KC_GEMM_KV1, KC_ATTN_SELF, KC_ATTN_CROSS, KC_LN, KC_CACHE_REPLAY, KC_PATCH, KC_MISC, KC_QUANT, KC_CACHE_HIT_STEP, KC_COUNT
};
static const char* kclass_names[KC_COUNT] = {
    "gemm_qkv", "gemm_attn1_out", "gemm_attn2_q", "gemm_attn2_out", "gemm_ff1_gelu", "gemm_ff2", "gemm_text_kv", "gemm_caption",
    "gemm_small", "gemm_attn1_kv_wide", "attn_self", "attn_cross", "ln_modulate", "cache_replay", "patch_embed", "misc", "fp8_quant_rows", "cache_hit_step"};

struct EngineBase {
  EcadModelCfg cfg;
  int dev = 0;
  std::string err;
  virtual ~EngineBase() {}
  int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
  virtual int init() = 0;
  virtual int load_weight(const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) = 0;
  virtual int finalize_weights() = 0;
  virtual int share_weights(EngineBase* src) = 0;
  virtual int set_fp8(int on) = 0;
  virtual int set_ln_fold(int on) = 0;
  virtual int set_batches_in_flight(int n) = 0;
  virtual int set_split_k(int on) = 0;
  virtual int set_native(const uint8_t* kinds, int steps, int blocks) = 0;
  virtual int set_added_cond(const float* resolution, const float* aspect_ratio, int Bt) = 0;
  virtual int set_attention_bias(const float* self_bias, const float* enc_bias, int Bt, int N, int T, hipStream_t st) = 0;
  virtual int set_schedule(const uint8_t* bits, int steps, int blocks) = 0;
  virtual int begin_batch(int Bt, int H, int W, const void* enc, const uint8_t* mask, int T, hipStream_t st, bool keep_caches = false) = 0;
  virtual int step(int step_idx, const void* latents, const float* timestep, void* out, hipStream_t st) = 0;
  virtual int run_schedule(int Bt, int H, int W, const void* enc, const uint8_t* mask, int T, float* latents, const float* ts, const float* coef,
                           float guidance, int use_graph, hipStream_t st) = 0;
  virtual int reset_cache() = 0;
  virtual int set_custom(const uint8_t* mask, int blocks, EcadComponentFn fn, void* user, void* io_in, void* io_out) = 0;
  virtual int block_attn(int block, int which, const void* x, void* out, hipStream_t st) = 0;
  virtual int block_ff(int block, const void* x, void* out, hipStream_t st) = 0;
  virtual int copy_context(void* ctx_out, float* bias_out, int* Bt, int* N, int* T, hipStream_t st) = 0;
  virtual int stats(EcadStats* s) = 0;
  virtual int set_profiling(int on) = 0;
  virtual int get_profile(int maxc, const char** names, double* ms, int64_t* launches, double* flops, double* bytes) = 0;
  virtual int op_gemm(int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, void* resid, const float* gate_table,
                      const void* gate_mod, int64_t gate_mod_stride, int rows_per_batch, hipStream_t st) = 0;
  virtual int op_gemm_ex(int kernel, int site, int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int64_t lda, int64_t ldc,
                         void* resid, const float* gate_table, const void* gate_mod, int64_t gate_mod_stride, int rows_per_batch, hipStream_t st) = 0;
  virtual int op_qproj_attention(int site, const void* x, const void* wq, const float* bq, const void* k, const void* v, int64_t ld_kv, const float* key_bias, void* out,
                                 int Bt_, int n_tokens, int n_keys, hipStream_t st) = 0;
  virtual int op_attention_ex(int kernel, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo, int B, int H,
                              int Nq, int Nk, int dh, const float* bias, hipStream_t st) = 0;
  virtual int op_ln(const void* x, void* y, int M, int D, const float* tsh, const float* tsc, const void* msh, const void* msc, int64_t mstride,
                    int rows_per_batch, hipStream_t st) = 0;
  virtual int op_accumulate(void* h, int M, int D, int n_src, const void* const* srcs, const float* const* gts, const void* const* gms, int64_t mstride,
                            int rows_per_batch, hipStream_t st) = 0;
  virtual int op_attention(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo, int B, int H,
                           int Nq, int Nk, int dh, const float* bias, hipStream_t st) = 0;
};

// ------------------------------------------------------------------------------------------------
template <typename T> struct Op {
  enum Kind { GEMM, GEMM8, QUANT, ATTN, LN, REPLAY, PATCH, SINUS, SILU, MASKBIAS, UNPATCH, SOLVER, DUP, SIZEEMB, ADDSIZE, COMB, CUSTOM, FINAL, CFGMEAN } kind;
  int cls = KC_MISC;
  int epi = 0;
  GemmArgs<T> g{};
  const uint8_t* w8 = nullptr;  // GEMM8 (opt-in fp8 mode): e4m3 weight copy + per-output-channel scales
  const float* sw8 = nullptr;
  AttnArgs a{};
  bool attn_masked = false;
  FlushArgs f{};
  int desc_off = -1;  // offset into the descriptor array (elements), patched to a device pointer at upload
  // misc payload
  const void* p0 = nullptr;
  const void* p1 = nullptr;
  void* p2 = nullptr;
  void* p3 = nullptr;
  size_t n = 0;
  int i0 = 0, i1 = 0;
  bool call_latents = false, call_timestep = false, call_out = false;  // pointers bound per ecad_step call
  double flops = 0, bytes = 0;
};

template <typename T> class EngineT : public EngineBase {
 public:
  // ---- geometry
  int L, NH, DH, D, CIN, COUT, P, CAP;
  // ---- weights
  char* warena = nullptr;
  size_t warena_bytes = 0, warena_used = 0;
  bool weights_shared = false;  // ecad_share_weights: warena and every pointer into it came from another engine
  // The arena is owned jointly by the engine that loaded the weights and every context that borrows them (ecad_share_weights): the last one to go frees it, so
  // destroying the donor first leaves the borrowers intact.  `gen` counts changes of the arena's content (ecad_load_weights / ecad_finalize_weights on the owner):
  // a borrower that sees a generation it has not lowered its plans for drops them (plan sets, hipGraph, the LayerNorm-fold copies W' / c1 / c2 are all
  // derived from the weights) before it runs anything.
  // Shared by an owner and its borrowers (share_weights).  refs / gen / finalized are plain fields: ecad_create / ecad_share_weights / ecad_load_weights /
  // ecad_finalize_weights / ecad_destroy of engines that share an arena must be serialized by the caller (include/ecad_hip.h says so at ecad_share_weights);
  // the per-batch calls only read them.
  struct W8 { uint8_t* w; float* s; };
  // (fp8 / w8 / w8allocs: the e4m3 weight copies of the owner's fp8 mode, keyed by the 16-bit weight they mirror; they live and die with the arena, so that the
  // contexts that borrow it multiply the same copies and outlive the engine that made them, as they outlive it for the 16-bit weights)
  struct SharedArena { char* base; int refs; uint64_t gen; bool finalized; bool fp8; std::map<const void*, W8> w8; std::vector<void*> w8allocs; };
  SharedArena* sarena = nullptr;
  uint64_t seen_gen = 0;
  struct Dest { void* ptr; bool f32; int64_t rows, cols, ld; bool loaded; };
  std::map<std::string, Dest> dests;
  struct Blk {
    T *w_qkv, *w_o1, *w_q2, *w_o2, *w_ff1, *w_ff2;
    float *b_qkv, *b_o1, *b_q2, *b_o2, *b_ff1, *b_ff2, *sst;
  };
  std::vector<Blk> blk;
  T *w_kv2_all, *w_patch, *w_t1, *w_t2, *w_ada, *w_c1, *w_c2, *w_proj;
  // micro-conditioning of the 1024-MS alpha checkpoints (use_additional_conditions): resolution / aspect-ratio embedders, E = D / 3 wide
  T *w_sz1[2] = {nullptr, nullptr}, *w_sz2[2] = {nullptr, nullptr};
  float *b_sz1[2] = {nullptr, nullptr}, *b_sz2[2] = {nullptr, nullptr};
  float* cond_dev = nullptr;  // [max_batch][3] = (height, width, aspect ratio) per sample
  T* size_emb = nullptr;      // [max_batch][D] = [res_h | res_w | aspect] embeddings, computed once per image batch
  std::vector<float> cond_host;
  int cond_bt = 0;
  bool cond_uniform = true;
  float *b_kv2_all, *b_patch, *b_t1, *b_t2, *b_ada, *b_c1, *b_c2, *b_proj, *sst_out;
  float* pos = nullptr;  // sincos table of the current token grid: its own allocation (never part of a shared weight arena)
  bool finalized = false;
  int pos_grid = 0;
  // ---- opt-in fp8 (e4m3) block GEMMs: gemm3s8.h; env ECAD_FP8=1 or ecad_set_fp8
  bool fp8 = false;
  std::map<const void*, W8> w8map;  // this engine's view of sarena->w8
  uint8_t* a8 = nullptr;  // [M][4D] activation rows quantised in front of each fp8 GEMM
  float* sa8 = nullptr;
  const void* a8_src = nullptr;  // plan construction: which tensor a8 currently holds (Q and K/V launches share one pass)
  // FF1 -> FF2 without the row quantiser (round 6, gemm3s8.h): FF1's epilogue leaves gelu(.) as e4m3 rows + one E8M0 byte per (row, 288-column tile) in a8b / e8buf,
  // FF2's MFMAs take the bytes as block scales.  env ECAD_FP8_SEG=0: the 16-bit ffmid + the row quantiser, as in round 5.
  bool fp8_seg = true;
  uint8_t* a8b = nullptr;    // [M][4D] e4m3
  uint8_t* e8buf = nullptr;  // [M][16]
  const void* a8b_src = nullptr;  // plan construction: the tensor a8b / e8buf mirror (ffmid right after an emitting FF1)
  bool f8_attr = false;
  // ---- workspaces
  char* ws = nullptr;
  size_t ws_bytes = 0;
  T *h, *xn, *qkv, *attn_o, *qc, *ffmid, *ctx_mid, *ctx, *kv2, *cache, *proj, *model_in, *noise, *enc_buf;
  T *tsin, *temb1, *temb, *temb_silu, *mod_all;
  T* comb = nullptr;  // [max_steps][L][4][D]: LN shift / scale of every (step, block) of a shared-timestep plan (comb_mod_kernel)
  bool comb_live = false;
  // ---- LayerNorm folded into the consumer GEMMs of a shared-timestep plan (env ECAD_LN_FOLD=0: off)
  bool ln_fold = true;
  bool fold_plan_on = false;          // build_plans: this plan set folds (geometry, dtype mode and kernels permit it, arena acquired)
  float* hstat = nullptr;             // [D / 144][max rows][2]: per-row partial (sum x, sum x^2) of the hidden states, written by the read-modify-write GEMMs
  struct FoldArena { char* base = nullptr; size_t cap = 0; };
  std::vector<FoldArena> fold_pool;   // arenas of parked / evicted plan sets, recycled: no hipFree (a device-wide sync) on a schedule change while they fit the budget
  size_t fold_budget = (size_t)24 << 30;  // bytes this engine may hold in fold arenas (live + parked plan sets + pool); env ECAD_LN_FOLD_BUDGET_MB.  An all-executed 20-step
                                      // schedule needs ~10 GB per plan set, `ours_fast` 3.3 GB; a plan set that does not fit is lowered WITHOUT the fold (same results
                                      // within the fold's tolerance, ~1.5 % slower) instead of failing
  FoldArena fold_cur;                 // pre-scaled weights W', c1, c2 of the live plan set
  std::vector<FoldDesc> fold_descs;   // one per folded (step, block, attn1 | ff); uploaded behind the weights in fold_cur
  size_t fold_used = 0;
  bool fold_built = false;            // the arena holds what fold_descs describe, for the timesteps in ts_host
  size_t fold_prep_ops = 0;           // plan_run[n_begin .. n_begin + fold_prep_ops) = timestep MLP + comb vectors: what the prep needs before the first replay
  size_t fold_prep_begin = 0;
  int ln_folded = 0, fold_preps = 0, fold_refused = 0;
  bool xsrc_fold = true;   // cached components replayed in the epilogue of the read-modify-write GEMM in front of them (env ECAD_REPLAY_FOLD=0: pending list only)
  int xsrc_folded = 0;
  std::vector<float> fold_ts;         // the timesteps the arena was prepared for
  float *bias_buf, *x0_prev, *lat_buf, *ts_dev, *coef_dev;
  float* splitk_ws = nullptr;
  size_t splitk_rows = 0;
  static constexpr int SPLITK_SLICES = 4;
  bool split_k = true;   // few-rows FF2 as K-slices + splitk_rmw_kernel (batch 1: -14 % per image; env ECAD_SPLIT_K=0 off).  On by default since round 5: the fp32 sum is then
                         // taken in another order than in the sequential-K kernels, so a sample run alone (batch 1-3) equals the same sample inside a larger batch only
                         // to one rounding of FF2's output, not bit for bit -- the reference gives no such guarantee either.  The rule looks at the GEMM's own
                         // rows only, never at the batches in flight: execution contexts and a lone engine still agree bit for bit.
  float *x0_prev2 = nullptr, *lat_buf2 = nullptr;  // second solver-state pair: the update of step s rides in the patch embed of step s + 1 and writes the pair it does not read (elementwise.h PatchSolve)
  uint8_t* mask_buf;
  // additive attention biases handed over as such (forward(attention_mask=..., encoder_attention_mask=<3-D bias>), pixart_transformer_2d_edited.py:255-291):
  // one float per (sample, key), broadcast over heads and queries.  self: [Bt][N] for attn1; enc: [Bt][T] replaces the bias derived from the 0/1 mask.
  float *self_bias_buf = nullptr, *enc_bias_user = nullptr;
  bool self_bias_on = false, enc_bias_on = false;
  int bias_bt = 0, bias_n = 0, bias_t = 0;  // geometry the active attention bias was copied with (set_attention_bias)
  SrcDesc* desc_dev = nullptr;
  size_t desc_cap = 0;
  // ---- schedule / plans
  int steps = 0;
  std::vector<uint8_t> bits, executed;
  int Bt = 0, Hl = 0, Wl = 0, N = 0, Tt = 0;
  bool plan_valid = false;
  std::vector<std::vector<Op<T>>> plan_step;  // ecad_step mode (per-sample timesteps)
  std::vector<Op<T>> plan_begin;              // begin_image_batch ops (shared)
  std::vector<Op<T>> plan_run;                // whole batch for run_schedule (shared timestep per step)
  std::vector<SrcDesc> desc_host;
  std::vector<float> coef_host, ts_host;
  EcadStats st{};
  // ---- plan sets of recently used (schedule, shapes): op lists + descriptors + the captured graph, keyed by content, so alternating between a few
  // schedules (the GA's reference run + candidates, a server's handful of presets) neither re-lowers nor re-captures (env ECAD_PLAN_CACHE=n, default 4)
  struct PlanStash {
    std::string key;
    std::vector<Op<T>> begin, run;
    std::vector<std::vector<Op<T>>> step;
    std::vector<SrcDesc> dh;
    SrcDesc* dd = nullptr;
    size_t dcap = 0;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    float gg = 0.f;
    EcadStats st{};
    bool comb = false;
    FoldArena fold;                    // the set's pre-scaled weights travel with it (no re-preparation when the GA alternates between schedules)
    std::vector<FoldDesc> fdescs;
    std::vector<float> fts;
    size_t fused = 0, fprep = 0, fbegin = 0;
    bool fbuilt = false;
    int folded = 0;
  };
  std::list<PlanStash> plan_lru;
  size_t plan_lru_cap = 4;
  std::string cur_key;
  int plan_builds = 0, graph_captures = 0;
  // ---- ecad_step: which cache slots hold a usable value RIGHT NOW (reset by begin_image_batch / reset_cache; the step plans keep every cache write --
  // dead-write elimination is the whole-loop plan's alone).  A step whose "flag or no usable cache" set differs from the static resolution runs an ad-hoc plan.
  std::vector<uint8_t> valid_rt;
  SrcDesc* desc_dyn_dev = nullptr;
  size_t desc_dyn_cap = 0, desc_dyn_pos = 0;
  int dynamic_steps = 0;
  // ---- user compute plug-ins (cached_transformer_block.py:125-165): components routed to a host callback in ecad_step mode
  std::vector<uint8_t> custom_mask;  // [L][3]
  bool any_custom = false;
  EcadComponentFn custom_fn = nullptr;
  void* custom_user = nullptr;
  T* io_in = nullptr;
  T* io_out = nullptr;
  int cur_step_idx = 0;
  // ---- graph
  hipStream_t cap_stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  float graph_guidance = 0.f;
  // ---- profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_cls;
  double prof_ms[KC_COUNT], prof_flops[KC_COUNT], prof_bytes[KC_COUNT];
  int64_t prof_launch[KC_COUNT];
  // ---- per-call bindings for ecad_step
  const void* call_latents = nullptr;
  const float* call_timestep = nullptr;
  void* call_out = nullptr;
  bool batch_begun = false;

  ~EngineT() override {
    (void)hipSetDevice(dev);  // teardown: nothing to report an error to
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    for (auto& e : plan_lru) free_stash(e);
    if (fold_cur.base) (void)hipFree(fold_cur.base);
    for (auto& a : fold_pool) if (a.base) (void)hipFree(a.base);
    if (desc_dyn_dev) (void)hipFree(desc_dyn_dev);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    if (sarena) {
      if (--sarena->refs == 0) { (void)hipFree(sarena->base); for (void* q : sarena->w8allocs) (void)hipFree(q); delete sarena; }
    } else if (warena) (void)hipFree(warena);
    if (pos) (void)hipFree(pos);
    if (ws) (void)hipFree(ws);
    if (a8) (void)hipFree(a8);
    if (a8b) (void)hipFree(a8b);
    if (e8buf) (void)hipFree(e8buf);
    if (sa8) (void)hipFree(sa8);
    if (desc_dev) (void)hipFree(desc_dev);
  }

  // -------------------------------------------------------------------------------------------
  template <typename U> U* walloc(size_t n) {
    size_t bytes = (n * sizeof(U) + 255) & ~(size_t)255;
    U* p = reinterpret_cast<U*>(warena + warena_used);
    warena_used += bytes;
    return p;
  }
  void reg(const std::string& name, void* ptr, bool f32, int64_t rows, int64_t cols, int64_t ld) { dests[name] = Dest{ptr, f32, rows, cols, ld, false}; }

  int init() override {
    L = cfg.num_layers; NH = cfg.num_heads; DH = cfg.head_dim; D = NH * DH;
    CIN = cfg.in_channels; COUT = cfg.out_channels; P = cfg.patch_size; CAP = cfg.caption_channels;
    if (D % 128 != 0 || D % 64 != 0) return fail(ECAD_E_UNSUPPORTED, "inner dim %d must be a multiple of 128", D);
    if (DH % 8 != 0 || DH > 72) return fail(ECAD_E_UNSUPPORTED, "head_dim %d unsupported (need %%8 == 0 and <= 72)", DH);
    if (CIN != 4 || P != 2) return fail(ECAD_E_UNSUPPORTED, "patch embed supports in_channels 4, patch 2");
    if (P * P * COUT > PROJ_PAD) return fail(ECAD_E_UNSUPPORTED, "proj_out too wide");
    if (CAP % 64 != 0) return fail(ECAD_E_UNSUPPORTED, "caption_channels must be a multiple of 64");
    if (cfg.max_batch < 1 || cfg.max_tokens < 1 || cfg.max_text_tokens < 1) return fail(ECAD_E_INVALID, "max_batch/max_tokens/max_text_tokens must be positive");
    if (cfg.use_additional_conditions && (D % 3 != 0 || (D / 3) % 4 != 0)) return fail(ECAD_E_UNSUPPORTED, "use_additional_conditions needs inner dim %d divisible by 12", D);
    HIPCHK(hipSetDevice(dev));
    HIPCHK(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    for (int c = 0; c < KC_COUNT; ++c) gemm_impl_cls[c] = 0;
    gemm_impl_cls[KC_GEMM_QKV] = gemm_impl_cls[KC_GEMM_OUT1] = gemm_impl_cls[KC_GEMM_OUT2] = gemm_impl_cls[KC_GEMM_Q2] = 0;
    // since gemm3s consumes its second fragment set across the K-step barrier it beats the persistent producer / consumer kernel on FF2
    // too (in-run: 19.77 -> 18.80 ms per batch = 1008 TFLOP/s) and on the caption / small GEMMs (pgemm.h, which nothing has launched since, was retired in round 3: scripts/retired_pgemm.h)
    gemm_impl_cls[KC_GEMM_FF2] = gemm_impl_cls[KC_GEMM_CAPTION] = gemm_impl_cls[KC_GEMM_SMALL] = 0;
    gemm_impl_cls[KC_GEMM_FF1] = gemm_impl_cls[KC_GEMM_TEXTKV] = gemm_impl_cls[KC_GEMM_KV1] = 2;
    if (const char* l = getenv("ECAD_FUSE_ATTN")) fuse_attn = atoi(l) != 0;
    if (const char* l = getenv("ECAD_ATTN_W472")) attn_w472_on = atoi(l) != 0;
    if (const char* l = getenv("ECAD_PLAN_CACHE")) plan_lru_cap = (size_t)(atoi(l) > 0 ? atoi(l) : 0);
    custom_mask.assign((size_t)cfg.num_layers * 3, 0);
    valid_rt.assign((size_t)cfg.num_layers * 3, 0);
    if (const char* l = getenv("ECAD_LN_FOLD")) ln_fold = atoi(l) != 0;
    if (const char* l = getenv("ECAD_REPLAY_FOLD")) xsrc_fold = atoi(l) != 0;
    if (const char* l = getenv("ECAD_LN_FOLD_BUDGET_MB")) fold_budget = (size_t)(atoll(l) > 0 ? atoll(l) : 0) << 20;
    if (const char* l = getenv("ECAD_FP8")) fp8 = atoi(l) != 0;
    if (const char* l = getenv("ECAD_LATENCY_TILES")) latency_tiles = atoi(l) != 0;
    if (const char* l = getenv("ECAD_TAIL_SPLIT")) tail_split = atoi(l) != 0;
    if (const char* l = getenv("ECAD_SPLIT_K")) split_k = atoi(l) != 0;
    if (const char* l = getenv("ECAD_GEMM4")) gemm4_wide = atoi(l) != 0;
    if (const char* l = getenv("ECAD_WIDE_RMW")) wide_rmw = atoi(l) != 0;
    if (const char* l = getenv("ECAD_GEMM3S4")) narrow4 = atoi(l) < 0 ? 0 : atoi(l);
    if (const char* l = getenv("ECAD_FP8_SEG")) fp8_seg = atoi(l) != 0;
    { hipDeviceProp_t pr; HIPCHK(hipGetDeviceProperties(&pr, dev)); num_cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }

    // ---- weight arena
    const size_t d = D;
    size_t per_blk_T = d * 3 * d + d * d * 3 + 4 * d * d * 2;  // qkv, o1, q2, o2, ff1, ff2
    size_t per_blk_f = 3 * d + d * 3 + 4 * d + d + 6 * d;
    size_t glob_T = (size_t)L * 2 * d * d + d * 16 + d * 256 + d * d + 6 * d * d + d * CAP + d * d + PROJ_PAD * d;
    size_t glob_f = (size_t)L * 2 * d + d + d + d + 6 * d + d + d + PROJ_PAD + 2 * d + (size_t)cfg.max_tokens * d;
    const size_t E = d / 3;
    if (cfg.use_additional_conditions) { glob_T += 2 * (E * 256 + E * E); glob_f += 4 * E; }
    warena_bytes = (per_blk_T * L + glob_T) * sizeof(T) + (per_blk_f * L + glob_f) * sizeof(float) + (size_t)(L * 16 + 64) * 256;
    HIPCHK(hipMalloc((void**)&warena, warena_bytes));
    HIPCHK(hipMemset(warena, 0, warena_bytes));
    sarena = new SharedArena{warena, 1, 0, false, false, {}, {}};
    blk.resize(L);
    w_kv2_all = walloc<T>((size_t)L * 2 * d * d);
    b_kv2_all = walloc<float>((size_t)L * 2 * d);
    for (int i = 0; i < L; ++i) {
      Blk& b = blk[i];
      b.w_qkv = walloc<T>(3 * d * d); b.b_qkv = walloc<float>(3 * d);
      b.w_o1 = walloc<T>(d * d); b.b_o1 = walloc<float>(d);
      b.w_q2 = walloc<T>(d * d); b.b_q2 = walloc<float>(d);
      b.w_o2 = walloc<T>(d * d); b.b_o2 = walloc<float>(d);
      b.w_ff1 = walloc<T>(4 * d * d); b.b_ff1 = walloc<float>(4 * d);
      b.w_ff2 = walloc<T>(4 * d * d); b.b_ff2 = walloc<float>(d);
      b.sst = walloc<float>(6 * d);
      const std::string p = "transformer_blocks." + std::to_string(i) + ".";
      const char* qkvn[3] = {"to_q", "to_k", "to_v"};
      for (int j = 0; j < 3; ++j) {
        reg(p + "attn1." + qkvn[j] + ".weight", b.w_qkv + (size_t)j * d * d, false, d, d, d);
        reg(p + "attn1." + qkvn[j] + ".bias", b.b_qkv + (size_t)j * d, true, 1, d, d);
      }
      reg(p + "attn1.to_out.0.weight", b.w_o1, false, d, d, d);
      reg(p + "attn1.to_out.0.bias", b.b_o1, true, 1, d, d);
      reg(p + "attn2.to_q.weight", b.w_q2, false, d, d, d);
      reg(p + "attn2.to_q.bias", b.b_q2, true, 1, d, d);
      for (int j = 0; j < 2; ++j) {
        reg(p + "attn2." + qkvn[j + 1] + ".weight", w_kv2_all + ((size_t)i * 2 + j) * d * d, false, d, d, d);
        reg(p + "attn2." + qkvn[j + 1] + ".bias", b_kv2_all + ((size_t)i * 2 + j) * d, true, 1, d, d);
      }
      reg(p + "attn2.to_out.0.weight", b.w_o2, false, d, d, d);
      reg(p + "attn2.to_out.0.bias", b.b_o2, true, 1, d, d);
      reg(p + "ff.net.0.proj.weight", b.w_ff1, false, 4 * d, d, d);
      reg(p + "ff.net.0.proj.bias", b.b_ff1, true, 1, 4 * d, 4 * d);
      reg(p + "ff.net.2.weight", b.w_ff2, false, d, 4 * d, 4 * d);
      reg(p + "ff.net.2.bias", b.b_ff2, true, 1, d, d);
      reg(p + "scale_shift_table", b.sst, true, 6, d, d);
    }
    w_patch = walloc<T>(d * 16); b_patch = walloc<float>(d);
    w_t1 = walloc<T>(d * 256); b_t1 = walloc<float>(d);
    w_t2 = walloc<T>(d * d); b_t2 = walloc<float>(d);
    w_ada = walloc<T>(6 * d * d); b_ada = walloc<float>(6 * d);
    w_c1 = walloc<T>(d * CAP); b_c1 = walloc<float>(d);
    w_c2 = walloc<T>(d * d); b_c2 = walloc<float>(d);
    w_proj = walloc<T>(PROJ_PAD * d); b_proj = walloc<float>(PROJ_PAD);
    sst_out = walloc<float>(2 * d);
    HIPCHK(hipMalloc((void**)&pos, (size_t)cfg.max_tokens * d * sizeof(float)));  // per engine (geometry-dependent), never part of a shared weight arena
    reg("pos_embed.proj.weight", w_patch, false, d, CIN * P * P, CIN * P * P);
    reg("pos_embed.proj.bias", b_patch, true, 1, d, d);
    reg("adaln_single.emb.timestep_embedder.linear_1.weight", w_t1, false, d, 256, 256);
    reg("adaln_single.emb.timestep_embedder.linear_1.bias", b_t1, true, 1, d, d);
    reg("adaln_single.emb.timestep_embedder.linear_2.weight", w_t2, false, d, d, d);
    reg("adaln_single.emb.timestep_embedder.linear_2.bias", b_t2, true, 1, d, d);
    reg("adaln_single.linear.weight", w_ada, false, 6 * d, d, d);
    reg("adaln_single.linear.bias", b_ada, true, 1, 6 * d, 6 * d);
    reg("caption_projection.linear_1.weight", w_c1, false, d, CAP, CAP);
    reg("caption_projection.linear_1.bias", b_c1, true, 1, d, d);
    reg("caption_projection.linear_2.weight", w_c2, false, d, d, d);
    reg("caption_projection.linear_2.bias", b_c2, true, 1, d, d);
    reg("scale_shift_table", sst_out, true, 2, d, d);
    reg("proj_out.weight", w_proj, false, P * P * COUT, d, d);
    reg("proj_out.bias", b_proj, true, 1, P * P * COUT, P * P * COUT);
    if (cfg.use_additional_conditions) {
      // PixArtAlphaCombinedTimestepSizeEmbeddings [ext]: two TimestepEmbedding(256 -> D/3) MLPs
      const char* nm[2] = {"resolution_embedder", "aspect_ratio_embedder"};
      for (int k = 0; k < 2; ++k) {
        w_sz1[k] = walloc<T>(E * 256); b_sz1[k] = walloc<float>(E);
        w_sz2[k] = walloc<T>(E * E); b_sz2[k] = walloc<float>(E);
        const std::string q = std::string("adaln_single.emb.") + nm[k] + ".";
        reg(q + "linear_1.weight", w_sz1[k], false, E, 256, 256);
        reg(q + "linear_1.bias", b_sz1[k], true, 1, E, E);
        reg(q + "linear_2.weight", w_sz2[k], false, E, E, E);
        reg(q + "linear_2.bias", b_sz2[k], true, 1, E, E);
      }
    }
    if (warena_used > warena_bytes) return fail(ECAD_E_INVALID, "internal: weight arena overflow");
    st.weight_bytes = (int64_t)warena_used;

    // ---- workspaces, sized for the largest batch
    const size_t Bm = cfg.max_batch, Nm = cfg.max_tokens, Tm = cfg.max_text_tokens;
    const size_t M = Bm * Nm, Mt = Bm * Tm;
    const size_t lat = Bm * (size_t)(CIN > COUT ? CIN : COUT) * Nm * P * P;
    const size_t max_steps = 64;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_h = take(M * d * sizeof(T)), o_xn = take(M * d * sizeof(T)), o_qkv = take(M * 3 * d * sizeof(T));
    const size_t o_ao = take(M * d * sizeof(T)), o_qc = take(M * d * sizeof(T)), o_ff = take(M * 4 * d * sizeof(T));
    const size_t o_cm = take(Mt * d * sizeof(T)), o_ctx = take(Mt * d * sizeof(T)), o_kv2 = take(Mt * (size_t)L * 2 * d * sizeof(T));
    const size_t o_cache = take((size_t)L * 3 * M * d * sizeof(T));
    const size_t o_proj = take(M * PROJ_PAD * sizeof(T)), o_min = take(lat * sizeof(T)), o_noise = take(lat * sizeof(T));
    const size_t o_enc = take(Mt * CAP * sizeof(T));
    const size_t trows = max_steps * Bm;
    const size_t o_tsin = take(trows * 256 * sizeof(T)), o_te1 = take(trows * d * sizeof(T)), o_te = take(trows * d * sizeof(T));
    const size_t o_tes = take(trows * d * sizeof(T)), o_mod = take(trows * 6 * d * sizeof(T));
    const size_t o_bias = take(Mt * sizeof(float)), o_x0 = take(lat * sizeof(float)), o_lat = take(lat * sizeof(float));
    const size_t o_ts = take(trows * sizeof(float)), o_coef = take(max_steps * 8 * sizeof(float)), o_mask = take(Mt);
    const size_t o_cond = take(Bm * 3 * sizeof(float)), o_szemb = take(Bm * d * sizeof(T));
    const size_t o_comb = take(max_steps * (size_t)L * 4 * d * sizeof(T));
    const size_t o_hstat = take((d / 144 + 1) * M * 2 * sizeof(float));
    const size_t o_sbias = take(M * sizeof(float)), o_ebias = take(Mt * sizeof(float));
    const size_t o_x02 = take(lat * sizeof(float)), o_lat2 = take(lat * sizeof(float));
    splitk_rows = 4096;  // split-K partial sums of a few-rows FF2 (launch_pg): slices x rows <= 4096 row-slices of D fp32 (18.9 MB)
    const size_t o_splitk = take(splitk_rows * d * sizeof(float));
    ws_bytes = off;
    HIPCHK(hipMalloc((void**)&ws, ws_bytes));
    HIPCHK(hipMemset(ws, 0, ws_bytes));
    h = (T*)(ws + o_h); xn = (T*)(ws + o_xn); qkv = (T*)(ws + o_qkv); attn_o = (T*)(ws + o_ao); qc = (T*)(ws + o_qc); ffmid = (T*)(ws + o_ff);
    ctx_mid = (T*)(ws + o_cm); ctx = (T*)(ws + o_ctx); kv2 = (T*)(ws + o_kv2); cache = (T*)(ws + o_cache); proj = (T*)(ws + o_proj);
    model_in = (T*)(ws + o_min); noise = (T*)(ws + o_noise); enc_buf = (T*)(ws + o_enc);
    tsin = (T*)(ws + o_tsin); temb1 = (T*)(ws + o_te1); temb = (T*)(ws + o_te); temb_silu = (T*)(ws + o_tes); mod_all = (T*)(ws + o_mod);
    bias_buf = (float*)(ws + o_bias); x0_prev = (float*)(ws + o_x0); lat_buf = (float*)(ws + o_lat); ts_dev = (float*)(ws + o_ts);
    coef_dev = (float*)(ws + o_coef); mask_buf = (uint8_t*)(ws + o_mask);
    cond_dev = (float*)(ws + o_cond); size_emb = (T*)(ws + o_szemb);
    comb = (T*)(ws + o_comb);
    hstat = (float*)(ws + o_hstat);
    self_bias_buf = (float*)(ws + o_sbias); enc_bias_user = (float*)(ws + o_ebias);
    x0_prev2 = (float*)(ws + o_x02); lat_buf2 = (float*)(ws + o_lat2);
    splitk_ws = (float*)(ws + o_splitk);
    st.workspace_bytes = (int64_t)ws_bytes;
    st.cache_arena_bytes = (int64_t)((size_t)L * 3 * M * d * sizeof(T));

    // ---- kernels that need > 64 KiB dynamic LDS would be registered here
    return set_kernel_attrs();
  }

  int set_kernel_attrs() {
    constexpr int lds = gemm_lds_bytes<128, 128>();
    HIPCHK(hipFuncSetAttribute((const void*)gemm_kernel<T, 128, 128, 2, 2, EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHK(hipFuncSetAttribute((const void*)gemm_kernel<T, 128, 128, 2, 2, EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHK(hipFuncSetAttribute((const void*)gemm_kernel<T, 128, 128, 2, 2, EPI_BIAS_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIPCHK(hipFuncSetAttribute((const void*)gemm_kernel<T, 128, 128, 2, 2, EPI_BIAS_CACHE_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return ECAD_OK;
  }

  // -------------------------------------------------------------------------------------------
  int load_weight(const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) override {
    HIPCHK(hipSetDevice(dev));
    if (weights_shared) return fail(ECAD_E_INVALID, "this context borrows its weights (ecad_share_weights): load them into the engine that owns them");
    auto it = dests.find(name);
    if (it == dests.end()) {
      if (strcmp(name, "pos_embed.pos_embed") == 0) return ECAD_OK;  // fixed table, recomputed here
      return fail(ECAD_E_INVALID, "unknown weight name '%s'", name);
    }
    Dest& d = it->second;
    int64_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    if (numel != d.rows * d.cols) return fail(ECAD_E_INVALID, "weight '%s': expected %lld elements, got %lld", name, (long long)(d.rows * d.cols), (long long)numel);
    // a Linear weight must arrive as [out_features, in_features]: a transposed tensor has the right element count and the wrong meaning
    if (ndim == 2 && d.rows > 1 && (shape[0] != d.rows || shape[1] != d.cols))
      return fail(ECAD_E_INVALID, "weight '%s': expected shape [%lld, %lld], got [%lld, %lld]", name, (long long)d.rows, (long long)d.cols, (long long)shape[0], (long long)shape[1]);
    size_t esz = dtype == ECAD_DTYPE_F32 ? 4 : 2;
    if (dtype != ECAD_DTYPE_F16 && dtype != ECAD_DTYPE_BF16 && dtype != ECAD_DTYPE_F32) return fail(ECAD_E_INVALID, "weight '%s': bad dtype %d", name, dtype);
    void* stage = nullptr;
    HIPCHK(hipMalloc(&stage, numel * esz));
    hipError_t e = hipMemcpy(stage, ptr, numel * esz, hipMemcpyDefault);
    if (e != hipSuccess) { (void)hipFree(stage); return fail(ECAD_E_HIP, "hipMemcpy of weight '%s' failed: %s", name, hipGetErrorString(e)); }
    const int grid = (int)((numel + 255) / 256 < 4096 ? (numel + 255) / 256 : 4096);
#define CONV(SRC, DST) hipLaunchKernelGGL((convert_kernel<SRC, DST>), dim3(grid), dim3(256), 0, 0, (const SRC*)stage, (DST*)d.ptr, (size_t)numel)
    if (d.f32) {
      if (dtype == ECAD_DTYPE_F16) CONV(f16_t, float); else if (dtype == ECAD_DTYPE_BF16) CONV(bf16_t, float); else CONV(float, float);
    } else {
      if (dtype == ECAD_DTYPE_F16) CONV(f16_t, T); else if (dtype == ECAD_DTYPE_BF16) CONV(bf16_t, T); else CONV(float, T);
    }
#undef CONV
    e = hipDeviceSynchronize();
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(ECAD_E_HIP, "weight conversion failed: %s", hipGetErrorString(e));
    d.loaded = true;
    finalized = false;
    sarena->finalized = false; ++sarena->gen;  // contexts that borrow this arena re-lower before their next run (weights_current)
    return ECAD_OK;
  }

  int finalize_weights() override {
    for (auto& kv : dests)
      if (!kv.second.loaded) return fail(ECAD_E_NOT_READY, "weight '%s' was never loaded", kv.first.c_str());
    finalized = true;
    drop_all_plans();
    sarena->fp8 = false;
    if (fp8) { int rc = build_fp8_weights(); if (rc) return rc; }  // (publishes the copies in the arena)
    sarena->finalized = true; seen_gen = ++sarena->gen;
    return ECAD_OK;
  }
  // a context that borrows its weights: has the owner changed them since this context last lowered its plans?
  int weights_current() {
    if (!weights_shared) return ECAD_OK;
    if (!sarena->finalized) { finalized = false; return fail(ECAD_E_NOT_READY, "the engine this context borrows its weights from is loading new ones: call ecad_finalize_weights on it first"); }
    finalized = true;
    if (seen_gen != sarena->gen) {
      if (graph_exec || graph || !plan_lru.empty()) (void)hipDeviceSynchronize();  // a graph of the old weights may still be in flight
      drop_all_plans();
      fold_release_current();
      batch_begun = false;
      std::fill(valid_rt.begin(), valid_rt.end(), 0);  // cache slots computed with the donor's previous weights are not usable (a keep_caches rebind must not keep them)
      seen_gen = sarena->gen;
      fp8 = sarena->fp8;  // the owner's mode and its (re-made) e4m3 copies
      w8map.clear();
      if (fp8) { w8map = sarena->w8; int rc = fp8_workspaces(); if (rc) return rc; }
    }
    return ECAD_OK;
  }
  // Second (third ...) execution context on the same GPU: this engine keeps its own workspaces, cache arena, plans and hipGraph but reads the weights of
  // `src` -- same configuration, so the arena was carved up in the same order and every pointer is the donor's base plus the same offset.
  int share_weights(EngineBase* src_) override {
    auto* src = dynamic_cast<EngineT<T>*>(src_);
    if (!src || src == this) return fail(ECAD_E_INVALID, "share_weights: the donor must be another engine of the same dtype");
    if (!src->finalized) return fail(ECAD_E_NOT_READY, "share_weights: the donor's weights are not finalized");
    if (src->dev != dev || src->L != L || src->D != D || src->CAP != CAP || src->CIN != CIN || src->COUT != COUT || src->P != P ||
        src->cfg.use_additional_conditions != cfg.use_additional_conditions || src->warena_bytes != warena_bytes || src->warena_used != warena_used)
      return fail(ECAD_E_INVALID, "share_weights: the two engines were created with different model configurations (or on different devices)");
    if (fp8 && !src->fp8) return fail(ECAD_E_UNSUPPORTED, "share_weights: this engine is in fp8 mode and the donor is not; the e4m3 weight copies are the donor's");
    if (weights_shared) return fail(ECAD_E_INVALID, "share_weights: this engine already borrows its weights");
    HIPCHK(hipSetDevice(dev));
    drop_all_plans();
    const ptrdiff_t delta = src->warena - warena;
    auto rb = [&](auto*& ptr) { if (ptr) ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(reinterpret_cast<char*>(ptr) + delta); };
    for (auto& b : blk) { rb(b.w_qkv); rb(b.w_o1); rb(b.w_q2); rb(b.w_o2); rb(b.w_ff1); rb(b.w_ff2); rb(b.b_qkv); rb(b.b_o1); rb(b.b_q2); rb(b.b_o2); rb(b.b_ff1); rb(b.b_ff2); rb(b.sst); }
    rb(w_kv2_all); rb(w_patch); rb(w_t1); rb(w_t2); rb(w_ada); rb(w_c1); rb(w_c2); rb(w_proj);
    rb(b_kv2_all); rb(b_patch); rb(b_t1); rb(b_t2); rb(b_ada); rb(b_c1); rb(b_c2); rb(b_proj); rb(sst_out);
    for (int k = 0; k < 2; ++k) { rb(w_sz1[k]); rb(w_sz2[k]); rb(b_sz1[k]); rb(b_sz2[k]); }
    for (auto& kv : dests) { kv.second.ptr = reinterpret_cast<char*>(kv.second.ptr) + delta; kv.second.loaded = true; }
    (void)hipFree(warena);
    for (void* q : sarena->w8allocs) (void)hipFree(q);  // e4m3 copies this engine made of its own weights before it became a context
    delete sarena;
    warena = src->warena;
    sarena = src->sarena;
    ++sarena->refs;
    seen_gen = sarena->gen;
    weights_shared = true;
    finalized = true;
    // fp8 mode travels with the weights: the context multiplies the donor's e4m3 copies (the keys of its map are the 16-bit weight pointers, now this engine's too)
    w8map.clear();
    fp8 = sarena->fp8;
    if (fp8) {
      w8map = sarena->w8;
      return fp8_workspaces();
    }
    return ECAD_OK;
  }
  // LayerNorm fold on / off at run time: a caller that runs every schedule for ONE batch (the NSGA-II evaluation) cannot amortise the
  // per-schedule preparation of the pre-scaled weights; plan sets are keyed by this switch, so both kinds coexist in the plan cache
  int set_ln_fold(int on) override {
    if ((on != 0) != ln_fold) { stash_current(); ln_fold = on != 0; }
    return ECAD_OK;
  }
  // How many image batches the caller keeps in flight on this GPU (execution contexts): the few-rows rule counts the other batches' tiles
  // as well -- latency tiles fill an idle chip, but with four batches of 4 images in flight they cost a quarter of the throughput (140 against 192 images/s).
  // Plan sets are keyed by it (the captured graphs hold different kernels).
  int set_batches_in_flight(int n) override {
    if (n < 1 || n > 64) return fail(ECAD_E_INVALID, "batches in flight: %d", n);
    if (n != in_flight) { stash_current(); in_flight = n; }
    return ECAD_OK;
  }
  // ECAD_SPLIT_K as a call (round-5 ADVICE): the choice is made where the launch is issued, i.e. it is baked into a captured graph -- plan sets are keyed by it
  int set_split_k(int on) override {
    if ((on != 0) != split_k) { stash_current(); split_k = on != 0; }
    return ECAD_OK;
  }
  // Native variants of the compute plug-ins (row a6; cached_transformer_block.py:125-165): per (step, block, component) one of ECAD_NATIVE_* -- gate rules the engine
  // lowers into its own kernels, so a schedule that names them stays in the whole-loop plan and its hipGraph (a host callback cannot).  NULL: all default.
  int set_native(const uint8_t* kinds, int nsteps, int nblocks) override {
    if (!kinds) {
      if (!native_kind.empty()) { stash_current(); native_kind.clear(); resolve_executed(); plan_valid = false; batch_begun = false; }
      return ECAD_OK;
    }
    if (nblocks != L || nsteps != steps || bits.empty()) return fail(ECAD_E_INVALID, "native variants are given per (step, block, component) of the schedule that is set: %d x %d, got %d x %d", steps, L, nsteps, nblocks);
    bool any = false;
    for (size_t i = 0; i < (size_t)nsteps * L * 3; ++i) {
      if (kinds[i] > ECAD_NATIVE_CFG_MEAN_CACHE) return fail(ECAD_E_INVALID, "native variant %d unknown", (int)kinds[i]);
      if (kinds[i] == ECAD_NATIVE_CFG_MEAN_CACHE && i % 3 != 1) return fail(ECAD_E_UNSUPPORTED, "the CFG-mean cache rule exists for attn2 (the component TGATE applies it to, cached_transformer_block.py:443-452)");
      any = any || kinds[i];
    }
    stash_current();
    native_kind.clear();
    if (any) native_kind.assign(kinds, kinds + (size_t)nsteps * L * 3);
    resolve_executed();
    plan_valid = false;
    batch_begun = false;
    return ECAD_OK;
  }
  int set_fp8(int on) override {
    if (on && weights_shared) return fail(ECAD_E_UNSUPPORTED, "fp8 weight copies are per engine: not on a context that borrows its weights");
    if ((on != 0) != fp8) { fp8 = on != 0; finalized = false; drop_all_plans(); }
    return ECAD_OK;
  }
  // added_cond_kwargs of forward (pixart_transformer_2d_edited.py:208-211): host arrays, copied here; read by the size-embedding
  // kernel at the start of every image batch (the captured graph reads the same device buffer)
  int set_added_cond(const float* resolution, const float* aspect_ratio, int Bt_) override {
    if (!cfg.use_additional_conditions) return fail(ECAD_E_INVALID, "engine was created without use_additional_conditions");
    if (!resolution || !aspect_ratio) return fail(ECAD_E_INVALID, "`added_cond_kwargs` cannot be None when using additional conditions for `adaln_single`.");
    if (Bt_ < 1 || Bt_ > cfg.max_batch) return fail(ECAD_E_INVALID, "batch %d outside [1, max_batch=%d]", Bt_, cfg.max_batch);
    HIPCHK(hipSetDevice(dev));
    cond_host.resize((size_t)Bt_ * 3);
    cond_uniform = true;
    for (int b = 0; b < Bt_; ++b) {
      cond_host[b * 3 + 0] = resolution[b * 2 + 0]; cond_host[b * 3 + 1] = resolution[b * 2 + 1]; cond_host[b * 3 + 2] = aspect_ratio[b];
      for (int k = 0; k < 3; ++k) cond_uniform = cond_uniform && cond_host[b * 3 + k] == cond_host[k];
    }
    HIPCHK(hipMemcpy(cond_dev, cond_host.data(), cond_host.size() * sizeof(float), hipMemcpyHostToDevice));
    cond_bt = Bt_;
    return ECAD_OK;
  }
  // attention_mask / encoder_attention_mask of forward when they arrive as additive biases (or, for self-attention, at all): the reference turns a 2-D 0/1
  // mask into (1 - m) * -10000 and adds a 3-D [B, 1, keys] bias as given (pixart_transformer_2d_edited.py:271-291); the host does the first, this the second.
  // Per (sample, key) floats on the device, copied on `st`; NULL switches that bias off.  Plans are keyed by which biases are on: with a self-attention bias
  // attn1 runs as Q/K/V projections + the masked stand-alone attention kernel (the fused tails take no key bias on self-attention).
  int set_attention_bias(const float* self_bias, const float* enc_bias, int Bt_, int N_, int T_, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (Bt_ < 1 || Bt_ > cfg.max_batch) return fail(ECAD_E_INVALID, "batch %d outside [1, max_batch=%d]", Bt_, cfg.max_batch);
    if (self_bias && (N_ < 1 || N_ > cfg.max_tokens)) return fail(ECAD_E_INVALID, "self-attention bias: %d keys outside [1, max_tokens=%d]", N_, cfg.max_tokens);
    if (enc_bias && (T_ < 1 || T_ > cfg.max_text_tokens)) return fail(ECAD_E_INVALID, "encoder attention bias: %d keys outside [1, max_text_tokens=%d]", T_, cfg.max_text_tokens);
    if ((self_bias != nullptr) != self_bias_on || (enc_bias != nullptr) != enc_bias_on) {
      stash_current();
      self_bias_on = self_bias != nullptr; enc_bias_on = enc_bias != nullptr;
      batch_begun = false;
    }
    if (self_bias) HIPCHK(hipMemcpyAsync(self_bias_buf, self_bias, (size_t)Bt_ * N_ * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (enc_bias) HIPCHK(hipMemcpyAsync(enc_bias_user, enc_bias, (size_t)Bt_ * T_ * sizeof(float), hipMemcpyDeviceToDevice, s));
    bias_bt = Bt_; bias_n = self_bias ? N_ : 0; bias_t = enc_bias ? T_ : 0;  // the rows are strided by these: begin_batch refuses another geometry
    return ECAD_OK;
  }
  int quant_weight(const T* w, size_t rows, int K) {
    W8 e{};
    HIPCHK(hipMalloc((void**)&e.w, rows * K));
    HIPCHK(hipMalloc((void**)&e.s, rows * sizeof(float)));
    sarena->w8allocs.push_back(e.w); sarena->w8allocs.push_back(e.s);
    launch_quant_rows_fp8<T>(w, (int64_t)K, K, (const T*)nullptr, (int64_t)0, 0, e.w, (int64_t)K, e.s, (int)rows, (hipStream_t)0);
    w8map[w] = e;
    return ECAD_OK;
  }
  int build_fp8_weights() {
    if (D % 128) return fail(ECAD_E_UNSUPPORTED, "fp8 mode needs D %% 128 == 0");
    if (!sarena->w8allocs.empty()) (void)hipDeviceSynchronize();  // a borrower's graph may still multiply the copies that are about to go
    for (void* q : sarena->w8allocs) (void)hipFree(q);
    sarena->w8allocs.clear(); sarena->w8.clear(); w8map.clear();
    const size_t d = D;
    int rc;
    for (auto& b : blk) {
      if ((rc = quant_weight(b.w_qkv, 3 * d, D)) || (rc = quant_weight(b.w_o1, d, D)) || (rc = quant_weight(b.w_q2, d, D)) || (rc = quant_weight(b.w_o2, d, D)) ||
          (rc = quant_weight(b.w_ff1, 4 * d, D)) || (rc = quant_weight(b.w_ff2, d, 4 * D)))
        return rc;
      const W8 e = w8map[b.w_qkv];
      w8map[b.w_qkv + d * d] = W8{e.w + d * d, e.s + d};  // the K,V rows of the fused weight (split QKV launch)
    }
    HIPCHK(hipDeviceSynchronize());
    sarena->w8 = w8map; sarena->fp8 = true;  // contexts that borrow this arena run their fp8 GEMMs on these copies (share_weights / weights_current)
    return fp8_workspaces();
  }
  // a8 / sa8 (the quantised A rows of the GEMM in flight) and the kernel attributes: per engine, also for a context that borrows the e4m3 weight copies
  int fp8_workspaces() {
    const size_t d = D;
    if (!a8) {
      const size_t Mmax = (size_t)cfg.max_batch * cfg.max_tokens;
      HIPCHK(hipMalloc((void**)&a8, Mmax * 4 * d));
      HIPCHK(hipMalloc((void**)&sa8, Mmax * sizeof(float)));
      HIPCHK(hipMalloc((void**)&a8b, Mmax * 4 * d));
      HIPCHK(hipMalloc((void**)&e8buf, Mmax * 16));
    }
    if (!f8_attr) {
      HIPCHK(hipFuncSetAttribute((const void*)gemm3s8_kernel<T, EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
      HIPCHK(hipFuncSetAttribute((const void*)gemm3s8_kernel<T, EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
      HIPCHK(hipFuncSetAttribute((const void*)gemm3s8_kernel<T, EPI_BIAS_CACHE_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
      HIPCHK(hipFuncSetAttribute((const void*)gemm2w8_kernel<T, EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
      HIPCHK(hipFuncSetAttribute((const void*)gemm2w8_kernel<T, EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
      HIPCHK(hipFuncSetAttribute((const void*)gemm2w8_kernel<T, EPI_BIAS_GELU, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
      HIPCHK(hipFuncSetAttribute((const void*)gemm3s8_kernel<T, EPI_BIAS_CACHE_RESID, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES + G3S8_TAB_BYTES));
      f8_attr = true;
    }
    st.weight_bytes = (int64_t)warena_used + (int64_t)blk.size() * 14 * (int64_t)d * d;
    return ECAD_OK;
  }
  // append a block GEMM; fp8 mode: one row-quantiser pass over its A operand (skipped when a8 already holds that tensor), then gemm3s8
  void push_gemm(std::vector<Op<T>>& ops, Op<T> o, bool share_prev = false) {
    auto it = fp8 ? w8map.find(o.g.W) : w8map.end();
    if (it == w8map.end() || o.g.K % GEMM8_BK || o.g.N % G3_BN || (o.epi != EPI_BIAS && o.epi != EPI_BIAS_GELU && o.epi != EPI_BIAS_CACHE_RESID)) {
      ops.push_back(o);
      a8_src = nullptr;  // a bf16 GEMM may have rewritten the tensor a8 mirrors (e.g. the residual stream)
      a8b_src = nullptr;
      return;
    }
    // FF2 right behind an FF1 that left its output as e4m3 + per-tile block scales (below): no quantiser pass, the MFMAs take the scale bytes
    const bool block_scaled = a8b_src && a8b_src == o.g.A && o.cls == KC_GEMM_FF2 && o.epi == EPI_BIAS_CACHE_RESID && o.g.lda == o.g.K;
    if (!block_scaled && !(share_prev && a8_src == o.g.A)) {  // share_prev: the previous push quantised the same, untouched tensor (Q then K,V of one LN output)
      Op<T> qo;
      qo.kind = Op<T>::QUANT; qo.cls = KC_QUANT; qo.p0 = o.g.A; qo.i0 = o.g.lda; qo.g.M = o.g.M; qo.g.K = o.g.K;
      qo.bytes = (double)o.g.M * o.g.K * (2 * sizeof(T) + 1);
      ops.push_back(qo);
    }
    if (!block_scaled) a8_src = o.epi == EPI_BIAS_CACHE_RESID ? nullptr : o.g.A;  // the residual epilogue rewrites h, which a8 may mirror
    a8b_src = nullptr;
    o.i0 = o.i1 = 0;
    if (block_scaled) o.i1 = 1;
    // FF1 + GELU on whole rounds of 256 x 288 tiles (what launch_op gives gemm2w8_kernel): its only reader is FF2 -- emit e4m3 + one E8M0 byte per (row, tile)
    if (fp8_seg && o.cls == KC_GEMM_FF1 && o.epi == EPI_BIAS_GELU && o.g.C == ffmid && o.g.ldc == o.g.N && o.g.M % G2_BM == 0 && o.g.N % G2_BN == 0 && o.g.N / G2_BN == 16 &&
        (o.g.M / G2_BM) * (o.g.N / G2_BN) >= num_cus) {
      o.i0 = 1;
      a8b_src = o.g.C;
    }
    o.kind = Op<T>::GEMM8;
    o.w8 = it->second.w; o.sw8 = it->second.s;
    o.bytes = ((double)o.g.M * o.g.K + (double)o.g.N * o.g.K) + (double)o.g.M * o.g.N * sizeof(T);
    ops.push_back(o);
  }

  // 2-D sincos table exactly as diffusers' PatchEmbed builds it [ext]: first half of the channels encodes the column.
  int build_pos_table(int gh, int gw) {
    if (gh * gw > cfg.max_tokens) return fail(ECAD_E_INVALID, "token count %d exceeds max_tokens %d", gh * gw, cfg.max_tokens);
    const int base = cfg.sample_size / P;
    float interp = cfg.interpolation_scale > 0 ? cfg.interpolation_scale : (float)(cfg.sample_size / 64 > 1 ? cfg.sample_size / 64 : 1);
    std::vector<float> tab((size_t)gh * gw * D);
    const int half = D / 2, quarter = half / 2;
    for (int i = 0; i < gh; ++i)
      for (int j = 0; j < gw; ++j) {
        // diffusers: arange(grid)/ (grid/base_size) / interpolation_scale, float32
        const float cj = (float)j / ((float)gw / (float)base) / interp;
        const float ci = (float)i / ((float)gh / (float)base) / interp;
        float* row = &tab[((size_t)i * gw + j) * D];
        for (int k = 0; k < quarter; ++k) {
          const double omega = 1.0 / pow(10000.0, (double)k / (double)quarter);
          row[k] = (float)sin((double)cj * omega);
          row[quarter + k] = (float)cos((double)cj * omega);
          row[half + k] = (float)sin((double)ci * omega);
          row[half + quarter + k] = (float)cos((double)ci * omega);
        }
      }
    HIPCHK(hipMemcpy(pos, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    pos_grid = gh * 100000 + gw;
    return ECAD_OK;
  }

  // -------------------------------------------------------------------------------------------
  // plan-set cache
  void free_stash(PlanStash& e) {
    if (e.ge) (void)hipGraphExecDestroy(e.ge);
    if (e.g) (void)hipGraphDestroy(e.g);
    if (e.dd) (void)hipFree(e.dd);
    if (e.fold.base) fold_pool.push_back(e.fold);  // recycled by the next build_plans, freed with the engine
    e.ge = nullptr; e.g = nullptr; e.dd = nullptr; e.fold = FoldArena{};
  }
  // an arena of at least `need` bytes: the smallest pooled one that fits, else a new allocation (the largest pooled one goes first, so the pool does not grow)
  size_t fold_held() const {
    size_t t = fold_cur.cap;
    for (auto& a : fold_pool) t += a.cap;
    for (auto& e : plan_lru) t += e.fold.cap;
    return t;
  }
  // false: no arena (over the budget, or the device is out of memory) -- the caller lowers the plan set without the fold
  bool fold_acquire(size_t need) {
    int best = -1;
    for (int i = 0; i < (int)fold_pool.size(); ++i)
      if (fold_pool[i].cap >= need && (best < 0 || fold_pool[i].cap < fold_pool[best].cap)) best = i;
    if (best >= 0) { fold_cur = fold_pool[best]; fold_pool.erase(fold_pool.begin() + best); return true; }
    fold_cur = FoldArena{};
    const size_t cap = need + (need >> 3);
    // nothing pooled fits: pooled arenas go (largest first) until the new one fits the budget; parked plan sets keep theirs
    for (bool first = true; !fold_pool.empty() && (first || fold_held() + cap > fold_budget); first = false) {  // (one always goes, so that the pool does not grow)
      int big = 0;
      for (int i = 1; i < (int)fold_pool.size(); ++i) if (fold_pool[i].cap > fold_pool[big].cap) big = i;
      (void)hipFree(fold_pool[big].base);
      fold_pool.erase(fold_pool.begin() + big);
    }
    if (fold_held() + cap > fold_budget) return false;
    if (hipMalloc((void**)&fold_cur.base, cap) != hipSuccess) {
      (void)hipGetLastError();  // out of device memory is not an error of this call: the plan set runs unfolded
      for (auto& a : fold_pool) (void)hipFree(a.base);
      fold_pool.clear();
      fold_cur = FoldArena{};
      return false;
    }
    fold_cur.cap = cap;
    return true;
  }
  void fold_release_current() {
    if (fold_cur.base) fold_pool.push_back(fold_cur);
    fold_cur = FoldArena{};
    fold_descs.clear(); fold_used = 0; fold_built = false; fold_prep_ops = 0; fold_prep_begin = 0; ln_folded = 0; fold_ts.clear();
  }
  std::string make_key() const {
    std::string k((const char*)bits.data(), bits.size());
    k.append((const char*)custom_mask.data(), custom_mask.size());
    k.append((const char*)native_kind.data(), native_kind.size());
    const int geo[10] = {steps, Bt, Hl, Wl, Tt, fp8 ? 1 : 0, fuse_attn ? 1 : 0, (ln_fold ? 1 : 0) | (split_k ? 2 : 0), in_flight,
                         (self_bias_on ? 1 : 0) | (enc_bias_on ? 2 : 0)};
    k.append((const char*)geo, sizeof geo);
    return k;
  }
  // move the live plan set (if any) to the front of the LRU; evict beyond the capacity (device sync first: an evicted graph may still be in flight)
  void stash_current() {
    if (!plan_valid) return;
    plan_valid = false;
    if (plan_lru_cap == 0) return;  // build_plans tears the stale set down
    PlanStash e;
    e.key = std::move(cur_key);
    e.begin = std::move(plan_begin); e.run = std::move(plan_run); e.step = std::move(plan_step); e.dh = std::move(desc_host);
    e.dd = desc_dev; e.dcap = desc_cap; e.g = graph; e.ge = graph_exec; e.gg = graph_guidance; e.st = st; e.comb = comb_live;
    e.fold = fold_cur; e.fdescs = std::move(fold_descs); e.fts = std::move(fold_ts); e.fused = fold_used; e.fprep = fold_prep_ops; e.fbegin = fold_prep_begin;
    e.fbuilt = fold_built; e.folded = ln_folded;
    fold_cur = FoldArena{}; fold_descs.clear(); fold_ts.clear(); fold_used = 0; fold_built = false; fold_prep_ops = 0; fold_prep_begin = 0; ln_folded = 0;
    desc_dev = nullptr; desc_cap = 0; graph = nullptr; graph_exec = nullptr;
    plan_begin.clear(); plan_run.clear(); plan_step.clear(); desc_host.clear(); cur_key.clear();
    plan_lru.push_front(std::move(e));
    if (plan_lru.size() > plan_lru_cap) {
      (void)hipDeviceSynchronize();  // an evicted graph may still be in flight; a device error here resurfaces at the next checked call
      while (plan_lru.size() > plan_lru_cap) { free_stash(plan_lru.back()); plan_lru.pop_back(); }
    }
  }
  bool restore_plans(const std::string& key) {
    for (auto it = plan_lru.begin(); it != plan_lru.end(); ++it) {
      if (it->key != key) continue;
      if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
      if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
      if (desc_dev) { (void)hipFree(desc_dev); desc_dev = nullptr; }
      PlanStash& e = *it;
      plan_begin = std::move(e.begin); plan_run = std::move(e.run); plan_step = std::move(e.step); desc_host = std::move(e.dh);
      desc_dev = e.dd; desc_cap = e.dcap; graph = e.g; graph_exec = e.ge; graph_guidance = e.gg; comb_live = e.comb;
      fold_release_current();
      fold_cur = e.fold; fold_descs = std::move(e.fdescs); fold_ts = std::move(e.fts); fold_used = e.fused; fold_prep_ops = e.fprep; fold_prep_begin = e.fbegin;
      fold_built = e.fbuilt; ln_folded = e.folded;
      e.fold = FoldArena{};
      st = e.st;
      cur_key = key;
      plan_lru.erase(it);
      plan_valid = true;
      return true;
    }
    return false;
  }
  void drop_all_plans() {
    plan_valid = false;
    if (!plan_lru.empty()) (void)hipDeviceSynchronize();
    for (auto& e : plan_lru) free_stash(e);
    plan_lru.clear();
  }

  // -------------------------------------------------------------------------------------------
  int set_schedule(const uint8_t* b, int nsteps, int nblocks) override {
    if (nblocks != L) return fail(ECAD_E_INVALID, "schedule has %d blocks, model has %d", nblocks, L);
    if (nsteps < 1 || nsteps > 64) return fail(ECAD_E_INVALID, "num_inference_steps %d out of range [1, 64]", nsteps);
    stash_current();  // the plans and graph of the outgoing schedule stay reachable by content
    steps = nsteps;
    bits.assign(b, b + (size_t)nsteps * L * 3);
    native_kind.clear();  // variants belong to the schedule they were given for (ecad_set_native_variants comes after ecad_set_schedule)
    resolve_executed();
    plan_valid = false;
    batch_begun = false;
    return ECAD_OK;
  }
  std::vector<uint8_t> native_kind;  // [steps][L][3] ECAD_NATIVE_* or empty (all default)
  int native_at(int s, int i) const { return native_kind.empty() ? 0 : native_kind[(size_t)s * L * 3 + i]; }
  // the gate's flag of (step s, component i of L * 3): the schedule bit, or "always" for a component that names the recompute variant
  bool flag_at(int s, int i) const { return bits[(size_t)s * L * 3 + i] != 0 || native_at(s, i) == ECAD_NATIVE_RECOMPUTE; }
  void resolve_executed() {
    executed.assign(bits.size(), 0);
    std::vector<uint8_t> have((size_t)L * 3, 0);
    st.components_executed = st.components_cached = 0;
    for (int s = 0; s < steps; ++s)
      for (int i = 0; i < L * 3; ++i) {
        const size_t k = (size_t)s * L * 3 + i;
        executed[k] = flag_at(s, i) || !have[i];  // recompute or cache empty (cached_transformer_block.py:340-347)
        have[i] |= executed[k];
        if (executed[k]) st.components_executed++; else st.components_cached++;
      }
  }

  // -------------------------------------------------------------------------------------------
  // plan construction
  Op<T> mk_gemm(int cls, int epi, const T* A, const T* W, const float* bias, T* C, int M, int Nn, int K, int lda, int ldc) {
    Op<T> o;
    o.kind = Op<T>::GEMM; o.cls = cls; o.epi = epi;
    o.g.A = A; o.g.W = W; o.g.bias = bias; o.g.C = C; o.g.resid = nullptr; o.g.gate_table = nullptr; o.g.gate_mod = nullptr;
    o.g.gate_mod_stride = 0; o.g.rows_per_batch = N > 0 ? N : 1; o.g.M = M; o.g.N = Nn; o.g.K = K; o.g.lda = lda; o.g.ldw = K; o.g.ldc = ldc; o.g.ldr = ldc;
    o.g.variant = 1; o.g.epi_direct = 1;
    o.flops = 2.0 * M * (double)Nn * K;
    o.bytes = ((double)M * K + (double)Nn * K + (double)M * Nn) * sizeof(T);
    return o;
  }
  Op<T> mk_gemm_resid(int cls, const T* A, const T* W, const float* bias, T* cache_slot, int M, int K, const float* gate_table, const T* gate_mod, int64_t mstride) {
    Op<T> o = mk_gemm(cls, EPI_BIAS_CACHE_RESID, A, W, bias, cache_slot, M, D, K, K, D);
    o.g.resid = h; o.g.ldr = D; o.g.gate_table = gate_table; o.g.gate_mod = gate_mod; o.g.gate_mod_stride = mstride;
    o.bytes += 2.0 * M * (double)D * sizeof(T);
    return o;
  }

  struct Pending { const T* cache; const float* gate_table; const T* gate_mod; };
  // A cache slot written at step s is dead when the same component executes again at step s+1 (or s is the last
  // step): nothing reads it before it is overwritten / dropped.  30 % of ours_fast's writes are dead.
  bool cache_write_live(int s, int b, int c) const { return s + 1 < steps && !executed[((size_t)(s + 1) * L + b) * 3 + c]; }

  void emit_ln(std::vector<Op<T>>& ops, std::vector<Pending>& pend, int M, const float* tsh, const float* tsc, const T* msh, const T* msc, int64_t ln_stride, int64_t mstride,
               bool feeds_fp8 = false, const T* csh = nullptr, const T* csc = nullptr) {
    if (pend.size() > 6) emit_replay(ops, pend, M, mstride);
    Op<T> o;
    o.kind = Op<T>::LN; o.cls = KC_LN;
    o.f.h = h; o.f.y = xn; o.f.srcs = nullptr; o.f.n_src = (int)pend.size(); o.f.M = M; o.f.D = D; o.f.mod_stride = mstride; o.f.ln_mod_stride = ln_stride;
    o.f.rows_per_batch = N; o.f.table_shift = tsh; o.f.table_scale = tsc; o.f.mod_shift = msh; o.f.mod_scale = msc; o.f.eps = cfg.norm_eps;
    o.f.comb_shift = csh; o.f.comb_scale = csc;
    if (!pend.empty()) {
      o.desc_off = (int)desc_host.size();
      for (auto& p : pend) desc_host.push_back(SrcDesc{p.cache, p.gate_table, p.gate_mod});
    }
    if (feeds_fp8 && fp8 && D == 1152) {  // the consumers are fp8 GEMMs: the kernel writes e4m3 + row scales straight into a8 / sa8
      o.f.y8 = a8; o.f.y8_scale = sa8;
      a8_src = xn;
    } else {
      a8_src = nullptr;
    }
    o.bytes = (2.0 + pend.size() + (pend.empty() ? 0 : 1)) * M * (double)D * sizeof(T);
    st.cache_bytes_read += (int64_t)pend.size() * M * D * (int64_t)sizeof(T);
    pend.clear();
    ops.push_back(o);
  }
  // the CFG + sigma-drop + DPM-Solver++ update between two forwards of the whole-loop plan is computed by the next forward's patch embed (P = 2, 4 latent channels:
  // what patch_embed_cols_kernel is written for) instead of a launch of its own
  bool solver_in_patch() const { return P == 2 && CIN == 4; }
  void emit_replay(std::vector<Op<T>>& ops, std::vector<Pending>& pend, int M, int64_t mstride) {
    if (pend.empty()) return;
    Op<T> o;
    // a pure cache-hit step (all 3L components replayed in one sweep, SURVEY 8d "algorithmic bytes on cache-hit steps") is its own class
    // and kernel symbol so that its HBM rate is reported separately from the short mid-step replays
    o.kind = Op<T>::REPLAY; o.cls = (int)pend.size() == 3 * L ? KC_CACHE_HIT_STEP : KC_CACHE_REPLAY;
    o.f.h = h; o.f.y = nullptr; o.f.n_src = (int)pend.size(); o.f.M = M; o.f.D = D; o.f.mod_stride = mstride; o.f.ln_mod_stride = 0; o.f.rows_per_batch = N;
    o.desc_off = (int)desc_host.size();
    for (auto& p : pend) desc_host.push_back(SrcDesc{p.cache, p.gate_table, p.gate_mod});
    o.bytes = (2.0 + pend.size()) * M * (double)D * sizeof(T);
    st.cache_bytes_read += (int64_t)pend.size() * M * D * (int64_t)sizeof(T);
    pend.clear();
    ops.push_back(o);
  }

  // ops of one transformer forward at schedule row s.  mod_s/emb_s: adaLN tables of this step.
  // `ex_rt` (ecad_step only): the components that execute in THIS call, overriding the static resolution (all cache writes are then kept)
  void emit_transformer(std::vector<Op<T>>& ops, int s, const T* mod_s, int64_t mstride, const T* emb_s, int64_t estride, bool bind_call, const uint8_t* ex_rt = nullptr) {
    const int M = Bt * N, Mt = Bt * Tt;
    // dead-write elimination belongs to the whole-loop plan, whose steps run in order by construction; ecad_step keeps every write: its caller may step out of
    // order or repeat a step, and the reference's caches are then as warm as the last forward left them (cached_transformer_block.py:355-360)
    auto live = [&](int b, int c) { return (ex_rt || bind_call) ? true : cache_write_live(s, b, c); };
    // a component routed to the user's compute plug-in (step plans only): its input lands in io_in, the callback fills io_out with the
    // sub-layer output, which then joins the pending gated sums exactly like a cached tensor would (gate applied by the engine, :243-246)
    auto custom = [&](int b, int c) { return bind_call && any_custom && custom_mask[(size_t)b * 3 + c]; };
    auto emit_custom = [&](int b, int c, const T* src) {
      Op<T> o; o.kind = Op<T>::CUSTOM; o.cls = KC_MISC; o.i0 = b; o.i1 = c; o.p0 = src; o.bytes = 2.0 * M * (double)D * sizeof(T);
      ops.push_back(o);
      a8_src = nullptr;
    };
    const size_t slot = (size_t)M * D;
    {
      Op<T> o; o.kind = Op<T>::PATCH; o.cls = KC_PATCH; o.p0 = model_in; o.call_latents = bind_call; o.bytes = (double)M * D * sizeof(T);
      o.i0 = (!bind_call && s > 0 && solver_in_patch()) ? s : 0;  // whole-loop plan: the solver update of step s - 1 is computed here (elementwise.h PatchSolve)
      ops.push_back(o);
    }
    std::vector<Pending> pend;
    const int ldkv = L * 2 * D;
    // LayerNorm folded into its consumer GEMMs: only in the shared-timestep plan (one shift / scale vector per (step, block): comb), only where the
    // LayerNorm launch would carry no pending cache replays (it then only reads h), and only while the row statistics published by the read-modify-write
    // GEMM that last wrote h are still current (`stats_ok`).
    const bool fold_plan = fold_plan_on && !bind_call && comb_live;
    bool stats_ok = false;
    // index of the read-modify-write GEMM that wrote h last, while it is still the LAST op of the list (nothing has read h since): a cached component that comes
    // next is replayed in that GEMM's epilogue (GemmArgs::xsrc, at most G3_XSRC_MAX of them) instead of joining the pending list -- the LayerNorm launch / replay
    // flush that would have carried it is not emitted, and the row statistics the GEMM publishes already describe the final h, so the LayerNorm fold goes on.
    int last_rmw = -1;
    auto push_rmw = [&](Op<T> o) {
      if (fold_plan) o.g.stat_out = hstat;
      push_gemm(ops, o);
      stats_ok = fold_plan;
      // only gemm3s_kernel / splitk_rmw_kernel implement GemmArgs::xsrc: an inner dim that is not a multiple of 144 goes to gemm_kernel<128,128> (launch_gemm),
      // whose read-modify-write epilogue knows nothing of it -- the cached components then stay on the pending list (LayerNorm launch / replay flush)
      last_rmw = (!ops.empty() && ops.back().kind == Op<T>::GEMM && xsrc_fold && ops.back().g.N % G3_BN == 0) ? (int)ops.size() - 1 : -1;
    };
    auto pend_push = [&](const Pending& q) {
      if (last_rmw >= 0 && last_rmw == (int)ops.size() - 1 && pend.empty() && ops[last_rmw].g.n_xsrc < G3_XSRC_MAX) {
        Op<T>& r = ops[last_rmw];
        if (r.g.n_xsrc == 0) { r.desc_off = (int)desc_host.size(); r.g.xsrc_mod_stride = mstride; }
        desc_host.push_back(SrcDesc{q.cache, q.gate_table, q.gate_mod});
        ++r.g.n_xsrc;
        r.bytes += (double)M * D * sizeof(T);
        st.cache_bytes_read += (int64_t)M * D * (int64_t)sizeof(T);
        ++xsrc_folded;
        return;
      }
      pend.push_back(q);
    };
    auto fold_take = [&](size_t bytes) { char* q = fold_cur.base + fold_used; fold_used += (bytes + 255) & ~(size_t)255; return q; };
    auto set_fold = [&](Op<T>& o, const float* c1) {
      o.g.ln_stat = hstat; o.g.ln_c1 = c1; o.g.ln_parts = D / G3_BN; o.g.ln_eps = cfg.norm_eps; o.g.ln_inv_d = 1.0f / (float)D;
    };
    for (int b = 0; b < L; ++b) {
      const Blk& B = blk[b];
      const uint8_t* ex = ex_rt ? ex_rt + (size_t)b * 3 : &executed[((size_t)s * L + b) * 3];
      const T* cb = comb_live && !bind_call ? comb + ((size_t)s * L + b) * 4 * D : nullptr;  // shared-timestep (run) plan only
      T* c0 = cache + ((size_t)b * 3 + 0) * slot;
      T* c1 = cache + ((size_t)b * 3 + 1) * slot;
      T* c2 = cache + ((size_t)b * 3 + 2) * slot;
      // --- attn1
      if (custom(b, 0)) {
        emit_ln(ops, pend, M, B.sst + 0 * D, B.sst + 1 * D, mod_s + 0 * D, mod_s + 1 * D, mstride, mstride, false, nullptr, nullptr);
        emit_custom(b, 0, xn);
        pend.push_back(Pending{io_out, B.sst + 2 * D, mod_s + 2 * D});  // (a plug-in's output: the host callback sits between the GEMM and this sum)
      } else if (ex[0]) {
        // attention in the tail of the Q projection (gemm3s_attn.h): the K,V columns first (wide tiles), then one launch that projects Q for
        // (sample, two heads) and runs their attention on the spot -- no attention launch, no Q round trip through HBM
        // Long sequences (N >= 1024, e.g. Sigma-1024's 4096 tokens): the attention kernel with FOUR query fragments per wave (attention_dma.h R = 4: K / V staged
        // once per 512 query rows, every fragment read feeds four MFMAs) beats the fused tail, whose 256-row tile fixes two fragments per wave: Sigma-1024 shape 750
        // against 850 us stand-alone, and the Q round trip it reintroduces is 2 % of that launch.
        const bool wide_items = !fp8 && DH == 72 && N >= 1024 && N % 512 == 0 && (2 * D) % G2_BN == 0 && !self_bias_on;
        const bool fuse1 = fuse_attn && !wide_items && !fp8 && DH == 72 && (2 * D) % G2_BN == 0 && N % G3_BM == 0 && N % G3A_KC == 0 && !self_bias_on;
        const bool fold1 = fold_plan && (fuse1 || wide_items) && cb && pend.empty() && stats_ok;
        const T* a_in = xn;
        const T* w_in = B.w_qkv;
        const float* b_in = B.b_qkv;
        const float* c1_in = nullptr;
        if (fold1) {  // no LayerNorm launch: the projections read h and carry (1 + scale) in their weights, shift in their bias
          T* wp = (T*)fold_take((size_t)3 * D * D * sizeof(T));
          float* c1 = (float*)fold_take((size_t)3 * D * sizeof(float));
          float* c2 = (float*)fold_take((size_t)3 * D * sizeof(float));
          fold_descs.push_back(FoldDesc{B.w_qkv, B.b_qkv, cb + 0 * D, cb + 1 * D, wp, c1, c2, 3 * D});
          a_in = h; w_in = wp; b_in = c2; c1_in = c1;
          ++ln_folded;
        } else {
          if (!pend.empty()) stats_ok = false;  // the launch (or the replay flush in front of it) folds pending replays into h
          emit_ln(ops, pend, M, B.sst + 0 * D, B.sst + 1 * D, mod_s + 0 * D, mod_s + 1 * D, mstride, mstride, true, cb ? cb + 0 * D : nullptr, cb ? cb + 1 * D : nullptr);
        }
        const double attn1_flops = 4.0 * Bt * (double)NH * N * (double)N * DH;
        if (fuse1) {
          {
            Op<T> kv = mk_gemm(KC_GEMM_KV1, EPI_BIAS, a_in, w_in + (size_t)D * D, b_in + D, qkv + D, M, 2 * D, D, D, 3 * D);
            if (fold1) set_fold(kv, c1_in + D);
            push_gemm(ops, kv, true);
          }
          Op<T> q = mk_gemm(KC_GEMM_QKV, EPI_BIAS_ATTN, a_in, w_in, b_in, qkv, M, D, D, D, 3 * D);
          if (fold1) set_fold(q, c1_in);
          q.g.att_k = qkv + D; q.g.att_v = qkv + 2 * D; q.g.att_out = attn_o; q.g.att_bias = nullptr; q.g.att_ldk = 3 * D; q.g.att_ldo = D; q.g.att_nk = N;
          q.g.att_scale_log2e = (float)(1.4426950408889634 / sqrt((double)DH));
          q.flops += attn1_flops; q.bytes += 3.0 * M * (double)D * sizeof(T) - (double)M * D * sizeof(T);  // + K, V, O; the Q tile never leaves the CU
          ops.push_back(q);
          a8_src = nullptr;
        } else {
        if ((2 * D) % G2_BN == 0) {
          // Q columns on 256x144 tiles (256 tiles) + K,V columns on 256x288 tiles (256 tiles): the wide tiles move 32 % fewer
          // L2->LDS bytes per flop, and both launches fill the chip exactly once
          Op<T> q = mk_gemm(KC_GEMM_QKV, EPI_BIAS, a_in, w_in, b_in, qkv, M, D, D, D, 3 * D);
          Op<T> kv = mk_gemm(KC_GEMM_KV1, EPI_BIAS, a_in, w_in + (size_t)D * D, b_in + D, qkv + D, M, 2 * D, D, D, 3 * D);
          if (fold1) { set_fold(q, c1_in); set_fold(kv, c1_in + D); }
          push_gemm(ops, q, true);
          push_gemm(ops, kv, true);
        } else {
          push_gemm(ops, mk_gemm(KC_GEMM_QKV, EPI_BIAS, xn, B.w_qkv, B.b_qkv, qkv, M, 3 * D, D, D, 3 * D), true);
        }
        Op<T> a; a.kind = Op<T>::ATTN; a.cls = KC_ATTN_SELF; a.attn_masked = (N % 128) != 0 || self_bias_on; a.i0 = wide_items ? 1 : 0;
        a.a.q = qkv; a.a.k = qkv + D; a.a.v = qkv + 2 * D; a.a.out = attn_o; a.a.bias = self_bias_on ? self_bias_buf : nullptr; a.a.ldq = a.a.ldk = a.a.ldv = 3 * D; a.a.ldo = D;
        a.a.B = Bt; a.a.H = NH; a.a.Nq = N; a.a.Nk = N; a.a.dh = DH; a.a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)DH));
        a.flops = attn1_flops; a.bytes = 4.0 * M * (double)D * sizeof(T);
        ops.push_back(a);
        }
        push_rmw(mk_gemm_resid(KC_GEMM_OUT1, attn_o, B.w_o1, B.b_o1, live(b, 0) ? c0 : nullptr, M, D, B.sst + 2 * D, mod_s + 2 * D, mstride));
        if (live(b, 0)) st.cache_bytes_written += (int64_t)slot * sizeof(T);
        st.linear_macs_reference += 4ll * M * D * D; st.linear_macs_executed += 4ll * M * D * D; st.attn_core_macs += 2ll * Bt * N * (int64_t)N * D;
      } else {
        pend_push(Pending{c0, B.sst + 2 * D, mod_s + 2 * D});
      }
      // --- attn2 (no norm for ada_norm_single, cached_transformer_block.py:264-267; no gate, :289)
      if (custom(b, 1)) {
        emit_replay(ops, pend, M, mstride);
        emit_custom(b, 1, h);
        pend.push_back(Pending{io_out, nullptr, nullptr});
      } else if (ex[1]) {
        if (!pend.empty()) stats_ok = false;
        emit_replay(ops, pend, M, mstride);
        const double attn2_flops = 4.0 * Bt * (double)NH * N * (double)Tt * DH;
        const bool fuse2 = fuse_attn && !fp8 && DH == 72 && N % G3_BM == 0;
        if (fuse2) {  // cross-attention in the tail of its query projection: K / V = the hoisted text K/V of this block, mask bias per key
          Op<T> q = mk_gemm(KC_GEMM_Q2, EPI_BIAS_ATTN_TAIL, h, B.w_q2, B.b_q2, qc, M, D, D, D, D);
          q.g.att_k = kv2 + (size_t)b * 2 * D; q.g.att_v = kv2 + (size_t)b * 2 * D + D; q.g.att_out = attn_o; q.g.att_bias = bias_buf; q.g.att_ldk = ldkv; q.g.att_ldo = D;
          q.g.att_nk = Tt; q.g.att_scale_log2e = (float)(1.4426950408889634 / sqrt((double)DH));
          q.flops += attn2_flops; q.bytes += 2.0 * Mt * (double)D * sizeof(T);
          ops.push_back(q);
          a8_src = nullptr;
        } else {
        push_gemm(ops, mk_gemm(KC_GEMM_Q2, EPI_BIAS, h, B.w_q2, B.b_q2, qc, M, D, D, D, D));
        Op<T> a; a.kind = Op<T>::ATTN; a.cls = KC_ATTN_CROSS; a.attn_masked = true;
        a.a.q = qc; a.a.k = kv2 + (size_t)b * 2 * D; a.a.v = kv2 + (size_t)b * 2 * D + D; a.a.out = attn_o; a.a.bias = bias_buf;
        a.a.ldq = D; a.a.ldk = a.a.ldv = ldkv; a.a.ldo = D; a.a.B = Bt; a.a.H = NH; a.a.Nq = N; a.a.Nk = Tt; a.a.dh = DH;
        a.a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)DH));
        a.flops = attn2_flops; a.bytes = (2.0 * M + 2.0 * Mt) * D * sizeof(T);
        ops.push_back(a);
        }
        push_rmw(mk_gemm_resid(KC_GEMM_OUT2, attn_o, B.w_o2, B.b_o2, live(b, 1) ? c1 : nullptr, M, D, nullptr, nullptr, 0));
        if (live(b, 1)) st.cache_bytes_written += (int64_t)slot * sizeof(T);
        if (native_at(s, b * 3 + 1) == ECAD_NATIVE_CFG_MEAN_CACHE && live(b, 1)) {
          // what is CACHED is the mean of the unconditional and the text half of the batch, in both halves (TGATE's caching rule, cached_transformer_block.py:443-452);
          // the hidden states took the un-averaged output above.  Its own launch on the slot the GEMM just wrote.
          Op<T> o; o.kind = Op<T>::CFGMEAN; o.cls = KC_CACHE_REPLAY; o.p2 = c1; o.n = slot / 2; o.bytes = 2.0 * (double)slot * sizeof(T);
          ops.push_back(o);
        }
        st.linear_macs_reference += 2ll * M * D * D + 2ll * Mt * D * D; st.linear_macs_executed += 2ll * M * D * D; st.attn_core_macs += 2ll * Bt * N * (int64_t)Tt * D;
      } else {
        pend_push(Pending{c1, nullptr, nullptr});
      }
      // --- ff
      if (custom(b, 2)) {
        emit_ln(ops, pend, M, B.sst + 3 * D, B.sst + 4 * D, mod_s + 3 * D, mod_s + 4 * D, mstride, mstride, false, nullptr, nullptr);
        emit_custom(b, 2, xn);
        pend.push_back(Pending{io_out, B.sst + 5 * D, mod_s + 5 * D});
      } else if (ex[2]) {
        const bool fold2 = fold_plan && cb && pend.empty() && stats_ok && !fp8 && gemm_impl_cls[KC_GEMM_FF1] == 2 && (4 * D) % G2_BN == 0;
        if (fold2) {
          T* wp = (T*)fold_take((size_t)4 * D * D * sizeof(T));
          float* fc1 = (float*)fold_take((size_t)4 * D * sizeof(float));
          float* fc2 = (float*)fold_take((size_t)4 * D * sizeof(float));
          fold_descs.push_back(FoldDesc{B.w_ff1, B.b_ff1, cb + 2 * D, cb + 3 * D, wp, fc1, fc2, 4 * D});
          Op<T> f1 = mk_gemm(KC_GEMM_FF1, EPI_BIAS_GELU, h, wp, fc2, ffmid, M, 4 * D, D, D, 4 * D);
          set_fold(f1, fc1);
          push_gemm(ops, f1, true);
          ++ln_folded;
        } else {
          if (!pend.empty()) stats_ok = false;
          emit_ln(ops, pend, M, B.sst + 3 * D, B.sst + 4 * D, mod_s + 3 * D, mod_s + 4 * D, mstride, mstride, true, cb ? cb + 2 * D : nullptr, cb ? cb + 3 * D : nullptr);
          push_gemm(ops, mk_gemm(KC_GEMM_FF1, EPI_BIAS_GELU, xn, B.w_ff1, B.b_ff1, ffmid, M, 4 * D, D, D, 4 * D), true);
        }
        push_rmw(mk_gemm_resid(KC_GEMM_FF2, ffmid, B.w_ff2, B.b_ff2, live(b, 2) ? c2 : nullptr, M, 4 * D, B.sst + 5 * D, mod_s + 5 * D, mstride));
        if (live(b, 2)) st.cache_bytes_written += (int64_t)slot * sizeof(T);
        st.linear_macs_reference += 8ll * M * D * D; st.linear_macs_executed += 8ll * M * D * D;
      } else {
        pend_push(Pending{c2, B.sst + 5 * D, mod_s + 5 * D});
      }
    }
    // --- _create_output: norm_out, modulate with scale_shift_table[2, D] + embedded_timestep, proj_out, unpatchify
    const size_t n_before = ops.size();
    const size_t n_pend = pend.size();
    emit_ln(ops, pend, M, sst_out, sst_out + D, emb_s, emb_s, estride, mstride);
    if (D == 1152 && P * P * COUT == 32 && ops.size() > n_before && ops.back().kind == Op<T>::LN && !ops.back().f.y8) {
      // one launch for replays + norm_out + modulate + proj_out + unpatchify (final_layer.h); the hidden states are not written back
      Op<T>& o = ops.back();
      o.kind = Op<T>::FINAL; o.cls = KC_GEMM_SMALL; o.p2 = noise; o.call_out = bind_call;
      o.flops = 2.0 * M * 32.0 * D;
      o.bytes = ((1.0 + (n_pend > 6 ? 0 : n_pend)) * M * (double)D + 32.0 * D + (double)M * 32) * sizeof(T);
    } else {
      ops.push_back(mk_gemm(KC_GEMM_SMALL, EPI_BIAS, xn, w_proj, b_proj, proj, M, PROJ_PAD, D, D, PROJ_PAD));
      {
        Op<T> o; o.kind = Op<T>::UNPATCH; o.cls = KC_MISC; o.p0 = proj; o.p2 = noise; o.call_out = bind_call;
        ops.push_back(o);
      }
    }
    // per-step reference overhead the calflops model counts (patch conv, t-embed, caption, proj_out)
    const int64_t per_sample_over = (int64_t)Tt * ((int64_t)CAP * D + (int64_t)D * D) + (int64_t)N * 16 * D + (256ll * D + 7ll * D * D) + (int64_t)N * D * 32;
    st.linear_macs_reference += (per_sample_over + (cfg.use_additional_conditions ? size_embedder_macs() : 0)) * Bt;
    st.linear_macs_executed += ((int64_t)N * 16 * D + (int64_t)N * D * 32) * Bt;
  }

  // timestep MLP + adaLN-single linear for `rows` timesteps at ts (device), results at row offset r0
  void emit_temb(std::vector<Op<T>>& ops, const float* ts, int rows, size_t r0, bool bind_call) {
    Op<T> s; s.kind = Op<T>::SINUS; s.cls = KC_MISC; s.p0 = ts; s.p2 = tsin + r0 * 256; s.i0 = rows; s.call_timestep = bind_call;
    ops.push_back(s);
    ops.push_back(mk_gemm(KC_GEMM_SMALL, EPI_BIAS_SILU, tsin + r0 * 256, w_t1, b_t1, temb1 + r0 * D, rows, D, 256, 256, D));
    ops.push_back(mk_gemm(KC_GEMM_SMALL, EPI_BIAS, temb1 + r0 * D, w_t2, b_t2, temb + r0 * D, rows, D, D, D, D));
    if (cfg.use_additional_conditions) {  // conditioning = timestep_emb + cat(resolution_emb, aspect_ratio_emb): per sample in a step plan, sample 0's in the run plan
      Op<T> z; z.kind = Op<T>::ADDSIZE; z.cls = KC_MISC; z.p2 = temb + r0 * D; z.i0 = rows; z.i1 = bind_call ? Bt : 0;
      ops.push_back(z);
    }
    Op<T> a; a.kind = Op<T>::SILU; a.cls = KC_MISC; a.p0 = temb + r0 * D; a.p2 = temb_silu + r0 * D; a.n = (size_t)rows * D;
    ops.push_back(a);
    ops.push_back(mk_gemm(KC_GEMM_SMALL, EPI_BIAS, temb_silu + r0 * D, w_ada, b_ada, mod_all + r0 * 6 * D, rows, 6 * D, D, D, 6 * D));
  }

  void emit_begin(std::vector<Op<T>>& ops) {
    const int Mt = Bt * Tt;
    Op<T> m; m.kind = Op<T>::MASKBIAS; m.cls = KC_MISC; m.p0 = mask_buf; m.p1 = enc_bias_on ? enc_bias_user : nullptr; m.p2 = bias_buf; m.i0 = Mt;
    ops.push_back(m);
    // caption projection, hoisted: step-invariant (pixart_transformer_2d_edited.py:315-321)
    ops.push_back(mk_gemm(KC_GEMM_CAPTION, EPI_BIAS_GELU, enc_buf, w_c1, b_c1, ctx_mid, Mt, D, CAP, CAP, D));
    ops.push_back(mk_gemm(KC_GEMM_CAPTION, EPI_BIAS, ctx_mid, w_c2, b_c2, ctx, Mt, D, D, D, D));
    // cross-attention K/V of the text for all blocks in ONE launch (weights stacked [L*2D, D])
    ops.push_back(mk_gemm(KC_GEMM_TEXTKV, EPI_BIAS, ctx, w_kv2_all, b_kv2_all, kv2, Mt, L * 2 * D, D, D, L * 2 * D));
    st.linear_macs_executed += (int64_t)Mt * ((int64_t)CAP * D + (int64_t)D * D) + (int64_t)Mt * L * 2 * D * D;
    if (cfg.use_additional_conditions) {  // step-invariant as well: the reference re-evaluates both embedders at every step
      Op<T> z; z.kind = Op<T>::SIZEEMB; z.cls = KC_MISC; z.i0 = Bt;
      ops.push_back(z);
      st.linear_macs_executed += (int64_t)Bt * size_embedder_macs();
    }
  }
  int64_t size_embedder_macs() const { const int64_t E = D / 3; return 3 * (256 * E + E * E); }

  int build_plans() {
    if (!finalized) return fail(ECAD_E_NOT_READY, "weights not finalized");
    if (steps == 0) return fail(ECAD_E_NOT_READY, "no schedule set");
    const int gh = Hl / P, gw = Wl / P;
    if (pos_grid != gh * 100000 + gw) { int rc = build_pos_table(gh, gw); if (rc) return rc; }
    desc_host.clear();
    st.linear_macs_reference = st.linear_macs_executed = st.attn_core_macs = 0;
    st.cache_bytes_read = st.cache_bytes_written = 0;
    plan_begin.clear(); plan_run.clear(); plan_step.assign(steps, {});
    emit_begin(plan_begin);
    // --- run plan: shared timestep per step, tables for all steps computed up front
    EcadStats keep = st;
    plan_run = plan_begin;
    fold_release_current();
    fold_prep_begin = plan_run.size();
    emit_temb(plan_run, ts_dev, steps, 0, false);
    comb_live = D == 1152 && (L < 2 || blk[1].sst - blk[0].sst == (blk[L - 1].sst - blk[0].sst) / (L - 1));
    if (comb_live) { Op<T> c; c.kind = Op<T>::COMB; c.cls = KC_MISC; plan_run.push_back(c); }
    fold_prep_ops = plan_run.size() - fold_prep_begin;
    // LayerNorm folded into the consumer GEMMs: needs the per-(step, block) comb vectors, fp16 / bf16 GEMMs, 144-column tiles for the statistics,
    // and a timestep embedding that depends on the timestep only (the 1024-MS micro-conditioning adds per-batch terms)
    fold_plan_on = ln_fold && comb_live && !fp8 && !cfg.use_additional_conditions && D == LN_PARTS * G3_BN && gemm_impl_cls[KC_GEMM_OUT1] == 0 &&
                   gemm_impl_cls[KC_GEMM_OUT2] == 0 && gemm_impl_cls[KC_GEMM_FF2] == 0 && gemm_impl_cls[KC_GEMM_KV1] == 2 && gemm_impl_cls[KC_GEMM_QKV] == 0;
    if (fold_plan_on) {
      size_t n1 = 0, n2 = 0;
      for (int s_ = 0; s_ < steps; ++s_)
        for (int b_ = 0; b_ < L; ++b_) { n1 += executed[((size_t)s_ * L + b_) * 3 + 0]; n2 += executed[((size_t)s_ * L + b_) * 3 + 2]; }
      const size_t need = n1 * ((size_t)3 * D * D * sizeof(T) + 2 * (((size_t)3 * D * 4 + 255) & ~(size_t)255) + 256) +
                          n2 * ((size_t)4 * D * D * sizeof(T) + 2 * (((size_t)4 * D * 4 + 255) & ~(size_t)255) + 256) + (n1 + n2 + 1) * sizeof(FoldDesc) + 4096;
      if (!fold_acquire(need)) { fold_plan_on = false; ++fold_refused; }
    }
    { Op<T> d; d.kind = Op<T>::DUP; d.cls = KC_MISC; plan_run.push_back(d); }
    xsrc_folded = 0;
    for (int s = 0; s < steps; ++s) {
      emit_transformer(plan_run, s, mod_all + (size_t)s * 6 * D, 0, temb + (size_t)s * D, 0, false);
      if (s + 1 < steps && solver_in_patch()) continue;  // this step's CFG + solver update rides in the next step's patch embed
      Op<T> o; o.kind = Op<T>::SOLVER; o.cls = KC_MISC; o.i0 = s;
      plan_run.push_back(o);
    }
    st.launches_per_batch = (int)plan_run.size();
    st.replays_folded = xsrc_folded;
    if (fold_plan_on) {
      if (fold_descs.empty()) { fold_pool.push_back(fold_cur); fold_cur = FoldArena{}; }
      else {  // the descriptors of the preparation kernel live behind the weights they describe
        char* dd = fold_cur.base + fold_used;
        if (fold_used + fold_descs.size() * sizeof(FoldDesc) > fold_cur.cap) return fail(ECAD_E_HIP, "fold arena too small");
        HIPCHK(hipMemcpy(dd, fold_descs.data(), fold_descs.size() * sizeof(FoldDesc), hipMemcpyHostToDevice));
      }
    }
    // temb macs: steps rows
    st.linear_macs_executed += (int64_t)steps * (256ll * D + 7ll * D * D);
    EcadStats run_stats = st;
    // --- step plans: per-sample timesteps, tables at rows [s*Bt, (s+1)*Bt)
    st = keep;
    for (int s = 0; s < steps; ++s) {
      emit_temb(plan_step[s], nullptr, Bt, (size_t)s * Bt, true);
      emit_transformer(plan_step[s], s, mod_all + (size_t)s * Bt * 6 * D, 6 * D, temb + (size_t)s * Bt * D, D, true);
    }
    st = run_stats;
    // --- upload descriptors and patch pointers
    if (desc_host.size() > desc_cap) {
      if (desc_dev) (void)hipFree(desc_dev);
      desc_cap = desc_host.size() + 1024;
      HIPCHK(hipMalloc((void**)&desc_dev, desc_cap * sizeof(SrcDesc)));
    }
    if (!desc_host.empty()) HIPCHK(hipMemcpy(desc_dev, desc_host.data(), desc_host.size() * sizeof(SrcDesc), hipMemcpyHostToDevice));
    auto patch = [&](std::vector<Op<T>>& ops) {
      for (auto& o : ops)
        if (o.desc_off >= 0) { if (o.kind == Op<T>::GEMM) o.g.xsrc = desc_dev + o.desc_off; else o.f.srcs = desc_dev + o.desc_off; }
    };
    patch(plan_run);
    for (auto& v : plan_step) patch(v);
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
    st.graph_nodes = 0;
    plan_valid = true;
    ++plan_builds;
    return ECAD_OK;
  }

  int ensure_shapes(int Bt_, int H_, int W_, int T_) {
    if (Bt_ < 1 || Bt_ > cfg.max_batch) return fail(ECAD_E_INVALID, "batch %d outside [1, max_batch=%d]", Bt_, cfg.max_batch);
    if (H_ % P || W_ % P || H_ < P || W_ < P) return fail(ECAD_E_INVALID, "latent size %dx%d not a multiple of patch %d", H_, W_, P);
    const int n = (H_ / P) * (W_ / P);
    if (n > cfg.max_tokens) return fail(ECAD_E_INVALID, "%d image tokens exceed max_tokens %d", n, cfg.max_tokens);
    if (T_ < 1 || T_ > cfg.max_text_tokens) return fail(ECAD_E_INVALID, "text tokens %d outside [1, %d]", T_, cfg.max_text_tokens);
    if (int rcw = weights_current()) return rcw;
    if (!plan_valid || Bt_ != Bt || H_ != Hl || W_ != Wl || T_ != Tt) {
      stash_current();
      Bt = Bt_; Hl = H_; Wl = W_; Tt = T_; N = n;
      if (!finalized) return fail(ECAD_E_NOT_READY, "weights not finalized");
      if (steps == 0) return fail(ECAD_E_NOT_READY, "no schedule set");
      const std::string key = make_key();
      if (restore_plans(key)) {
        const int gh = Hl / P, gw = Wl / P;
        if (pos_grid != gh * 100000 + gw) { int rc = build_pos_table(gh, gw); if (rc) return rc; }
        return ECAD_OK;
      }
      int rc = build_plans();
      if (rc == ECAD_OK) cur_key = key;
      return rc;
    }
    return ECAD_OK;
  }

  // -------------------------------------------------------------------------------------------
  // launch
  template <int EPI> void launch_gemm_epi(const GemmArgs<T>& g, hipStream_t s) {
    const int tiles = ((g.M + 127) / 128) * (g.N / 128);
    constexpr int lds = gemm_lds_bytes<128, 128>();
    hipLaunchKernelGGL((gemm_kernel<T, 128, 128, 2, 2, EPI>), dim3(tiles), dim3(256), lds, s, g);
  }
  int num_cus = 256;
  bool g3_attr[4][KC_COUNT] = {};
  // per call-site class: 0 = one 256 x 144 tile per workgroup (gemm3s.h), 5 = its 64-row latency tile; 2 = wide 256x288 tiles (gemm2w.h, N % 288 == 0).
  // Defaults from interleaved in-process A/B runs with cold caches (scripts/bench_gemm.py --cold): wide tiles wherever the tile count still fills the chip
  // (FF1, text K/V, attn1 K,V); 256 x 144 tiles for the N = 1152 shapes and FF2.  ecad_op_gemm_ex overrides the choice for one launch (parity tests).
  int gemm_impl_cls[KC_COUNT];
  bool fuse_attn = true;  // attention in the tail of the query projection when the geometry allows (env ECAD_FUSE_ATTN=0: separate launches; bit-identical)
  bool g2_attr[4][KC_COUNT] = {};
  bool gemm4_wide = true;  // the wide tile on four waves with the hand-placed K loop (gemm2w4.h; env ECAD_GEMM4=0: the 8-wave gemm2w.h; bit-identical)
  bool g2w4_attr[4][KC_COUNT] = {};
  bool force_wide_rmw = false;  // ecad_op_gemm_ex(kernel 2) at a read-modify-write site
  bool wide_rmw = true;  // the read-modify-write launches on that tile where it pays (launch_pg; env ECAD_WIDE_RMW=0: always gemm3s.h; bit-identical)
  // The same launches on gemm3s.h's own 256 x 144 tile, but on four waves with the hand-placed K loop (gemm3s4.h; round 6; bit-identical to gemm3s): where the wide
  // tile does not apply.  Measured (scripts/probe_gemm3s4.hip, one stream / two streams, us per launch): FF2 77.4 / 72.6 against gemm3s 81.1 / 80.2; the N = K = 1152
  // out-projections 30.2 / 26.2 against 29.3 / 27.2 (the K loop is 9 % shorter per step, the one-wave-per-SIMD row pass longer); with replayed components the
  // replay loads have no K loop to hide under: gemm3s.  env ECAD_GEMM3S4: 0 off, 1 (default) FF2, 2 also the out-projections, 3 also launches with replays.
  int narrow4 = 1;
  bool force_narrow4 = false;  // ecad_op_gemm_ex(kernel 7)
  bool g3s4_attr[KC_COUNT] = {};
  int64_t rmw_launches_wide = 0, rmw_launches_narrow4 = 0, rmw_launches_g3s = 0;  // read-modify-write launches by the kernel that took them (EcadStats)
  // Few rows (batch 1-4): a launch whose 256-row tiles cover less than half the chip is a latency chain (18-72 K-steps of ~0.8 us whatever the tile count);
  // the 64 x 144 latency variant of gemm3s.h (four waves of 16 x 144, two workgroups per CU) has a quarter of the MFMA work per K-step and workgroup.
  // env ECAD_LATENCY_TILES=0: off.  Attention-fused query projections take it too (64 query rows x two heads per workgroup, gemm3s_attn.h).
  bool latency_tiles = true;
  int in_flight = 1;  // image batches the caller keeps in flight on this GPU (ecad_set_batches_in_flight)
  static constexpr int latency_fill = 50;  // "few" = the 256-row tiles of the launch (times the batches in flight) cover at most this percentage of the CUs (100 / 200 / 400 measured slower)
  bool g3q_attr[4][KC_COUNT] = {};
  static int rows_of(const GemmArgs<T>& g) { return (g.m_end ? g.m_end : g.M) - g.m_begin; }  // rows this launch covers (tail split below)
  bool few_rows(const GemmArgs<T>& g) const { return latency_tiles && g.N % G3_BN == 0 && ((rows_of(g) + G3_BM - 1) / G3_BM) * (g.N / G3_BN) * in_flight * 100 <= num_cus * latency_fill; }
  bool few_rows_alone(const GemmArgs<T>& g) const { return latency_tiles && g.N % G3_BN == 0 && ((rows_of(g) + G3_BM - 1) / G3_BM) * (g.N / G3_BN) * 100 <= num_cus * latency_fill; }
  template <int EPI, int TAG> int launch_quarter(const GemmArgs<T>& g, hipStream_t s) {
    if (!g3q_attr[EPI][TAG]) {
      HIPCHK(hipFuncSetAttribute((const void*)gemm3s_kernel<T, EPI, TAG, G3Q_BM, G3Q_NW, G3Q_STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, G3Q_LDS_BYTES));
      g3q_attr[EPI][TAG] = true;
    }
    const int tq = ((rows_of(g) + G3Q_BM - 1) / G3Q_BM) * (g.N / G3_BN);
    hipLaunchKernelGGL((gemm3s_kernel<T, EPI, TAG, G3Q_BM, G3Q_NW, G3Q_STAGES>), dim3(tq), dim3(G3Q_NW * 64), G3Q_LDS_BYTES, s, g);
    return ECAD_OK;
  }
  // Few rows AND a long K (FF2: K = 4608 = 72 K-steps on 8-64 latency tiles, 44 us per launch at batch 1 -- 28 % of that batch): SPLITK_SLICES workgroups per tile, each
  // multiplying a quarter of K into fp32 partial sums, then splitk_rmw_kernel (gemm3s.h) adds them in slice order and runs the read-modify-write epilogue.  Two launches,
  // four times the workgroups, an 18-step chain instead of a 72-step one.  Deterministic; not bit-identical to the sequential-K kernels (one rounding of an fp32 sum
  // taken in another order).  Kernel id 6 of ecad_op_gemm_ex forces it.
  bool g3sk_attr[KC_COUNT] = {};
  // K-slices per tile: 4, or 2 when four would not fit the chip in one round of two co-resident workgroups per CU (batch 4: 256 latency tiles); 0 = not applicable
  // (measured: batch 1 31.4 -> 27.0 ms, batch 2 33.3 -> 30.7 ms per batch; batch 4 with four slices = two rounds: 38.7 -> 40.6, hence the one-round rule)
  int splitk_slices(const GemmArgs<T>& g) const {
    if (g.K < 2304 || g.m_begin || g.m_end || g.N != D || !g.resid || g.ldr != g.N) return 0;
    const int tq = ((g.M + G3Q_BM - 1) / G3Q_BM) * (g.N / G3_BN);
    for (int sl = SPLITK_SLICES; sl >= 2; sl >>= 1)
      if (g.K % (sl * GEMM_BK) == 0 && tq * sl * (sl == 2 ? 5 : 4) <= 8 * num_cus && (size_t)g.M * sl <= splitk_rows) return sl;  // two slices only up to 80 % of a round (batch 3: -3.6 %; batch 4, a full round: +1.8 %)
    return 0;
  }
  template <int TAG> int launch_splitk(const GemmArgs<T>& g, int slices, hipStream_t s) {
    if (!g3sk_attr[TAG]) {
      HIPCHK(hipFuncSetAttribute((const void*)gemm3s_kernel<T, EPI_SPLITK, TAG, G3Q_BM, G3Q_NW, G3Q_STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, G3Q_LDS_BYTES));
      g3sk_attr[TAG] = true;
    }
    GemmArgs<T> a = g;
    a.part = splitk_ws;
    const int tq = ((g.M + G3Q_BM - 1) / G3Q_BM) * (g.N / G3_BN);
    hipLaunchKernelGGL((gemm3s_kernel<T, EPI_SPLITK, TAG, G3Q_BM, G3Q_NW, G3Q_STAGES>), dim3(tq, slices), dim3(G3Q_NW * 64), G3Q_LDS_BYTES, s, a);
    hipLaunchKernelGGL((splitk_rmw_kernel<T>), dim3((g.M * (g.N / G3_BN) + 3) / 4), dim3(256), 0, s, a, slices);
    return ECAD_OK;
  }
  template <int EPI, int TAG> int launch_pg(const GemmArgs<T>& g, hipStream_t s) {
    if constexpr (EPI == EPI_BIAS_CACHE_RESID && (TAG == KC_GEMM_FF2 || TAG == KC_GEMM_SMALL)) {
      const int slices = (gemm_impl_cls[TAG] == 6 || (split_k && gemm_impl_cls[TAG] != 5 && few_rows_alone(g))) ? splitk_slices(g) : 0;
      if (slices) return launch_splitk<TAG>(g, slices, s);
      if (gemm_impl_cls[TAG] == 6) return fail(ECAD_E_UNSUPPORTED, "split-K needs K >= 2304 in whole K-steps per slice, N = D and one round of at most %d slice-tiles", 2 * num_cus);
    }
    if (gemm_impl_cls[TAG] == 6) return fail(ECAD_E_UNSUPPORTED, "split-K is instantiated for the read-modify-write epilogue of the FF2 / small call sites");
    if (gemm_impl_cls[TAG] == 5 || few_rows(g)) return launch_quarter<EPI, TAG>(g, s);
    if constexpr (TAG == KC_GEMM_FF1 || TAG == KC_GEMM_TEXTKV || TAG == KC_GEMM_SMALL || TAG == KC_GEMM_KV1 || (TAG == KC_GEMM_QKV && EPI == EPI_BIAS)) {
      // wide 256x288 tiles -- unless they would leave half the chip idle (FF1 at batch 4-7 alone on the GPU: 128-224 tiles): then the 256x144 tiles below, twice as many
      const bool wide_underfills = g.N % G3_BN == 0 && ((rows_of(g) + G2_BM - 1) / G2_BM) * (g.N / G2_BN) * in_flight * 2 <= num_cus;
      // the stand-alone Q projection (N = 1152; Sigma-1024 and every geometry whose self-attention is its own launch): wide tiles as soon as they come in whole rounds of the
      // chip -- M = 32768: 512 wide tiles, 80.2 us against 99.8 us on 1024 narrow ones (same box, hot operands); at M = 8192 (128 wide tiles) the two are level and the narrow tile stays
      const int wide_tiles = ((rows_of(g) + G2_BM - 1) / G2_BM) * (g.N / G2_BN);
      const bool wide_q = TAG == KC_GEMM_QKV && gemm_impl_cls[TAG] == 0 && g.N % G2_BN == 0 && wide_tiles >= num_cus && wide_tiles % num_cus == 0;
      if ((gemm_impl_cls[TAG] == 2 || wide_q) && g.N % G2_BN == 0 && !wide_underfills) {
        if (!g2_attr[EPI][TAG]) {
          HIPCHK(hipFuncSetAttribute((const void*)gemm2w_kernel<T, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
          g2_attr[EPI][TAG] = true;
        }
        const int tiles2 = ((rows_of(g) + G2_BM - 1) / G2_BM) * (g.N / G2_BN);
        if constexpr (EPI != EPI_BIAS_CACHE_RESID) {
          // whole 256-row tiles: the same tile on four waves, one per SIMD, K loop hand-placed (gemm2w4.h): bit-identical, 9-14 % less time per launch stand-alone
          if (gemm4_wide && g2w4_shape_ok(g)) {
            if (!g2w4_attr[EPI][TAG]) {
              HIPCHK(hipFuncSetAttribute((const void*)gemm2w4_kernel<T, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G2W4_LDS_BYTES));
              g2w4_attr[EPI][TAG] = true;
            }
            hipLaunchKernelGGL((gemm2w4_kernel<T, EPI, TAG>), dim3(tiles2 > num_cus && tiles2 % num_cus == 0 ? num_cus : tiles2), dim3(256), G2W4_LDS_BYTES, s, g);
            return ECAD_OK;
          }
        }
        // direct epilogue (LDS-free): one workgroup per CU walks its tiles and fetches the next tile's first K-step under the epilogue (gemm2w.h)
        const bool walk = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && g.epi_direct > (EPI == EPI_BIAS_GELU ? 0 : 1) && tiles2 > num_cus && tiles2 % num_cus == 0;
        hipLaunchKernelGGL((gemm2w_kernel<T, EPI, TAG>), dim3(walk ? num_cus : tiles2), dim3(G2_NW * 64), G2_LDS_BYTES, s, g);
        return ECAD_OK;
      }
    }
    if constexpr (EPI == EPI_BIAS_CACHE_RESID && (TAG == KC_GEMM_OUT1 || TAG == KC_GEMM_OUT2 || TAG == KC_GEMM_FF2)) {
      // Read-modify-write launches on the wide 256 x 288 tile (gemm2w4.h; bit-identical to gemm3s, so the choice may depend on the batches in flight): 1.5x the
      // MFMA work per operand byte of the 256 x 144 tile.  Taken when the wide tiles come in whole rounds of the chip (Sigma-1024: 512; -13 % per launch), or when
      // they cover half of it and another batch is in flight on its own stream -- two half-chip launches of different streams run side by side: 24-25 us per
      // launch against 27.3 (N = K = 1152), 65-68 against 81 (FF2) at M = 8192; alone on the GPU the same launch would take 41 / 99 us, hence the condition.
      if (gemm4_wide && wide_rmw && gemm_impl_cls[TAG] == 0 && g.N % G2_BN == 0 && g2w4_shape_ok(g) && g2w4_rmw_ok(g)) {
        const int tiles2 = (rows_of(g) / G2_BM) * (g.N / G2_BN);
        if (force_wide_rmw || tiles2 % num_cus == 0 || (in_flight >= 2 && 2 * tiles2 >= num_cus && tiles2 < num_cus)) {
          if (!g2w4_attr[EPI][TAG]) {
            HIPCHK(hipFuncSetAttribute((const void*)gemm2w4_kernel<T, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G2W4_LDS_BYTES));
            g2w4_attr[EPI][TAG] = true;
          }
          hipLaunchKernelGGL((gemm2w4_kernel<T, EPI, TAG>), dim3(tiles2 > num_cus ? num_cus : tiles2), dim3(256), G2W4_LDS_BYTES, s, g);
          ++rmw_launches_wide;
          return ECAD_OK;
        }
      }
    }
    if constexpr (EPI == EPI_BIAS_CACHE_RESID && (TAG == KC_GEMM_OUT1 || TAG == KC_GEMM_OUT2 || TAG == KC_GEMM_FF2)) {
      // ... or on the 256 x 144 tile on four waves (gemm3s4.h): full-chip, bit-identical to gemm3s below
      const bool want = force_narrow4 || (gemm4_wide && gemm_impl_cls[TAG] == 0 && (TAG == KC_GEMM_FF2 ? narrow4 >= 1 : narrow4 >= 2) && (g.n_xsrc == 0 || narrow4 >= 3));
      if (want && g3s4_ok(g)) {
        if (!g3s4_attr[TAG]) {
          HIPCHK(hipFuncSetAttribute((const void*)gemm3s4_kernel<T, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G3S4_LDS_BYTES));
          g3s4_attr[TAG] = true;
        }
        hipLaunchKernelGGL((gemm3s4_kernel<T, TAG>), dim3((rows_of(g) / G3_BM) * (g.N / G3_BN)), dim3(256), G3S4_LDS_BYTES, s, g);
        ++rmw_launches_narrow4;
        return ECAD_OK;
      }
      if (force_narrow4) return fail(ECAD_E_UNSUPPORTED, "gemm3s4 needs whole 256-row tiles inside one sample each, K %% 192 == 0 and at most %d replayed components", G3_XSRC_MAX);
      ++rmw_launches_g3s;
    }
    if (gemm_impl_cls[TAG] == 0 || (gemm_impl_cls[TAG] == 2 && g.N % G3_BN == 0)) {  // (2: a wide call site whose wide tiles would underfill the chip)
      if (!g3_attr[EPI][TAG]) {
        HIPCHK(hipFuncSetAttribute((const void*)gemm3s_kernel<T, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
        g3_attr[EPI][TAG] = true;
      }
      const int tiles3 = ((rows_of(g) + G3_BM - 1) / G3_BM) * (g.N / G3_BN);
      hipLaunchKernelGGL((gemm3s_kernel<T, EPI, TAG>), dim3(tiles3), dim3(G3_NW * 64), G3_LDS_BYTES, s, g);
      return ECAD_OK;
    }
    return fail(ECAD_E_INVALID, "no GEMM kernel selected for call site %d (implementation id %d)", TAG, gemm_impl_cls[TAG]);
  }
  // query projection with its attention in the tail (gemm3s_attn.h): gemm3s only, tiles inside one sample, two 72-wide heads per tile
  bool g3a_attr[2] = {}, g3qa_attr[2] = {};
  template <int EPI, int TAG> int launch_g3_attn(const GemmArgs<T>& g, hipStream_t s) {
    if (g.N % G3_BN || g.rows_per_batch % G3_BM || g.M % G3_BM || DH != 72 || !g.att_k || !g.att_v || !g.att_out || g.att_nk < 1 || g.att_ldo % 4 || g.att_ldk % 8)
      return fail(ECAD_E_UNSUPPORTED, "fused attention needs 256-row tiles inside one sample and head_dim 72");
    if (EPI == EPI_BIAS_ATTN && g.att_nk % G3A_KC) return fail(ECAD_E_UNSUPPORTED, "fused self-attention needs a key count that is a multiple of 128");
    if (gemm_impl_cls[TAG] == 5 || few_rows(g)) {  // few rows: 64 query rows x two heads per workgroup, four times the workgroups (the same arithmetic per row)
      if (!g3qa_attr[EPI - EPI_BIAS_ATTN]) {
        HIPCHK(hipFuncSetAttribute((const void*)gemm3s_kernel<T, EPI, TAG, G3Q_BM, G3Q_NW, G3Q_STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, G3Q_LDS_BYTES));
        g3qa_attr[EPI - EPI_BIAS_ATTN] = true;
      }
      hipLaunchKernelGGL((gemm3s_kernel<T, EPI, TAG, G3Q_BM, G3Q_NW, G3Q_STAGES>), dim3((rows_of(g) / G3Q_BM) * (g.N / G3_BN)), dim3(G3Q_NW * 64), G3Q_LDS_BYTES, s, g);
      return ECAD_OK;
    }
    if (!g3a_attr[EPI - EPI_BIAS_ATTN]) {
      HIPCHK(hipFuncSetAttribute((const void*)gemm3s_kernel<T, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
      g3a_attr[EPI - EPI_BIAS_ATTN] = true;
    }
    const int tiles3 = (rows_of(g) / G3_BM) * (g.N / G3_BN);
    hipLaunchKernelGGL((gemm3s_kernel<T, EPI, TAG>), dim3(tiles3), dim3(G3_NW * 64), G3_LDS_BYTES, s, g);
    return ECAD_OK;
  }
  // Tail split (round 3): a launch of w whole rounds of 256-row tiles plus a last round that would leave at least half the CUs idle (batch 100:
  // 1600 tiles of the N = 1152 GEMMs = 6.25 rounds) goes out as two launches over row ranges -- the w whole rounds as they are, and the remaining rows as a
  // problem of their own, which the few-rows / wide-underfill rules then put on 64-row latency tiles or 256 x 144 tiles.  Only with one batch in flight (with
  // more, the other batches' kernels fill the idle CUs).  Same arithmetic per row, so results do not depend on it (env ECAD_TAIL_SPLIT=0: off).
  bool tail_split = true;
  int launch_gemm(int epi, int cls, const GemmArgs<T>& g, hipStream_t s) {
    if (g.K % GEMM_BK || g.M < 1) return fail(ECAD_E_INVALID, "gemm shape M=%d N=%d K=%d unsupported (K%%64)", g.M, g.N, g.K);
    if (tail_split && in_flight == 1 && !g.m_begin && !g.m_end && g.N % G3_BN == 0 && g.M % G3_BM == 0) {
      const bool wide_site = cls == KC_GEMM_FF1 || cls == KC_GEMM_TEXTKV || cls == KC_GEMM_SMALL || cls == KC_GEMM_KV1;
      const int tn_sel = wide_site && gemm_impl_cls[cls] == 2 && g.N % G2_BN == 0 ? g.N / G2_BN : g.N / G3_BN;
      const int total = (g.M / G3_BM) * tn_sel, whole = total / num_cus, rem = total % num_cus;
      if (whole >= 1 && rem > 0 && rem * 2 <= num_cus && (whole * num_cus) % tn_sel == 0 && gemm_impl_cls[cls] != 5) {
        GemmArgs<T> part = g;
        part.m_end = (whole * num_cus / tn_sel) * G3_BM;
        int rc = launch_gemm(epi, cls, part, s);
        if (rc) return rc;
        part.m_begin = part.m_end;
        part.m_end = g.M;
        return launch_gemm(epi, cls, part, s);
      }
    }
    if (epi == EPI_BIAS_ATTN && cls == KC_GEMM_QKV) return launch_g3_attn<EPI_BIAS_ATTN, KC_GEMM_QKV>(g, s);
    if (epi == EPI_BIAS_ATTN_TAIL && cls == KC_GEMM_Q2) return launch_g3_attn<EPI_BIAS_ATTN_TAIL, KC_GEMM_Q2>(g, s);
    if (epi == EPI_BIAS_ATTN || epi == EPI_BIAS_ATTN_TAIL) return fail(ECAD_E_INVALID, "fused attention epilogue on call site %d", cls);
    if (g.N % G3_BN == 0) {
#define G144(E, C) if (epi == E && cls == C) return launch_pg<E, C>(g, s)
      G144(EPI_BIAS, KC_GEMM_QKV); G144(EPI_BIAS, KC_GEMM_KV1); G144(EPI_BIAS_CACHE_RESID, KC_GEMM_OUT1); G144(EPI_BIAS, KC_GEMM_Q2); G144(EPI_BIAS_CACHE_RESID, KC_GEMM_OUT2);
      G144(EPI_BIAS_GELU, KC_GEMM_FF1); G144(EPI_BIAS_CACHE_RESID, KC_GEMM_FF2); G144(EPI_BIAS, KC_GEMM_TEXTKV);
      G144(EPI_BIAS_GELU, KC_GEMM_CAPTION); G144(EPI_BIAS, KC_GEMM_CAPTION);
      G144(EPI_BIAS, KC_GEMM_SMALL); G144(EPI_BIAS_GELU, KC_GEMM_SMALL); G144(EPI_BIAS_SILU, KC_GEMM_SMALL); G144(EPI_BIAS_CACHE_RESID, KC_GEMM_SMALL);
#undef G144
      return fail(ECAD_E_INVALID, "no 144-column GEMM instantiation for epilogue %d class %d", epi, cls);
    }
    if (g.N % 128) return fail(ECAD_E_INVALID, "gemm N=%d must be a multiple of 144 or 128", g.N);
    if (g.n_xsrc > 0) return fail(ECAD_E_INVALID, "gemm N=%d: %d replayed components attached to a kernel without the replay epilogue", g.N, g.n_xsrc);
    switch (epi) {
      case EPI_BIAS: launch_gemm_epi<EPI_BIAS>(g, s); break;
      case EPI_BIAS_GELU: launch_gemm_epi<EPI_BIAS_GELU>(g, s); break;
      case EPI_BIAS_SILU: launch_gemm_epi<EPI_BIAS_SILU>(g, s); break;
      case EPI_BIAS_CACHE_RESID: launch_gemm_epi<EPI_BIAS_CACHE_RESID>(g, s); break;
      default: return fail(ECAD_E_INVALID, "bad epilogue %d", epi);
    }
    return ECAD_OK;
  }
  bool attn_attr[3] = {};
  template <int KC, bool MASKED> int launch_attn_t(const AttnArgs& a, int slot, int wg_per_cu, hipStream_t s) {
    constexpr int lds = attn_lds_bytes<KC>();
    if (!attn_attr[slot]) {
      HIPCHK(hipFuncSetAttribute((const void*)attn_kernel<T, KC, MASKED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attn_attr[slot] = true;
    }
    const int items = a.B * a.H * ((a.Nq + ATT_QB - 1) / ATT_QB);
    const int cap = num_cus * wg_per_cu;
    hipLaunchKernelGGL((attn_kernel<T, KC, MASKED>), dim3(items < cap ? items : cap), dim3(512), lds, s, a);
    return ECAD_OK;
  }
  int attn_shared = 1;  // attention128.h (d_h 72: both query fragments of a wave share each K / V fragment read); 0 (ecad_op_attention_ex only): the generic attention.h, which serves every other head size
  bool attn_shared_attr[2] = {};
  bool attn_dma_attr = false;
  bool attn_w472_on = true;   // env ECAD_ATTN_W472=0: long-sequence self-attention stays on attn_dma_kernel<T, 72, 4> (eight waves, compiler-scheduled)
  bool attn_w472_attr = false;
  int launch_attn(const AttnArgs& a, bool masked, hipStream_t s, bool wide = false, bool four_fragment = false) {
    if (wide && !four_fragment && attn_w472_on && attn_w472_ok(a, masked)) {  // one wave per SIMD, generated instruction stream (attention_w472.h): 256 query rows per work item
      if (!attn_w472_attr) {
        HIPCHK(hipFuncSetAttribute((const void*)attn_w472_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES));
        attn_w472_attr = true;
      }
      const int items = a.B * a.H * (a.Nq / 256);
      hipLaunchKernelGGL((attn_w472_kernel<T>), dim3(items < num_cus ? items : num_cus), dim3(256), W4_LDS_BYTES, s, a);
      return ECAD_OK;
    }
    if (wide && a.dh == 72 && !masked && a.Nk % 128 == 0 && a.seg_t == 0 && a.Nq % 512 == 0) {  // four query fragments per wave: 512 query rows per work item (attention_dma.h)
      if (!attn_dma_attr) {
        HIPCHK(hipFuncSetAttribute((const void*)attn_dma_kernel<T, 72, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_dma_lds_bytes<72>()));
        attn_dma_attr = true;
      }
      const int items4 = a.B * a.H * (a.Nq / 512);
      hipLaunchKernelGGL((attn_dma_kernel<T, 72, 4>), dim3(items4 < num_cus ? items4 : num_cus), dim3(512), attn_dma_lds_bytes<72>(), s, a);
      return ECAD_OK;
    }
    if (attn_shared && a.dh == 72) {
      const bool tail = masked || a.Nk % 128;
      if (!attn_shared_attr[tail]) {
        if (tail) HIPCHK(hipFuncSetAttribute((const void*)attn128_kernel<T, true, 72>, hipFuncAttributeMaxDynamicSharedMemorySize, A72_LDS_BYTES));
        else HIPCHK(hipFuncSetAttribute((const void*)attn128_kernel<T, false, 72>, hipFuncAttributeMaxDynamicSharedMemorySize, A72_LDS_BYTES));
        attn_shared_attr[tail] = true;
      }
      const int items = a.B * a.H * ((a.Nq + ATT_QB - 1) / ATT_QB);
      const int grid = items < num_cus ? items : num_cus;
      if (tail) hipLaunchKernelGGL((attn128_kernel<T, true, 72>), dim3(grid), dim3(512), A72_LDS_BYTES, s, a);
      else hipLaunchKernelGGL((attn128_kernel<T, false, 72>), dim3(grid), dim3(512), A72_LDS_BYTES, s, a);
      return ECAD_OK;
    }
    // KC = 128 keys per LDS chunk for both kinds (a 256-key strip in registers spills: 64 score + 36 staging VGPRs)
    if (masked || a.Nk % 128) return launch_attn_t<128, true>(a, 0, 1, s);
    return launch_attn_t<128, false>(a, 1, 1, s);
  }
  static constexpr int ln16 = 1;  // 16 lanes per row (D = 1152: 9 chunks per lane), ln16 rows per lane group: 16.2 -> 15.3 us per launch at 1 row, 18.3 at 2, 25.7 at 4
                 // (the pass is bound by the number of loads in flight, not by bandwidth); env ECAD_LN16=0: one wave per row
  int launch_ln(const FlushArgs& f, hipStream_t s) {
    const int cpr = f.D / 8;
    const int grid = (f.M + 3) / 4;
    if (f.D % 8) return fail(ECAD_E_INVALID, "D %% 8 != 0");
    if (f.D == 1152) {  // 16 lanes per row, 9 chunks per lane
      const dim3 g16(((f.M + 15) / 16 + ln16 - 1) / ln16);  // ln16 rows per 16-lane group
      if (f.comb_shift) hipLaunchKernelGGL((ln_modulate16_kernel<T, 9, true>), g16, dim3(256), 0, s, f);
      else hipLaunchKernelGGL((ln_modulate16_kernel<T, 9, false>), g16, dim3(256), 0, s, f);
      return ECAD_OK;
    }
    if (cpr <= 64 * 3) hipLaunchKernelGGL((ln_modulate_kernel<T, 3>), dim3(grid), dim3(256), 0, s, f);
    else if (cpr <= 64 * 6) hipLaunchKernelGGL((ln_modulate_kernel<T, 6>), dim3(grid), dim3(256), 0, s, f);
    else return fail(ECAD_E_UNSUPPORTED, "D=%d too wide for ln kernel", f.D);
    return ECAD_OK;
  }
  // Rows kernel (4 rows per thread, gate fetched once per 4 chunks, non-temporal cache reads) whenever the shapes allow: in-run A/B on the
  // 84-source cache-hit step at Bt 32: 389 -> 258 us (4.17 -> 6.29 TB/s), mid-step replays 162 -> 105 us; 2 rows 340 us, 8 rows 276 us;
  // capping the grid at 6 or 8 resident blocks per CU, or 8 sources in flight in the flat kernel: slower or equal.  The
  // one-chunk-per-thread kernel takes the ragged shapes.
  void launch_replay(const FlushArgs& f, hipStream_t s, bool hit_step = false) {
    if (f.M % 4 == 0 && f.rows_per_batch % 4 == 0) {
      const size_t nb = ((size_t)(f.M / 4) * (f.D / 8) + 255) / 256;
      if (hit_step) hipLaunchKernelGGL((cache_accumulate_rows_kernel<T, 1, 4>), dim3((unsigned)nb), dim3(256), 0, s, f);
      else hipLaunchKernelGGL((cache_accumulate_rows_kernel<T, 0, 4>), dim3((unsigned)nb), dim3(256), 0, s, f);
      return;
    }
    const size_t total = (size_t)f.M * (f.D / 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (hit_step) hipLaunchKernelGGL((cache_accumulate_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, s, f);
    else hipLaunchKernelGGL((cache_accumulate_kernel<T, 0>), dim3((unsigned)blocks), dim3(256), 0, s, f);
  }

  int launch_op(const Op<T>& o, hipStream_t s, float guidance) {
    const int M = Bt * N;
    switch (o.kind) {
      case Op<T>::GEMM: return launch_gemm(o.epi, o.cls, o.g, s);
      case Op<T>::QUANT:
        launch_quant_rows_fp8<T>((const T*)o.p0, (int64_t)o.i0, o.g.K, (const T*)nullptr, (int64_t)0, 0, a8, (int64_t)o.g.K,
                           sa8, o.g.M, s);
        break;
      case Op<T>::GEMM8: {
        Gemm8Args<T> q{};
        q.A8 = a8; q.W8 = o.w8; q.sa = sa8; q.sw = o.sw8; q.lda8 = o.g.K; q.e = o.g;
        if (o.i0) {  // FF1 emitting e4m3 + block scales for FF2 (push_gemm checked the shape: whole rounds of wide tiles)
          q.q8_out = a8b; q.e8_out = e8buf; q.ldq8 = o.g.N;
          hipLaunchKernelGGL((gemm2w8_kernel<T, EPI_BIAS_GELU, true>), dim3((o.g.M / G2_BM) * (o.g.N / G2_BN)), dim3(G2_NW * 64), G2_LDS_BYTES, s, q);
          break;
        }
        if (o.i1) {  // FF2 reading them
          q.A8 = a8b; q.e8_in = e8buf; q.e8_ld = o.g.K / G2_BN;
          hipLaunchKernelGGL((gemm3s8_kernel<T, EPI_BIAS_CACHE_RESID, true>), dim3(((o.g.M + G3_BM - 1) / G3_BM) * (o.g.N / G3_BN)), dim3(G3_NW * 64), G3_LDS_BYTES + G3S8_TAB_BYTES, s, q);
          break;
        }
        if (o.epi != EPI_BIAS_CACHE_RESID && o.g.N % G2_BN == 0 && ((o.g.M + G2_BM - 1) / G2_BM) * (o.g.N / G2_BN) >= num_cus) {  // 256 x 288 tiles, 4 x 2 waves
          const int tiles2 = ((o.g.M + G2_BM - 1) / G2_BM) * (o.g.N / G2_BN);
          if (o.epi == EPI_BIAS) hipLaunchKernelGGL((gemm2w8_kernel<T, EPI_BIAS>), dim3(tiles2), dim3(G2_NW * 64), G2_LDS_BYTES, s, q);
          else hipLaunchKernelGGL((gemm2w8_kernel<T, EPI_BIAS_GELU>), dim3(tiles2), dim3(G2_NW * 64), G2_LDS_BYTES, s, q);
          break;
        }
        const int tiles = ((o.g.M + G3_BM - 1) / G3_BM) * (o.g.N / G3_BN);
        if (o.epi == EPI_BIAS) hipLaunchKernelGGL((gemm3s8_kernel<T, EPI_BIAS>), dim3(tiles), dim3(G3_NW * 64), G3_LDS_BYTES, s, q);
        else if (o.epi == EPI_BIAS_GELU) hipLaunchKernelGGL((gemm3s8_kernel<T, EPI_BIAS_GELU>), dim3(tiles), dim3(G3_NW * 64), G3_LDS_BYTES, s, q);
        else hipLaunchKernelGGL((gemm3s8_kernel<T, EPI_BIAS_CACHE_RESID>), dim3(tiles), dim3(G3_NW * 64), G3_LDS_BYTES, s, q);
        break;
      }
      case Op<T>::ATTN: return launch_attn(o.a, o.attn_masked, s, o.i0 == 1);
      case Op<T>::LN: return launch_ln(o.f, s);
      case Op<T>::REPLAY: launch_replay(o.f, s, o.cls == KC_CACHE_HIT_STEP); break;
      case Op<T>::PATCH: {
        const T* src = o.call_latents ? (const T*)call_latents : (const T*)o.p0;
        if (o.i0 > 0) {  // solver update of step o.i0 - 1 on the fly: state pair A = (lat_buf, x0_prev) after an even number of updates, B after an odd one
          const int k = o.i0 - 1;
          PatchSolve sv{};
          sv.noise = noise; sv.lat_old = (k & 1) ? lat_buf2 : lat_buf; sv.x0_old = (k & 1) ? x0_prev2 : x0_prev; sv.lat_new = (k & 1) ? lat_buf : lat_buf2;
          sv.x0_new = (k & 1) ? x0_prev : x0_prev2; sv.coef = coef_dev + k * 8; sv.guidance = guidance; sv.B = Bt / 2; sv.Cout = COUT;
          hipLaunchKernelGGL((patch_embed_cols_kernel<T, 256, true>), dim3(D / 128, Bt * ((N + 255) / 256)), dim3(256), 0, s, src, w_patch, b_patch, pos, h, Hl, Wl, D, sv);
          break;
        }
        hipLaunchKernelGGL((patch_embed_cols_kernel<T, 256, false>), dim3(D / 128, Bt * ((N + 255) / 256)), dim3(256), 0, s, src, w_patch, b_patch, pos, h, Hl, Wl, D, PatchSolve{});
        break;
      }
      case Op<T>::SINUS: {
        const float* ts = o.call_timestep ? call_timestep : (const float*)o.p0;
        hipLaunchKernelGGL((timestep_sinusoid_kernel<T>), dim3(o.i0), dim3(128), 0, s, ts, (T*)o.p2, o.i0);
        break;
      }
      case Op<T>::SILU: hipLaunchKernelGGL((silu_kernel<T>), dim3((unsigned)((o.n + 255) / 256)), dim3(256), 0, s, (const T*)o.p0, (T*)o.p2, o.n); break;
      case Op<T>::SIZEEMB:
        hipLaunchKernelGGL((size_embed_kernel<T>), dim3(o.i0 * 3), dim3(256), (256 + D / 3) * sizeof(float), s, cond_dev, w_sz1[0], b_sz1[0], w_sz2[0], b_sz2[0], w_sz1[1], b_sz1[1],
                           w_sz2[1], b_sz2[1], size_emb, D / 3);
        break;
      case Op<T>::ADDSIZE: hipLaunchKernelGGL((add_size_emb_kernel<T>), dim3(o.i0), dim3(256), 0, s, (T*)o.p2, (const T*)size_emb, D, o.i1); break;
      case Op<T>::COMB:
        hipLaunchKernelGGL((comb_mod_kernel<T>), dim3(steps, L), dim3(256), 0, s, blk[0].sst, (int64_t)(L > 1 ? blk[1].sst - blk[0].sst : 0), (const T*)mod_all, comb, L, D);
        break;
      case Op<T>::MASKBIAS: hipLaunchKernelGGL(mask_bias_kernel, dim3((o.i0 + 255) / 256), dim3(256), 0, s, (const uint8_t*)o.p0, (const float*)o.p1, (float*)o.p2, o.i0); break;
      case Op<T>::FINAL: {
        FinalProj fq{};
        fq.w = w_proj; fq.bias = b_proj; fq.out = o.call_out ? call_out : o.p2; fq.gh = Hl / P; fq.gw = Wl / P; fq.P = P; fq.Cout = COUT;
        hipLaunchKernelGGL((final_layer_kernel<T, 36>), dim3((o.f.M + 15) / 16), dim3(256), 0, s, o.f, fq);
        break;
      }
      case Op<T>::UNPATCH: {
        T* dst = o.call_out ? (T*)call_out : (T*)o.p2;
        const int total = Bt * COUT * Hl * Wl;
        hipLaunchKernelGGL((unpatchify_kernel<T>), dim3((total + 255) / 256), dim3(256), 0, s, (const T*)o.p0, dst, Bt, Hl / P, Wl / P, P, COUT, PROJ_PAD);
        break;
      }
      case Op<T>::SOLVER: {
        const int B = Bt / 2, total = B * CIN * Hl * Wl;
        // o.i0 updates ran before this one; with the solver in the patch embed they alternated between the two state pairs, and this launch (the last step's) ends in pair A
        const bool odd = solver_in_patch() && (o.i0 & 1);
        hipLaunchKernelGGL((cfg_solver_kernel<T>), dim3((total + 255) / 256), dim3(256), 0, s, noise, odd ? lat_buf2 : lat_buf, odd ? x0_prev2 : x0_prev, lat_buf, x0_prev,
                           solver_in_patch() ? (T*)nullptr : model_in, coef_dev + o.i0 * 8, guidance, B, CIN, COUT, Hl * Wl);
        break;
      }
      case Op<T>::CUSTOM: {
        if (!custom_fn || !io_in || !io_out) return fail(ECAD_E_NOT_READY, "block %d component %d is routed to a compute plug-in but no callback is registered", o.i0, o.i1);
        hipError_t ce = hipMemcpyAsync(io_in, o.p0, (size_t)M * D * sizeof(T), hipMemcpyDeviceToDevice, s);
        if (ce != hipSuccess) return fail(ECAD_E_HIP, "staging the plug-in input failed: %s", hipGetErrorString(ce));
        const int rc = custom_fn(custom_user, cur_step_idx, o.i0, o.i1, io_in, io_out, Bt, N, D, (void*)s);
        if (rc) return fail(ECAD_E_INVALID, "compute plug-in of block %d component %d failed (callback returned %d)", o.i0, o.i1, rc);
        break;
      }
      case Op<T>::CFGMEAN:
        hipLaunchKernelGGL((cfg_mean_halves_kernel<T>), dim3((unsigned)((o.n / 8 + 255) / 256)), dim3(256), 0, s, (T*)o.p2, o.n);
        break;
      case Op<T>::DUP: {
        const int total = (Bt / 2) * CIN * Hl * Wl;
        hipLaunchKernelGGL((cfg_duplicate_kernel<T>), dim3((total + 255) / 256), dim3(256), 0, s, lat_buf, model_in, total);
        break;
      }
    }
    (void)M;
    return ECAD_OK;
  }

  int run_ops(const std::vector<Op<T>>& ops, hipStream_t s, float guidance, bool allow_profile) {
    const bool prof = profiling && allow_profile;
    if (prof) {
      while (ev_pool.size() < 2 * ops.size()) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); ev_pool.push_back(e); }
      ev_cls.clear();
    }
    size_t k = 0;
    for (const auto& o : ops) {
      if (prof) HIPCHK(hipEventRecord(ev_pool[2 * k], s));
      int rc = launch_op(o, s, guidance);
      if (rc) return rc;
      if (prof) { HIPCHK(hipEventRecord(ev_pool[2 * k + 1], s)); ev_cls.push_back(o.cls); }
      ++k;
    }
    HIPCHK(hipGetLastError());
    if (prof) {
      HIPCHK(hipStreamSynchronize(s));
      for (size_t i = 0; i < ops.size(); ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev_pool[2 * i], ev_pool[2 * i + 1]));
        const int c = ops[i].cls;
        prof_ms[c] += ms; prof_launch[c]++; prof_flops[c] += ops[i].flops; prof_bytes[c] += ops[i].bytes;
      }
    }
    return ECAD_OK;
  }

  // -------------------------------------------------------------------------------------------
  int stage_inputs(const void* enc, const uint8_t* mask, hipStream_t s) {
    const size_t Mt = (size_t)Bt * Tt;
    HIPCHK(hipMemcpyAsync(enc_buf, enc, Mt * CAP * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (mask) HIPCHK(hipMemcpyAsync(mask_buf, mask, Mt, hipMemcpyDeviceToDevice, s));
    else HIPCHK(hipMemsetAsync(mask_buf, 1, Mt, s));
    return ECAD_OK;
  }

  int begin_batch(int Bt_, int H_, int W_, const void* enc, const uint8_t* mask, int T_, hipStream_t s, bool keep_caches) override {
    HIPCHK(hipSetDevice(dev));
    if (!enc) return fail(ECAD_E_INVALID, "enc_dev is NULL");
    if (keep_caches && (!batch_begun || !plan_valid || Bt_ != Bt || H_ != Hl || W_ != Wl || T_ != Tt))
      return fail(ECAD_E_NOT_READY, "ecad_rebind_batch_inputs: no batch of that geometry has begun");
    int rc = ensure_shapes(Bt_, H_, W_, T_);
    if (rc) return rc;
    if (keep_caches && !batch_begun)  // ensure_shapes found new weights in the shared arena (weights_current): the batch this call wants to continue is gone
      return fail(ECAD_E_NOT_READY, "ecad_rebind_batch_inputs: the weights changed since the batch began; call ecad_begin_image_batch");
    if ((self_bias_on && (bias_bt != Bt_ || bias_n != N)) || (enc_bias_on && (bias_bt != Bt_ || bias_t != T_)))
      return fail(ECAD_E_INVALID, "the attention bias was set for batch %d, %d tokens, %d text tokens; this batch has %d, %d, %d: call ecad_set_attention_bias for it", bias_bt, bias_n, bias_t, Bt_, N, T_);
    if (cfg.use_additional_conditions && cond_bt != Bt_)
      return fail(ECAD_E_NOT_READY, "`added_cond_kwargs` cannot be None when using additional conditions for `adaln_single`: call ecad_set_added_cond for batch %d first", Bt_);
    rc = stage_inputs(enc, mask, s);
    if (rc) return rc;
    rc = run_ops(plan_begin, s, 0.f, true);
    if (rc) return rc;
    batch_begun = true;
    if (!keep_caches) std::fill(valid_rt.begin(), valid_rt.end(), 0);  // caches dropped: reset_cache (:155-158)
    return ECAD_OK;
  }

  int step(int step_idx, const void* latents, const float* timestep, void* out, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (int rcw = weights_current()) return rcw;
    if (!batch_begun || !plan_valid) return fail(ECAD_E_NOT_READY, "ecad_begin_image_batch must be called before ecad_step");
    if (step_idx < 0 || step_idx >= steps) return fail(ECAD_E_INVALID, "step_idx %d outside [0, %d)", step_idx, steps);
    if (!latents || !timestep || !out) return fail(ECAD_E_INVALID, "NULL pointer argument");
    call_latents = latents; call_timestep = timestep; call_out = out;
    cur_step_idx = step_idx;
    // what executes now = flag or no usable cache (cached_transformer_block.py:340-347), from the caches as they are at this call
    std::vector<uint8_t> ex((size_t)L * 3);
    bool as_planned = true;
    for (int i = 0; i < L * 3; ++i) {
      const size_t k = (size_t)step_idx * L * 3 + i;
      ex[i] = flag_at(step_idx, i) || !valid_rt[i];
      if (!(any_custom && custom_mask[i]) && ex[i] != executed[k]) as_planned = false;
    }
    if (as_planned) {
      int rc = run_ops(plan_step[step_idx], s, 0.f, true);
      if (rc) return rc;
      for (int i = 0; i < L * 3; ++i)
        if (ex[i]) valid_rt[i] = 1;
      return ECAD_OK;
    }
    // the caller reset the caches mid-run (or stepped out of order): lower this one step from the live cache state
    std::vector<Op<T>> ops;
    std::vector<SrcDesc> keep_desc;
    keep_desc.swap(desc_host);
    const EcadStats keep_st = st;
    emit_temb(ops, nullptr, Bt, (size_t)step_idx * Bt, true);
    emit_transformer(ops, step_idx, mod_all + (size_t)step_idx * Bt * 6 * D, 6 * D, temb + (size_t)step_idx * Bt * D, D, true, ex.data());
    std::vector<SrcDesc> dyn;
    dyn.swap(desc_host);
    desc_host.swap(keep_desc);
    st = keep_st;
    if (!dyn.empty()) {
      const size_t region = 8192;  // descriptors per ad-hoc plan (3L + a few): a ring of regions, so plans still in flight keep theirs
      if (dyn.size() > region) return fail(ECAD_E_UNSUPPORTED, "ad-hoc step plan needs %zu descriptors", dyn.size());
      if (!desc_dyn_dev) { desc_dyn_cap = region * 8; HIPCHK(hipMalloc((void**)&desc_dyn_dev, desc_dyn_cap * sizeof(SrcDesc))); }
      if (desc_dyn_pos + region > desc_dyn_cap) { HIPCHK(hipStreamSynchronize(s)); desc_dyn_pos = 0; }
      SrcDesc* base = desc_dyn_dev + desc_dyn_pos;
      desc_dyn_pos += region;
      HIPCHK(hipMemcpy(base, dyn.data(), dyn.size() * sizeof(SrcDesc), hipMemcpyHostToDevice));
      for (auto& o : ops)
        if (o.desc_off >= 0) { if (o.kind == Op<T>::GEMM) o.g.xsrc = base + o.desc_off; else o.f.srcs = base + o.desc_off; }
    }
    ++dynamic_steps;
    int rc = run_ops(ops, s, 0.f, true);
    if (rc) return rc;
    for (int i = 0; i < L * 3; ++i) if (ex[i]) valid_rt[i] = 1;
    return ECAD_OK;
  }

  int run_schedule(int Bt_, int H_, int W_, const void* enc, const uint8_t* mask, int T_, float* latents, const float* ts, const float* coef,
                   float guidance, int use_graph, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (Bt_ % 2) return fail(ECAD_E_INVALID, "run_schedule needs a CFG-doubled batch (Bt even), got %d", Bt_);
    if (any_custom) return fail(ECAD_E_UNSUPPORTED, "the schedule routes components to host compute plug-ins: the whole-loop graph cannot call back into the host, use ecad_step");
    if (!enc || !latents || !ts || !coef) return fail(ECAD_E_INVALID, "NULL pointer argument");
    int rc = ensure_shapes(Bt_, H_, W_, T_);
    if (rc) return rc;
    if (cfg.use_additional_conditions && (cond_bt != Bt_ || !cond_uniform))
      return fail(cond_bt != Bt_ ? ECAD_E_NOT_READY : ECAD_E_INVALID,
                  "ecad_run_schedule shares one adaLN table per step across the batch: call ecad_set_added_cond for batch %d with one (resolution, aspect_ratio) for all samples, as the pipeline does", Bt_);
    // per-schedule constants (timesteps, solver coefficients): uploaded synchronously, outside the graph
    std::vector<float> c8((size_t)steps * 8, 0.f), tsv(ts, ts + steps);
    for (int i = 0; i < steps; ++i) for (int j = 0; j < 5; ++j) c8[(size_t)i * 8 + j] = coef[i * 5 + j];
    if (c8 != coef_host || tsv != ts_host) {  // constants of the schedule: uploaded once, no per-batch sync
      HIPCHK(hipMemcpy(ts_dev, tsv.data(), steps * sizeof(float), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(coef_dev, c8.data(), c8.size() * sizeof(float), hipMemcpyHostToDevice));
      coef_host = c8; ts_host = tsv;
    }
    rc = stage_inputs(enc, mask, s);
    if (rc) return rc;
    if (!fold_descs.empty() && (!fold_built || fold_ts != tsv)) {
      // once per (plan set, timestep grid): the adaLN tables and comb vectors of all steps (the same ops the graph replays every batch), then
      // W' = W (.) (1 + scale), c1, c2 for every folded (step, block, component) in one launch
      std::vector<Op<T>> prep(plan_run.begin() + fold_prep_begin, plan_run.begin() + fold_prep_begin + fold_prep_ops);
      rc = run_ops(prep, s, 0.f, false);
      if (rc) return rc;
      int max_rows = 0;
      for (auto& d : fold_descs) max_rows = d.rows > max_rows ? d.rows : max_rows;
      hipLaunchKernelGGL((ln_fold_weights_kernel<T>), dim3(max_rows, (unsigned)fold_descs.size()), dim3(256), 0, s,
                         reinterpret_cast<const FoldDesc*>(fold_cur.base + fold_used), D);
      HIPCHK(hipGetLastError());
      fold_built = true;
      fold_ts = tsv;
      ++fold_preps;
    }
    const size_t nlat = (size_t)(Bt / 2) * CIN * Hl * Wl;
    HIPCHK(hipMemcpyAsync(lat_buf, latents, nlat * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (use_graph && !profiling) {
      if (!graph_exec || graph_guidance != guidance) {
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
        HIPCHK(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
        rc = run_ops(plan_run, cap_stream, guidance, false);
        hipError_t e = hipStreamEndCapture(cap_stream, &graph);
        if (rc) return rc;
        if (e != hipSuccess) return fail(ECAD_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        HIPCHK(hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
        ++graph_captures;
        size_t nn = 0;
        HIPCHK(hipGraphGetNodes(graph, nullptr, &nn));
        st.graph_nodes = (int)nn;
        graph_guidance = guidance;
      }
      HIPCHK(hipGraphLaunch(graph_exec, s));
    } else {
      rc = run_ops(plan_run, s, guidance, true);
      if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(latents, lat_buf, nlat * sizeof(float), hipMemcpyDeviceToDevice, s));
    batch_begun = false;  // caches are dropped after the last step (image_generator.py:197-202)
    std::fill(valid_rt.begin(), valid_rt.end(), 0);
    return ECAD_OK;
  }

  // reset_cache (:155-158) drops the three cached outputs of every block and nothing else: the hoisted caption projection / text K/V of the batch are
  // recomputations, not caches, so a batch that has begun stays begun and the next ecad_step recomputes whatever it finds no usable cache for
  int reset_cache() override { std::fill(valid_rt.begin(), valid_rt.end(), 0); return ECAD_OK; }
  int set_custom(const uint8_t* mask, int blocks, EcadComponentFn fn, void* user, void* in, void* out) override {
    if (mask && blocks != L) return fail(ECAD_E_INVALID, "plug-in mask has %d blocks, model has %d", blocks, L);
    bool any = false;
    for (int i = 0; mask && i < L * 3; ++i) any = any || mask[i];
    if (any && (!fn || !in || !out)) return fail(ECAD_E_INVALID, "a compute plug-in needs a callback and the two [max_batch * max_tokens, D] I/O buffers");
    stash_current();
    custom_mask.assign((size_t)L * 3, 0);
    if (any) custom_mask.assign(mask, mask + (size_t)L * 3);
    any_custom = any; custom_fn = any ? fn : nullptr; custom_user = user; io_in = (T*)in; io_out = (T*)out;
    return ECAD_OK;
  }
  // the block's own sub-layers for use inside a plug-in (the reference's `self.attn1(...)` / `self.attn2(...)` / `self.ff(...)`): plain bias epilogues,
  // no gate, no cache; geometry and text K/V of the batch that has begun.  fp16 / bf16 kernels also in fp8 mode.
  int block_attn(int block, int which, const void* x, void* out, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (!batch_begun || !plan_valid) return fail(ECAD_E_NOT_READY, "ecad_begin_image_batch must be called first");
    if (block < 0 || block >= L || (which != 1 && which != 2) || !x || !out) return fail(ECAD_E_INVALID, "bad block %d / attention %d / NULL pointer", block, which);
    const Blk& B = blk[block];
    const int M = Bt * N;
    AttnArgs a{};
    a.out = attn_o; a.ldo = D; a.B = Bt; a.H = NH; a.Nq = N; a.dh = DH; a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)DH));
    int rc;
    if (which == 1) {
      if ((rc = launch_gemm(EPI_BIAS, KC_GEMM_QKV, mk_gemm(KC_GEMM_QKV, EPI_BIAS, (const T*)x, B.w_qkv, B.b_qkv, qkv, M, 3 * D, D, D, 3 * D).g, s))) return rc;
      a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D; a.Nk = N; a.bias = self_bias_on ? self_bias_buf : nullptr;
      if ((rc = launch_attn(a, (N % 128) != 0 || self_bias_on, s))) return rc;
      if ((rc = launch_gemm(EPI_BIAS, KC_GEMM_SMALL, mk_gemm(KC_GEMM_SMALL, EPI_BIAS, attn_o, B.w_o1, B.b_o1, (T*)out, M, D, D, D, D).g, s))) return rc;
    } else {
      if ((rc = launch_gemm(EPI_BIAS, KC_GEMM_Q2, mk_gemm(KC_GEMM_Q2, EPI_BIAS, (const T*)x, B.w_q2, B.b_q2, qc, M, D, D, D, D).g, s))) return rc;
      a.q = qc; a.ldq = D; a.k = kv2 + (size_t)block * 2 * D; a.v = kv2 + (size_t)block * 2 * D + D; a.ldk = a.ldv = L * 2 * D; a.Nk = Tt; a.bias = bias_buf;
      if ((rc = launch_attn(a, true, s))) return rc;
      if ((rc = launch_gemm(EPI_BIAS, KC_GEMM_SMALL, mk_gemm(KC_GEMM_SMALL, EPI_BIAS, attn_o, B.w_o2, B.b_o2, (T*)out, M, D, D, D, D).g, s))) return rc;
    }
    HIPCHK(hipGetLastError());
    return ECAD_OK;
  }
  int block_ff(int block, const void* x, void* out, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (!batch_begun || !plan_valid) return fail(ECAD_E_NOT_READY, "ecad_begin_image_batch must be called first");
    if (block < 0 || block >= L || !x || !out) return fail(ECAD_E_INVALID, "bad block %d / NULL pointer", block);
    const Blk& B = blk[block];
    const int M = Bt * N;
    int rc;
    if ((rc = launch_gemm(EPI_BIAS_GELU, KC_GEMM_FF1, mk_gemm(KC_GEMM_FF1, EPI_BIAS_GELU, (const T*)x, B.w_ff1, B.b_ff1, ffmid, M, 4 * D, D, D, 4 * D).g, s))) return rc;
    if ((rc = launch_gemm(EPI_BIAS, KC_GEMM_SMALL, mk_gemm(KC_GEMM_SMALL, EPI_BIAS, ffmid, B.w_ff2, B.b_ff2, (T*)out, M, D, 4 * D, 4 * D, D).g, s))) return rc;
    HIPCHK(hipGetLastError());
    return ECAD_OK;
  }
  int copy_context(void* ctx_out, float* bias_out, int* Bt_, int* N_, int* T_, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (!batch_begun || !plan_valid) return fail(ECAD_E_NOT_READY, "ecad_begin_image_batch must be called first");
    if (ctx_out) HIPCHK(hipMemcpyAsync(ctx_out, ctx, (size_t)Bt * Tt * D * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (bias_out) HIPCHK(hipMemcpyAsync(bias_out, bias_buf, (size_t)Bt * Tt * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (Bt_) *Bt_ = Bt;
    if (N_) *N_ = N;
    if (T_) *T_ = Tt;
    return ECAD_OK;
  }
  int stats(EcadStats* s) override {
    *s = st;
    s->plan_builds = plan_builds; s->graph_captures = graph_captures; s->dynamic_step_plans = dynamic_steps; s->plan_sets_cached = (int32_t)plan_lru.size();
    s->ln_folded = ln_folded; s->fold_preps = fold_preps; s->fold_bytes = (int64_t)fold_used;
    s->fold_arena_bytes = (int64_t)fold_held(); s->fold_refused = fold_refused;
    s->rmw_launches_wide = rmw_launches_wide; s->rmw_launches_narrow4 = rmw_launches_narrow4; s->rmw_launches_gemm3s = rmw_launches_g3s;
    s->split_k = split_k ? 1 : 0; s->batches_in_flight = in_flight;
    return ECAD_OK;
  }
  int set_profiling(int on) override {
    profiling = on != 0;
    for (int i = 0; i < KC_COUNT; ++i) { prof_ms[i] = prof_flops[i] = prof_bytes[i] = 0; prof_launch[i] = 0; }
    return ECAD_OK;
  }
  int get_profile(int maxc, const char** names, double* ms, int64_t* launches, double* flops, double* bytes) override {
    int n = 0;
    for (int i = 0; i < KC_COUNT && n < maxc; ++i) {
      names[n] = kclass_names[i]; ms[n] = prof_ms[i]; launches[n] = prof_launch[i]; flops[n] = prof_flops[i]; bytes[n] = prof_bytes[i];
      ++n;
    }
    return n;
  }

  // -------------------------------------------------------------------------------------------
  // single-op entry points
  int op_gemm(int epi, const void* A, const void* W, const float* bias, void* out, int M, int Nn, int K, void* resid, const float* gate_table,
              const void* gate_mod, int64_t gate_mod_stride, int rows_per_batch, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    GemmArgs<T> g{};
    g.A = (const T*)A; g.W = (const T*)W; g.bias = bias; g.C = (T*)out; g.resid = (T*)resid; g.gate_table = gate_table; g.gate_mod = (const T*)gate_mod;
    g.gate_mod_stride = gate_mod_stride; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; g.M = M; g.N = Nn; g.K = K; g.lda = K; g.ldw = K; g.ldc = Nn; g.ldr = Nn;
    g.variant = 1; g.epi_direct = 1;
    if (epi == EPI_BIAS_CACHE_RESID && !resid) return fail(ECAD_E_INVALID, "epilogue 3 needs resid");
    int rc = launch_gemm(epi, KC_GEMM_SMALL, g, s);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return ECAD_OK;
  }
  // The kernel is named by the caller: the per-class choice table is overridden for this one launch and restored, so the instantiation is
  // exactly the one the engine's call site `site` launches.  Refuses (never substitutes) when the kernel cannot run the shape.
  int op_qproj_attention(int site, const void* x, const void* wq, const float* bq, const void* k, const void* v, int64_t ld_kv, const float* key_bias, void* out, int Bt_,
                         int n_tokens, int n_keys, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    const bool tail = site == KC_GEMM_Q2;
    if (site != KC_GEMM_QKV && !tail) return fail(ECAD_E_INVALID, "call_site %d: the fused launch exists for the self- (0) and cross-attention (2) query projections", site);
    if (DH != G3A_DH || D % G3_BN || n_tokens % G3_BM || Bt_ < 1 || n_keys < 1 || (!tail && (n_keys != n_tokens || n_keys % G3A_KC)))
      return fail(ECAD_E_UNSUPPORTED, "fused query projection + attention needs d_h 72, n_tokens %% 256 == 0 (self: keys == tokens); got d_h %d, %d tokens, %d keys", DH, n_tokens, n_keys);
    // the K / V staging reads 16-byte chunks (gload16): base pointers on 16 bytes, row stride a multiple of 8 elements
    if (!x || !wq || !k || !v || !out || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)x & 15) || ((uintptr_t)wq & 15) || ((uintptr_t)out & 15) || ld_kv < D || ld_kv % 8)
      return fail(ECAD_E_INVALID, "fused query projection + attention: x / wq / k / v / out must be 16-byte aligned and ld_kv (%lld) a multiple of 8 >= %d", (long long)ld_kv, D);
    GemmArgs<T> g{};
    g.A = (const T*)x; g.W = (const T*)wq; g.bias = bq; g.C = nullptr; g.M = Bt_ * n_tokens; g.N = D; g.K = D; g.lda = D; g.ldw = D; g.ldc = D; g.ldr = D;
    g.rows_per_batch = n_tokens; g.variant = 1; g.epi_direct = 1;
    g.att_k = (const T*)k; g.att_v = (const T*)v; g.att_out = (T*)out; g.att_bias = tail ? key_bias : nullptr; g.att_ldk = (int)ld_kv; g.att_ldo = D; g.att_nk = n_keys;
    g.att_scale_log2e = (float)(1.4426950408889634 / sqrt((double)DH));
    const int rc = launch_gemm(tail ? EPI_BIAS_ATTN_TAIL : EPI_BIAS_ATTN, site, g, s);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return ECAD_OK;
  }
  int op_gemm_ex(int kernel, int site, int epi, const void* A, const void* W, const float* bias, void* out, int M, int Nn, int K, int64_t lda, int64_t ldc, void* resid,
                 const float* gate_table, const void* gate_mod, int64_t gate_mod_stride, int rows_per_batch, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (site < 0 || site > KC_GEMM_KV1) return fail(ECAD_E_INVALID, "call_site %d is not a GEMM class", site);
    if (epi == EPI_BIAS_CACHE_RESID && !resid) return fail(ECAD_E_INVALID, "epilogue 3 needs resid");
    GemmArgs<T> g{};
    g.A = (const T*)A; g.W = (const T*)W; g.bias = bias; g.C = (T*)out; g.resid = (T*)resid; g.gate_table = gate_table; g.gate_mod = (const T*)gate_mod;
    g.gate_mod_stride = gate_mod_stride; g.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; g.M = M; g.N = Nn; g.K = K;
    g.lda = lda > 0 ? (int)lda : K; g.ldw = K; g.ldc = ldc > 0 ? (int)ldc : Nn; g.ldr = Nn;
    g.variant = (kernel >> 8) ? ((kernel >> 8) - 1) & 3 : 1;  // bits 8+ of `kernel`: 1 + schedule variant to force (A/B in one process); 0 = the engine's
    g.epi_direct = (kernel >> 8) && (((kernel >> 8) - 1) & 12) ? ((((kernel >> 8) - 1) & 4) ? 2 : 0) : 1;  // + 4: direct epilogue forced on, + 8: forced off
    kernel &= 0xff;
    if (g.K % GEMM_BK || g.M < 1) return fail(ECAD_E_INVALID, "gemm shape M=%d N=%d K=%d unsupported (K%%64)", g.M, g.N, g.K);
    if (kernel == 3) {
      if (Nn % 128) return fail(ECAD_E_UNSUPPORTED, "gemm.h tiles need N %% 128 == 0, got %d", Nn);
      switch (epi) {
        case EPI_BIAS: launch_gemm_epi<EPI_BIAS>(g, s); break;
        case EPI_BIAS_GELU: launch_gemm_epi<EPI_BIAS_GELU>(g, s); break;
        case EPI_BIAS_SILU: launch_gemm_epi<EPI_BIAS_SILU>(g, s); break;
        case EPI_BIAS_CACHE_RESID: launch_gemm_epi<EPI_BIAS_CACHE_RESID>(g, s); break;
        default: return fail(ECAD_E_INVALID, "bad epilogue %d", epi);
      }
      HIPCHK(hipGetLastError());
      return ECAD_OK;
    }
    if (kernel == 4) return fail(ECAD_E_UNSUPPORTED, "the 128 x 144 half-tile variant was measured slower at every call site (DESIGN 5c) and removed in round 4");
    if (kernel == 5 && epi != EPI_BIAS && epi != EPI_BIAS_GELU && epi != EPI_BIAS_SILU && epi != EPI_BIAS_CACHE_RESID) return fail(ECAD_E_UNSUPPORTED, "the latency tiles have no attention tail through this entry");
    if (kernel == 1) return fail(ECAD_E_UNSUPPORTED, "the persistent producer / consumer kernel (pgemm.h) was retired in round 3: no call site used it");
    if (kernel == 6 && epi != EPI_BIAS_CACHE_RESID) return fail(ECAD_E_UNSUPPORTED, "split-K exists for the read-modify-write epilogue only");
    const bool narrow4_forced = kernel == 7;
    if (narrow4_forced) {
      if (epi != EPI_BIAS_CACHE_RESID || (site != KC_GEMM_OUT1 && site != KC_GEMM_OUT2 && site != KC_GEMM_FF2))
        return fail(ECAD_E_UNSUPPORTED, "gemm3s4 exists for the read-modify-write epilogue at both out-projections and FF2");
      kernel = 0;
    }
    if ((kernel < 0 || kernel > 2) && kernel != 5 && kernel != 6) return fail(ECAD_E_INVALID, "kernel %d unknown", kernel);
    if (Nn % G3_BN) return fail(ECAD_E_UNSUPPORTED, "256x144 / 256x288 tiles need N %% 144 == 0, got %d", Nn);
    // kernel 2 at a read-modify-write site = that epilogue on the wide tile (gemm2w4.h), whatever the tile count and the batches in flight say
    const bool rmw_wide = kernel == 2 && epi == EPI_BIAS_CACHE_RESID && (site == KC_GEMM_OUT1 || site == KC_GEMM_OUT2 || site == KC_GEMM_FF2);
    if (kernel == 2) {
      if (Nn % G2_BN) return fail(ECAD_E_UNSUPPORTED, "gemm2w tiles need N %% 288 == 0, got %d", Nn);
      if (rmw_wide) {
        if (!g2w4_shape_ok(g) || !g2w4_rmw_ok(g)) return fail(ECAD_E_UNSUPPORTED, "the read-modify-write epilogue on the wide tile needs whole 256-row tiles inside one sample each and K %% 128 == 0");
      } else if (site != KC_GEMM_FF1 && site != KC_GEMM_TEXTKV && site != KC_GEMM_SMALL && site != KC_GEMM_KV1)
        return fail(ECAD_E_UNSUPPORTED, "gemm2w is instantiated for the FF1 / text-KV / attn1-KV / small call sites (and, with the read-modify-write epilogue, both out-projections and FF2) only");
    }
    const int keep = gemm_impl_cls[site];
    const bool keep_lat = latency_tiles, keep_wide = wide_rmw;
    const int keep_force = force_wide_rmw;
    const int keep_n4 = narrow4;
    gemm_impl_cls[site] = rmw_wide ? 0 : kernel;
    latency_tiles = false;  // the caller named the kernel: no few-rows substitution (kernel 5 IS the latency tile)
    wide_rmw = rmw_wide;    // ... and no wide-tile substitution either
    force_wide_rmw = rmw_wide;
    force_narrow4 = narrow4_forced;
    if (!narrow4_forced) narrow4 = 0;  // ... nor the four-wave 256 x 144 kernel unless it was the one named
    const bool keep_tail = tail_split;
    if (narrow4_forced) tail_split = false;
    const int rc = launch_gemm(epi, site, g, s);
    gemm_impl_cls[site] = keep;
    latency_tiles = keep_lat;
    wide_rmw = keep_wide;
    force_wide_rmw = keep_force;
    force_narrow4 = false;
    narrow4 = keep_n4;
    tail_split = keep_tail;
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return ECAD_OK;
  }
  int op_attention_ex(int kernel, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo, int B, int H, int Nq, int Nk,
                      int dh, const float* bias, hipStream_t s) override {
    if (kernel < 0 || kernel > 4) return fail(ECAD_E_INVALID, "attention kernel %d unknown", kernel);
    if (kernel >= 2 && dh != 72) return fail(ECAD_E_UNSUPPORTED, "attention128.h / attention_dma.h are instantiated for head_dim 72 on this path, got %d", dh);
    if (kernel == 4) {  // attention_w472.h: what the plans launch for self-attention over >= 1024 tokens (round 6)
      if (bias || Nk % 256 || Nq % 256) return fail(ECAD_E_UNSUPPORTED, "attention_w472 needs Nq %% 256 == 0, Nk %% 256 == 0 and no key bias");
      HIPCHK(hipSetDevice(dev));
      AttnArgs a{};
      a.q = q; a.k = k; a.v = v; a.out = out; a.bias = nullptr; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.dh = dh;
      a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)dh));
      const bool keep_on = attn_w472_on;
      attn_w472_on = true;
      const int rc = launch_attn(a, false, s, true);
      attn_w472_on = keep_on;
      if (rc) return rc;
      HIPCHK(hipGetLastError());
      return ECAD_OK;
    }
    if (kernel == 3) {  // attention_dma.h with four query fragments per wave: the plans' kernel for self-attention over >= 1024 tokens until round 6 (ECAD_ATTN_W472=0)
      if (bias || Nk % 128 || Nq % 512) return fail(ECAD_E_UNSUPPORTED, "the four-fragment kernel needs Nq %% 512 == 0, Nk %% 128 == 0 and no key bias");
      HIPCHK(hipSetDevice(dev));
      AttnArgs a{};
      a.q = q; a.k = k; a.v = v; a.out = out; a.bias = nullptr; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.dh = dh;
      a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)dh));
      const int rc = launch_attn(a, false, s, true, true);
      if (rc) return rc;
      HIPCHK(hipGetLastError());
      return ECAD_OK;
    }
    const int keep = attn_shared;
    if (kernel) attn_shared = kernel == 2;
    const int rc = op_attention(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Nq, Nk, dh, bias, s);
    attn_shared = keep;
    return rc;
  }
  int op_ln(const void* x, void* y, int M, int Dd, const float* tsh, const float* tsc, const void* msh, const void* msc, int64_t mstride, int rows_per_batch,
            hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    FlushArgs f{};
    f.h = const_cast<void*>(x); f.y = y; f.srcs = nullptr; f.n_src = 0; f.M = M; f.D = Dd; f.mod_stride = mstride; f.ln_mod_stride = mstride;
    f.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; f.table_shift = tsh; f.table_scale = tsc; f.mod_shift = msh; f.mod_scale = msc; f.eps = cfg.norm_eps;
    int rc = launch_ln(f, s);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return ECAD_OK;
  }
  int op_accumulate(void* hh, int M, int Dd, int n_src, const void* const* srcs, const float* const* gts, const void* const* gms, int64_t mstride,
                    int rows_per_batch, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (n_src < 0 || n_src > 4096) return fail(ECAD_E_INVALID, "n_src out of range");
    std::vector<SrcDesc> d(n_src);
    for (int i = 0; i < n_src; ++i) d[i] = SrcDesc{srcs[i], gts ? gts[i] : nullptr, gms ? gms[i] : nullptr};
    SrcDesc* dd = nullptr;
    HIPCHK(hipMalloc((void**)&dd, sizeof(SrcDesc) * (n_src + 1)));
    {
      const hipError_t ce = hipMemcpy(dd, d.data(), sizeof(SrcDesc) * n_src, hipMemcpyHostToDevice);
      if (ce != hipSuccess) { (void)hipFree(dd); return fail(ECAD_E_HIP, "hipMemcpy of the replay descriptors failed: %s", hipGetErrorString(ce)); }
    }
    FlushArgs f{};
    f.h = hh; f.srcs = dd; f.n_src = n_src; f.M = M; f.D = Dd; f.mod_stride = mstride; f.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    launch_replay(f, s);
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(dd);
    if (e != hipSuccess) return fail(ECAD_E_HIP, "cache_accumulate failed: %s", hipGetErrorString(e));
    return ECAD_OK;
  }
  int op_attention(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo, int B, int H, int Nq, int Nk,
                   int dh, const float* bias, hipStream_t s) override {
    HIPCHK(hipSetDevice(dev));
    if (dh % 8 || dh > 72) return fail(ECAD_E_UNSUPPORTED, "head_dim %d unsupported", dh);
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.out = out; a.bias = bias; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.dh = dh;
    a.scale_log2e = (float)(1.4426950408889634 / sqrt((double)dh));
    int rc = launch_attn(a, bias != nullptr, s);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return ECAD_OK;
  }
};

}  // namespace ecad

// ================================================================================================
using namespace ecad;
struct EcadEngine { EngineBase* impl; };

extern "C" {

int ecad_abi_version(void) { return ECAD_ABI_VERSION; }

int ecad_create(const EcadModelCfg* cfg, int device, EcadEngine** out) {
  if (!cfg || !out) { g_create_error = "NULL argument"; return ECAD_E_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_error = "no HIP device visible"; return ECAD_E_HIP; }
  if (device < 0 || device >= ndev) { g_create_error = "device index out of range"; return ECAD_E_INVALID; }
  EngineBase* impl = nullptr;
  if (cfg->dtype == ECAD_DTYPE_F16) impl = new EngineT<f16_t>();
  else if (cfg->dtype == ECAD_DTYPE_BF16) impl = new EngineT<bf16_t>();
  else { g_create_error = "dtype must be ECAD_DTYPE_F16 or ECAD_DTYPE_BF16"; return ECAD_E_INVALID; }
  impl->cfg = *cfg;
  impl->dev = device;
  int rc = impl->init();
  if (rc) { g_create_error = impl->err; delete impl; return rc; }
  *out = new EcadEngine{impl};
  return ECAD_OK;
}
void ecad_destroy(EcadEngine* e) { if (e) { delete e->impl; delete e; } }
const char* ecad_last_error(const EcadEngine* e) { return e ? e->impl->err.c_str() : g_create_error.c_str(); }

#define ENG(e) if (!(e)) return ECAD_E_INVALID; EngineBase* E = (e)->impl

int ecad_load_weights(EcadEngine* e, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) { ENG(e); if (!name || !ptr || !shape) return E->fail(ECAD_E_INVALID, "NULL argument"); return E->load_weight(name, ptr, dtype, shape, ndim); }
int ecad_finalize_weights(EcadEngine* e) { ENG(e); return E->finalize_weights(); }
int ecad_set_ln_fold(EcadEngine* e, int on) { ENG(e); return E->set_ln_fold(on); }
int ecad_set_batches_in_flight(EcadEngine* e, int n) { ENG(e); return E->set_batches_in_flight(n); }
int ecad_set_split_k(EcadEngine* e, int on) { ENG(e); return E->set_split_k(on); }
int ecad_set_native_variants(EcadEngine* e, const uint8_t* kinds, int steps, int blocks) { ENG(e); return E->set_native(kinds, steps, blocks); }
int ecad_share_weights(EcadEngine* e, EcadEngine* donor) {
  ENG(e);
  if (!donor) return E->fail(ECAD_E_INVALID, "NULL donor");
  return E->share_weights(donor->impl);
}
int ecad_set_fp8(EcadEngine* e, int on) { ENG(e); return E->set_fp8(on); }
int ecad_set_added_cond(EcadEngine* e, const float* resolution, const float* aspect_ratio, int Bt) { ENG(e); return E->set_added_cond(resolution, aspect_ratio, Bt); }
int ecad_set_attention_bias(EcadEngine* e, const float* self_bias_dev, const float* enc_bias_dev, int Bt, int N, int T, void* stream) { ENG(e); return E->set_attention_bias(self_bias_dev, enc_bias_dev, Bt, N, T, (hipStream_t)stream); }
int ecad_set_schedule(EcadEngine* e, const uint8_t* bits, int steps, int blocks) { ENG(e); if (!bits) return E->fail(ECAD_E_INVALID, "NULL bits"); return E->set_schedule(bits, steps, blocks); }
int ecad_begin_image_batch(EcadEngine* e, int Bt, int H, int W, const void* enc, const uint8_t* mask, int T, void* stream) { ENG(e); return E->begin_batch(Bt, H, W, enc, mask, T, (hipStream_t)stream); }
int ecad_rebind_batch_inputs(EcadEngine* e, int Bt, int H, int W, const void* enc, const uint8_t* mask, int T, void* stream) { ENG(e); return E->begin_batch(Bt, H, W, enc, mask, T, (hipStream_t)stream, true); }
int ecad_step(EcadEngine* e, int step_idx, const void* lat, const float* ts, void* out, void* stream) { ENG(e); return E->step(step_idx, lat, ts, out, (hipStream_t)stream); }
int ecad_run_schedule(EcadEngine* e, int Bt, int H, int W, const void* enc, const uint8_t* mask, int T, float* latents, const float* ts, const float* coef, float guidance, int use_graph, void* stream) {
  ENG(e); return E->run_schedule(Bt, H, W, enc, mask, T, latents, ts, coef, guidance, use_graph, (hipStream_t)stream);
}
int ecad_reset_cache(EcadEngine* e) { ENG(e); return E->reset_cache(); }
int ecad_set_custom_components(EcadEngine* e, const uint8_t* mask, int blocks, EcadComponentFn fn, void* user, void* io_in_dev, void* io_out_dev) { ENG(e); return E->set_custom(mask, blocks, fn, user, io_in_dev, io_out_dev); }
int ecad_block_attn(EcadEngine* e, int block, int which, const void* x_dev, void* out_dev, void* stream) { ENG(e); return E->block_attn(block, which, x_dev, out_dev, (hipStream_t)stream); }
int ecad_block_ff(EcadEngine* e, int block, const void* x_dev, void* out_dev, void* stream) { ENG(e); return E->block_ff(block, x_dev, out_dev, (hipStream_t)stream); }
int ecad_copy_batch_context(EcadEngine* e, void* ctx_out_dev, float* bias_out_dev, int* Bt, int* N, int* T, void* stream) { ENG(e); return E->copy_context(ctx_out_dev, bias_out_dev, Bt, N, T, (hipStream_t)stream); }
int ecad_cache_stats(EcadEngine* e, EcadStats* s) { ENG(e); if (!s) return E->fail(ECAD_E_INVALID, "NULL stats"); return E->stats(s); }
int ecad_set_profiling(EcadEngine* e, int on) { ENG(e); return E->set_profiling(on); }
int ecad_get_profile(EcadEngine* e, int maxc, const char** names, double* ms, int64_t* launches, double* flops, double* bytes) { ENG(e); return E->get_profile(maxc, names, ms, launches, flops, bytes); }
int ecad_op_gemm(EcadEngine* e, int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, void* resid, const float* gate_table, const void* gate_mod, int64_t gate_mod_stride, int rows_per_batch, void* stream) {
  ENG(e); return E->op_gemm(epi, A, W, bias, out, M, N, K, resid, gate_table, gate_mod, gate_mod_stride, rows_per_batch, (hipStream_t)stream);
}
int ecad_op_gemm_ex(EcadEngine* e, int kernel, int call_site, int epi, const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int64_t lda, int64_t ldc,
                    void* resid, const float* gate_table, const void* gate_mod, int64_t gate_mod_stride, int rows_per_batch, void* stream) {
  ENG(e); return E->op_gemm_ex(kernel, call_site, epi, A, W, bias, out, M, N, K, lda, ldc, resid, gate_table, gate_mod, gate_mod_stride, rows_per_batch, (hipStream_t)stream);
}
int ecad_op_qproj_attention(EcadEngine* e, int call_site, const void* x, const void* wq, const float* bq, const void* k, const void* v, int64_t ld_kv, const float* key_bias,
                            void* out, int Bt, int n_tokens, int n_keys, void* stream) {
  ENG(e); return E->op_qproj_attention(call_site, x, wq, bq, k, v, ld_kv, key_bias, out, Bt, n_tokens, n_keys, (hipStream_t)stream);
}
int ecad_op_attention_ex(EcadEngine* e, int kernel, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo, int B, int H, int Nq, int Nk, int dh, const float* bias, void* stream) {
  ENG(e); return E->op_attention_ex(kernel, q, ldq, k, ldk, v, ldv, out, ldo, B, H, Nq, Nk, dh, bias, (hipStream_t)stream);
}
int ecad_op_ln_modulate(EcadEngine* e, const void* x, void* y, int M, int D, const float* tsh, const float* tsc, const void* msh, const void* msc, int64_t mstride, int rows_per_batch, void* stream) {
  ENG(e); return E->op_ln(x, y, M, D, tsh, tsc, msh, msc, mstride, rows_per_batch, (hipStream_t)stream);
}
int ecad_op_cache_accumulate(EcadEngine* e, void* h, int M, int D, int n_src, const void* const* srcs, const float* const* gts, const void* const* gms, int64_t mstride, int rows_per_batch, void* stream) {
  ENG(e); return E->op_accumulate(h, M, D, n_src, srcs, gts, gms, mstride, rows_per_batch, (hipStream_t)stream);
}
int ecad_op_attention(EcadEngine* e, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* out, int64_t ldo, int B, int H, int Nq, int Nk, int dh, const float* bias, void* stream) {
  ENG(e); return E->op_attention(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Nq, Nk, dh, bias, (hipStream_t)stream);
}

}  // extern "C"
