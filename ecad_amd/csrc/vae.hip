// vae.hip -- AutoencoderKL decoder (SD / sd-vae-ft-ema geometry: block_out_channels 128,256,512,512, 3 resnets per up
// block, one mid attention, 32 groups) behind the C ABI: latents -> uint8 NHWC RGB, entirely in hand-written HIP.
// Replaces `self.vae.decode(latents / scaling_factor)` + `image_processor.postprocess` of the reference's timed region
// (ecad/pipelines/pass_through.py:382-396).  Weight names are diffusers' AutoencoderKL state-dict keys [ext].
#include "../../include/ecad_hip.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "vae_kernels.h"
#include "conv4.h"

namespace ecad {

static std::string g_vae_create_error;

#define VHIPCHK(expr)                                                                                                  \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess) return fail(ECAD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

template <typename TS, typename TD> __global__ void vae_convert_kernel(const TS* __restrict__ src, TD* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (TD)(float)src[i];
}
// [Cout][Cin][taps] -> [Cout][taps][Cin]
template <typename TS, typename TD> __global__ void vae_relayout_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int Cout, int Cin, int taps) {
  const size_t n = (size_t)Cout * Cin * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int tap = (int)((i / Cin) % taps);
    const int co = (int)(i / ((size_t)Cin * taps));
    dst[i] = (TD)(float)src[((size_t)co * Cin + ci) * taps + tap];
  }
}

struct VaeBase {
  int dev = 0;
  int latent_ch = 4;   // 4: SD / PixArt VAE; 16: FLUX VAE
  bool use_pq = true;  // post_quant_conv present (AutoencoderKL use_post_quant_conv; False for the FLUX VAE)
  std::string err;
  virtual ~VaeBase() {}
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
  virtual int load(const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) = 0;
  virtual int finalize() = 0;
  virtual int decode(const float* lat, int B, int h, int w, float scaling, float shift, uint8_t* out, hipStream_t s) = 0;
  // roofline accounting of the last decode: flops of its convolution GEMMs (2 M Cout taps Cin each) and how many of them ran on the 256 x 128 / 3-stage kernel
  double conv_flops = 0;
  double conv_flops_exec = 0;  // what the launches multiplied: below conv_flops when the upsampler convolutions ran in their four-phase 2 x 2 form (conv4.h)
  int conv_launches = 0, conv3s_launches = 0, conv4_launches = 0;
};

template <typename T> struct VaeT : VaeBase {
  enum Kind { F32, CONV3, LIN, CONVOUT };
  struct W { void* ptr; Kind kind; int cout, cin; size_t numel; bool loaded; };
  std::map<std::string, W> wt;
  std::map<std::string, void*> phase_w;  // upsampler convolutions: pre-summed 2 x 2 weights of the four output phases (conv4.h)
  std::map<std::string, void*> chunk_w, phase_chunk_w;  // the 3 x 3 / phase weights again in conv4_kernel's default K order: [Cout][Cin / 64][taps][64]
  bool korder_tap = false;    // env ECAD_VAE_KORDER=tap: conv4_kernel multiplies in conv3s_kernel's K order (kernel row, tap, channel) -- its bytes; the default fetches each input once
  std::vector<void*> allocs;
  void* zero_page = nullptr;
  // activation buffers
  T *bufA = nullptr, *bufB = nullptr, *bufC = nullptr, *bufD = nullptr;
  T* bufP = nullptr;  // zero-bordered input of the conv4_kernel launches: [B][H + 2][W + 2][C] (conv4.h)
  float *gn_sums = nullptr, *gn_scale = nullptr, *gn_shift = nullptr;
  size_t act_elems = 0, bufD_elems = 0;
  int cap_B = 0, cap_h = 0, cap_w = 0;
  bool finalized = false;
  bool async_failed = false;  // an asynchronous call inside a void helper failed: decode() turns it into ECAD_E_HIP
  int gn_blocks_cap = 0;  // per-block partial sums the GroupNorm statistics may write: sized by ensure_acts for the largest stage of the geometry it was given
  bool conv_tile128 = false;  // env ECAD_VAE_CONV=tile128: every convolution on the 128 x 128 kernel (A/B against conv3s_kernel)
  bool tile_linear = false;   // env ECAD_VAE_TILE=linear: conv4_kernel's M tiles as 256 consecutive pixels instead of 16 x 16 patches (A/B)
  bool up_nine = false;       // env ECAD_VAE_UP=nine: upsampler convolutions as nine taps over the materialised upsampled image (A/B against the phase form)
  bool gn_separate = false;   // env ECAD_VAE_GN=separate: GroupNorm statistics always by gn_stats_kernel (A/B and the byte-identity test against the conv3s path)
  const void* stats_of = nullptr;  // the tensor whose per-tile GroupNorm partial sums a conv4_kernel epilogue left in gn_sums (conv4.h), or null
  int stats_nblk = 0;
  bool conv_gather = false;   // env ECAD_VAE_CONV=conv3s: no conv4_kernel launches (the round-3 path: per-tap gathered A on eight waves; A/B and the byte-identity test)
  int num_cus = 256;

  ~VaeT() override {
    (void)hipSetDevice(dev);  // teardown: nothing to report an error to
    for (void* p : allocs) (void)hipFree(p);
    free_acts();
  }
  void free_acts() {
    for (void* p : {(void*)bufA, (void*)bufB, (void*)bufC, (void*)bufD, (void*)bufP, (void*)gn_sums, (void*)gn_scale, (void*)gn_shift})
      if (p) (void)hipFree(p);
    bufA = bufB = bufC = bufD = bufP = nullptr;
    gn_sums = gn_scale = gn_shift = nullptr;
  }
  int reg(const std::string& name, Kind kind, int cout, int cin) {
    size_t n = kind == CONV3 ? (size_t)cout * cin * 9 : kind == LIN ? (size_t)cout * cin : kind == CONVOUT ? (size_t)cout * cin * 9 : (size_t)cout * (cin > 0 ? cin : 1);
    void* p = nullptr;
    size_t bytes = n * ((kind == CONV3 || kind == LIN) ? sizeof(T) : sizeof(float));
    if (kind == CONVOUT) bytes = (size_t)16 * 9 * cin * sizeof(T);  // 3 real output rows + 13 zero rows: one MFMA N-fragment
    VHIPCHK(hipMalloc(&p, bytes));
    if (kind == CONVOUT) VHIPCHK(hipMemset(p, 0, bytes));
    allocs.push_back(p);
    wt[name] = W{p, kind, cout, cin, n, false};
    if (kind == CONV3 && cin % 64 == 0) {
      void* c = nullptr;
      VHIPCHK(hipMalloc(&c, bytes));
      allocs.push_back(c);
      chunk_w[name] = c;
    }
    return ECAD_OK;
  }
#define REG(...) { int rc_ = reg(__VA_ARGS__); if (rc_) return rc_; }
  int reg_resnet(const std::string& p, int cin, int cout) {
    REG(p + "norm1.weight", F32, cin, 0) REG(p + "norm1.bias", F32, cin, 0)
    REG(p + "conv1.weight", CONV3, cout, cin) REG(p + "conv1.bias", F32, cout, 0)
    REG(p + "norm2.weight", F32, cout, 0) REG(p + "norm2.bias", F32, cout, 0)
    REG(p + "conv2.weight", CONV3, cout, cout) REG(p + "conv2.bias", F32, cout, 0)
    if (cin != cout) { REG(p + "conv_shortcut.weight", LIN, cout, cin) REG(p + "conv_shortcut.bias", F32, cout, 0) }
    return ECAD_OK;
  }
  int init() override {
    VHIPCHK(hipSetDevice(dev));
    VHIPCHK(hipMalloc(&zero_page, 256));
    VHIPCHK(hipMemset(zero_page, 0, 256));
    allocs.push_back(zero_page);
    if (latent_ch != 4 && latent_ch != 16) return fail(ECAD_E_UNSUPPORTED, "latent_channels %d unsupported (4 or 16)", latent_ch);
    if (use_pq) { REG("post_quant_conv.weight", F32, latent_ch, latent_ch) REG("post_quant_conv.bias", F32, latent_ch, 0) }
    REG("decoder.conv_in.weight", F32, 512, latent_ch * 9) REG("decoder.conv_in.bias", F32, 512, 0)
    int rc;
    if ((rc = reg_resnet("decoder.mid_block.resnets.0.", 512, 512))) return rc;
    if ((rc = reg_resnet("decoder.mid_block.resnets.1.", 512, 512))) return rc;
    const std::string a = "decoder.mid_block.attentions.0.";
    REG(a + "group_norm.weight", F32, 512, 0) REG(a + "group_norm.bias", F32, 512, 0)
    for (const char* n : {"to_q", "to_k", "to_v", "to_out.0"}) { REG(a + n + ".weight", LIN, 512, 512) REG(a + n + ".bias", F32, 512, 0) }
    const int chans[4] = {512, 512, 256, 128};
    int prev = 512;
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 3; ++j)
        if ((rc = reg_resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".", j == 0 ? prev : chans[i], chans[i]))) return rc;
      if (i < 3) {
        const std::string un = "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv.";
        REG(un + "weight", CONV3, chans[i], chans[i]) REG(un + "bias", F32, chans[i], 0)
        void *ph = nullptr, *phc = nullptr;   // [4 phases][Cout][2][2][Cin]: filled by load() from the relaid-out nine-tap weight; and chunk-major
        VHIPCHK(hipMalloc(&ph, (size_t)16 * chans[i] * chans[i] * sizeof(T)));
        VHIPCHK(hipMalloc(&phc, (size_t)16 * chans[i] * chans[i] * sizeof(T)));
        allocs.push_back(ph);
        allocs.push_back(phc);
        phase_w[un + "weight"] = ph;
        phase_chunk_w[un + "weight"] = phc;
      }
      prev = chans[i];
    }
    REG("decoder.conv_norm_out.weight", F32, 128, 0) REG("decoder.conv_norm_out.bias", F32, 128, 0)
    REG("decoder.conv_out.weight", CONVOUT, 3, 128) REG("decoder.conv_out.bias", F32, 3, 0)
    VHIPCHK(hipFuncSetAttribute((const void*)conv_gemm_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    VHIPCHK(hipFuncSetAttribute((const void*)conv3s_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS_BYTES));
    VHIPCHK(hipFuncSetAttribute((const void*)conv4_kernel<T, 256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, conv4_lds_bytes<256>()));
    VHIPCHK(hipFuncSetAttribute((const void*)conv4_kernel<T, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, conv4_lds_bytes<128>()));
    VHIPCHK(hipFuncSetAttribute((const void*)conv4_kernel<T, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, conv4_lds_bytes<256>()));
    VHIPCHK(hipFuncSetAttribute((const void*)conv4_kernel<T, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, conv4_lds_bytes<128>()));
    if (const char* e = getenv("ECAD_VAE_TILE")) tile_linear = strcmp(e, "linear") == 0;
    if (const char* e = getenv("ECAD_VAE_KORDER")) korder_tap = strcmp(e, "tap") == 0;
    if (const char* e = getenv("ECAD_VAE_UP")) up_nine = strcmp(e, "nine") == 0;
    if (const char* e = getenv("ECAD_VAE_GN")) gn_separate = strcmp(e, "separate") == 0;
    if (const char* e = getenv("ECAD_VAE_CONV")) { conv_tile128 = strcmp(e, "tile128") == 0; conv_gather = strcmp(e, "conv3s") == 0; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) num_cus = prop.multiProcessorCount;
    return ECAD_OK;
  }

  int load(const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) override {
    VHIPCHK(hipSetDevice(dev));
    auto it = wt.find(name);
    if (it == wt.end()) {
      if (strncmp(name, "encoder.", 8) == 0 || strncmp(name, "quant_conv.", 11) == 0) return ECAD_OK;  // encoder half is unused
      if (!use_pq && strncmp(name, "post_quant_conv.", 16) == 0) return ECAD_OK;
      return fail(ECAD_E_INVALID, "unknown VAE weight '%s'", name);
    }
    W& w = it->second;
    int64_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    if ((size_t)numel != w.numel) return fail(ECAD_E_INVALID, "VAE weight '%s': expected %zu elements, got %lld", name, w.numel, (long long)numel);
    const size_t esz = dtype == ECAD_DTYPE_F32 ? 4 : 2;
    void* stage = nullptr;
    VHIPCHK(hipMalloc(&stage, numel * esz));
    hipError_t e = hipMemcpy(stage, ptr, numel * esz, hipMemcpyDefault);
    if (e != hipSuccess) { (void)hipFree(stage); return fail(ECAD_E_HIP, "copy of '%s' failed: %s", name, hipGetErrorString(e)); }
    const int grid = (int)((numel + 255) / 256 < 4096 ? (numel + 255) / 256 : 4096);
#define DISPATCH(KERNEL, DST, ...)                                                                                   \
  if (dtype == ECAD_DTYPE_F16) hipLaunchKernelGGL((KERNEL<f16_t, DST>), dim3(grid), dim3(256), 0, 0, (const f16_t*)stage, (DST*)w.ptr, __VA_ARGS__);   \
  else if (dtype == ECAD_DTYPE_BF16) hipLaunchKernelGGL((KERNEL<bf16_t, DST>), dim3(grid), dim3(256), 0, 0, (const bf16_t*)stage, (DST*)w.ptr, __VA_ARGS__); \
  else hipLaunchKernelGGL((KERNEL<float, DST>), dim3(grid), dim3(256), 0, 0, (const float*)stage, (DST*)w.ptr, __VA_ARGS__);
    if (w.kind == CONV3) {
      DISPATCH(vae_relayout_kernel, T, w.cout, w.cin, 9)
      auto ph = phase_w.find(name);
      if (ph != phase_w.end()) {
        hipLaunchKernelGGL((vae_phase_weights_kernel<T>), dim3(4096), dim3(256), 0, 0, (const T*)w.ptr, (T*)ph->second, w.cout, w.cin);
        hipLaunchKernelGGL((vae_chunk_major_kernel<T>), dim3(4096), dim3(256), 0, 0, (const T*)ph->second, (T*)phase_chunk_w.at(name), (size_t)4 * w.cout, 4, w.cin);
      }
      auto cw = chunk_w.find(name);
      if (cw != chunk_w.end()) hipLaunchKernelGGL((vae_chunk_major_kernel<T>), dim3(4096), dim3(256), 0, 0, (const T*)w.ptr, (T*)cw->second, (size_t)w.cout, 9, w.cin);
    }
    else if (w.kind == CONVOUT) { DISPATCH(vae_relayout_kernel, T, w.cout, w.cin, 9) }
    else if (w.kind == LIN) { DISPATCH(vae_convert_kernel, T, (size_t)numel) }
    else { DISPATCH(vae_convert_kernel, float, (size_t)numel) }
#undef DISPATCH
    e = hipDeviceSynchronize();
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(ECAD_E_HIP, "conversion of '%s' failed: %s", name, hipGetErrorString(e));
    w.loaded = true;
    return ECAD_OK;
  }
  int finalize() override {
    for (auto& kv : wt)
      if (!kv.second.loaded) return fail(ECAD_E_NOT_READY, "VAE weight '%s' was never loaded", kv.first.c_str());
    finalized = true;
    return ECAD_OK;
  }

  template <typename U> U* W_(const std::string& n) { return reinterpret_cast<U*>(wt.at(n).ptr); }

  int ensure_acts(int B, int h, int w) {
    if (B <= cap_B && h <= cap_h && w <= cap_w && bufA) return ECAD_OK;
    free_acts();
    // largest activation: output of the third upsampler: (8h x 8w) pixels x 256 channels
    act_elems = (size_t)B * (8 * h) * (8 * w) * 256;
    const size_t tokens = (size_t)h * w;
    const size_t attn_elems = (size_t)tokens * tokens;  // per-image score matrix lives in bufD
    VHIPCHK(hipMalloc((void**)&bufA, act_elems * sizeof(T)));
    VHIPCHK(hipMalloc((void**)&bufB, act_elems * sizeof(T)));
    VHIPCHK(hipMalloc((void**)&bufC, act_elems * sizeof(T)));
    bufD_elems = attn_elems > act_elems ? attn_elems : act_elems;
    VHIPCHK(hipMalloc((void**)&bufD, bufD_elems * sizeof(T)));
    VHIPCHK(hipMalloc((void**)&bufP, (size_t)B * (8 * h + 2) * (8 * w + 2) * 256 * sizeof(T)));
    // per-block partial (sum, sum of squares) of the 32 groups: the largest stage has 64 h w pixels at 512 per block (group_norm_stats), the small ones at most 64 blocks
    const size_t blocks_max = (64 * tokens + 255) / 256;   // gn_stats_kernel: 512 pixels per block at the largest stage; conv4_kernel's fused partials: one row per 256-pixel tile
    gn_blocks_cap = (int)(blocks_max > 64 ? blocks_max : 64);
    VHIPCHK(hipMalloc((void**)&gn_sums, (size_t)B * gn_blocks_cap * 64 * sizeof(float)));
    VHIPCHK(hipMalloc((void**)&gn_scale, (size_t)B * 512 * sizeof(float)));
    VHIPCHK(hipMalloc((void**)&gn_shift, (size_t)B * 512 * sizeof(float)));
    cap_B = B; cap_h = h; cap_w = w;
    return ECAD_OK;
  }

  // ---- layer launchers
  // which conv4_kernel tile a convolution takes (0: none -- conv3s_kernel / conv_gemm_kernel): whole 256-row tiles, whole trips of the generated K loop, and enough
  // tiles for the chip.  A 3 x 3 convolution on conv4_kernel reads the zero-bordered layout: its producer (GroupNorm apply or the upsample pass) asks here first.
  int conv4_bn(int M, int cin, int cout, int taps) const {
    if (conv_tile128 || conv_gather || M % 256 || cin % 64) return 0;
    if ((double)M * 1.13 * cin * sizeof(T) >= 4294967296.0) return 0;   // the statement's gathered A offsets are 32-bit (bordered input: at most 1.13 x the plain one at 32 x 32)
    const int stages = taps * cin / 64, tiles_m = M / 256;
    if (cout % 256 == 0 && stages % G4Period<T, 256>::value == 0 && tiles_m * (cout / 256) >= num_cus) return 256;
    if (cout % 128 == 0 && stages % G4Period<T, 128>::value == 0 && tiles_m * (cout / 128) >= num_cus) return 128;
    return 0;
  }
  int conv(const T* in, const std::string& name, T* out, const T* resid, int B, int H, int W, int cin, int cout, int taps, int up, hipStream_t s,
           const T* w_override = nullptr, const float* b_override = nullptr, bool no_bias = false, bool bordered = false) {
    if (cin % 64 || cout % 128) return fail(ECAD_E_UNSUPPORTED, "conv %s: Cin %% 64 / Cout %% 128 required (%d, %d)", name.c_str(), cin, cout);
    ConvArgs a{};
    a.in = in; a.w = w_override ? w_override : W_<T>(name + ".weight"); a.bias = no_bias ? nullptr : (b_override ? b_override : W_<float>(name + ".bias"));
    a.resid = resid; a.out = out; a.zero_page = zero_page; a.B = B; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout; a.taps = taps; a.up = up;
    const int M = B * H * W;
    conv_flops += 2.0 * M * (double)cout * taps * cin;
    conv_flops_exec += 2.0 * M * (double)cout * taps * cin;
    ++conv_launches;
    if (stats_of == out) stats_of = nullptr;
    if (const int bn = (up || (taps == 9) != bordered) ? 0 : conv4_bn(M, cin, cout, taps)) {
      const dim3 grid((M / 256) * (cout / bn));
      a.tile2d = taps == 9 && !tile_linear && H % 16 == 0 && W % 16 == 0;
      a.korder = korder_tap ? 1 : 0;
      if (taps == 9 && !korder_tap && !w_override) a.w = chunk_w.at(name + ".weight");
      // every 3 x 3 convolution's output goes through a GroupNorm next: its statistics come out of this epilogue (per-tile partial sums in tile order)
      if (taps == 9 && !gn_separate && (H * W) % 256 == 0 && (H * W) / 256 <= gn_blocks_cap && cout % 128 == 0) {
        a.gn_part = gn_sums;
        stats_of = out;
        stats_nblk = (H * W) / 256;
      }
      if (bn == 256 && resid) hipLaunchKernelGGL((conv4_kernel<T, 256, true>), grid, dim3(256), conv4_lds_bytes<256>(), s, a);
      else if (bn == 256) hipLaunchKernelGGL((conv4_kernel<T, 256, false>), grid, dim3(256), conv4_lds_bytes<256>(), s, a);
      else if (resid) hipLaunchKernelGGL((conv4_kernel<T, 128, true>), grid, dim3(256), conv4_lds_bytes<128>(), s, a);
      else hipLaunchKernelGGL((conv4_kernel<T, 128, false>), grid, dim3(256), conv4_lds_bytes<128>(), s, a);
      ++conv4_launches;
      return ECAD_OK;
    }
    if (bordered) return fail(ECAD_E_INVALID, "internal: conv %s was handed the bordered layout but does not run on conv4_kernel", name.c_str());
    // 256 x 128 tiles with the 3-stage K loop (conv3s_kernel) once they fill the chip; the 128 x 128 kernel for the small launches (mid-block attention, 8 x 8 latents)
    if (!conv_tile128 && ((M + C3_BM - 1) / C3_BM) * (cout / C3_BN) >= num_cus && (taps * cin) / 64 >= 4) {
      hipLaunchKernelGGL((conv3s_kernel<T>), dim3(((M + C3_BM - 1) / C3_BM) * (cout / C3_BN)), dim3(C3_NW * 64), C3_LDS_BYTES, s, a);
      ++conv3s_launches;
      return ECAD_OK;
    }
    const int tiles = ((M + 127) / 128) * (cout / 128);
    hipLaunchKernelGGL((conv_gemm_kernel<T>), dim3(tiles), dim3(256), 65536, s, a);
    return ECAD_OK;
  }
  // out[i] = in[i] w[i]^T for nb images (grid.y): the mid-block attention's score and value GEMMs on the 128 x 128 kernel
  int gemm_batched(const T* in, const T* w, T* out, int nb, int M, int K, int N, size_t bs_in, size_t bs_w, size_t bs_out, hipStream_t s) {
    if (K % 64 || N % 128) return fail(ECAD_E_UNSUPPORTED, "attention GEMM: K %% 64 / N %% 128 required (%d, %d)", K, N);
    ConvArgs a{};
    a.in = in; a.w = w; a.out = out; a.zero_page = zero_page; a.B = 1; a.H = M; a.W = 1; a.Cin = K; a.Cout = N; a.taps = 1;
    a.bs_in = bs_in; a.bs_w = bs_w; a.bs_out = bs_out;
    conv_flops += 2.0 * nb * M * (double)N * K;
    conv_flops_exec += 2.0 * nb * M * (double)N * K;
    ++conv_launches;
    hipLaunchKernelGGL((conv_gemm_kernel<T>), dim3(((M + 127) / 128) * (N / 128), nb), dim3(256), 65536, s, a);
    return ECAD_OK;
  }
  // statistics -> per-(image, channel) scale/shift in gn_scale / gn_shift
  void group_norm_stats(const T* x, const std::string& name, int B, int HW, int C, hipStream_t s) {
    if (stats_of == x) {   // the convolution that wrote x left the partial sums (conv4.h): only the finalize pass is left
      stats_of = nullptr;
      hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, gn_sums, stats_nblk, W_<float>(name + ".weight"), W_<float>(name + ".bias"), gn_scale, gn_shift, C, HW, 32, 1e-6f);
      return;
    }
    const int ppb = HW >= 16384 ? 512 : HW >= 4096 ? 256 : 64;
    const int nblk = (HW + ppb - 1) / ppb;
    if (nblk > gn_blocks_cap) { async_failed = true; return; }  // cannot happen for a geometry ensure_acts has sized the buffers for; decode() reports it
    hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(nblk, B), dim3(256), 0, s, x, gn_sums, HW, C, 32, ppb);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, gn_sums, nblk, W_<float>(name + ".weight"), W_<float>(name + ".bias"), gn_scale, gn_shift, C, HW, 32, 1e-6f);
  }
  // GroupNorm + SiLU into the zero-bordered layout (conv4.h)
  void group_norm_bordered(const T* x, T* y, const std::string& name, int B, int H, int W, int C, hipStream_t s) {
    group_norm_stats(x, name, B, H * W, C, s);
    const size_t chunks = (size_t)B * (H + 2) * (W + 2) * (C / 8);
    const unsigned grid = (unsigned)((chunks + 255) / 256 < 65536 ? (chunks + 255) / 256 : 65536);
    hipLaunchKernelGGL((gn_apply_pad_kernel<T, true>), dim3(grid), dim3(256), 0, s, x, y, gn_scale, gn_shift, chunks, H, W, C);
  }
  void group_norm(const T* x, T* y, const std::string& name, int B, int HW, int C, bool act, hipStream_t s) {
    group_norm_stats(x, name, B, HW, C, s);
    if (stats_of == y) stats_of = nullptr;
    const size_t chunks = (size_t)B * HW * (C / 8);
    const unsigned grid = (unsigned)((chunks + 255) / 256 < 65536 ? (chunks + 255) / 256 : 65536);
    if (act) hipLaunchKernelGGL((gn_apply_kernel<T, true>), dim3(grid), dim3(256), 0, s, x, y, gn_scale, gn_shift, chunks, HW, C);
    else hipLaunchKernelGGL((gn_apply_kernel<T, false>), dim3(grid), dim3(256), 0, s, x, y, gn_scale, gn_shift, chunks, HW, C);
  }
  // x (in `x`) -> resnet -> result in `out`; t1, t2 scratch
  int resnet(const std::string& p, const T* x, T* out, T* t1, T* t2, int B, int H, int W, int cin, int cout, hipStream_t s) {
    int rc;
    const bool b1 = conv4_bn(B * H * W, cin, cout, 9) != 0, b2 = conv4_bn(B * H * W, cout, cout, 9) != 0;
    if (b1) group_norm_bordered(x, bufP, p + "norm1", B, H, W, cin, s);
    else group_norm(x, t1, p + "norm1", B, H * W, cin, true, s);
    if ((rc = conv(b1 ? bufP : t1, p + "conv1", t2, nullptr, B, H, W, cin, cout, 9, 0, s, nullptr, nullptr, false, b1))) return rc;
    if (b2) group_norm_bordered(t2, bufP, p + "norm2", B, H, W, cout, s);
    else group_norm(t2, t2, p + "norm2", B, H * W, cout, true, s);
    const T* res = x;
    if (cin != cout) {
      if ((rc = conv(x, p + "conv_shortcut", t1, nullptr, B, H, W, cin, cout, 1, 0, s))) return rc;
      res = t1;
    }
    return conv(b2 ? bufP : t2, p + "conv2", out, res, B, H, W, cout, cout, 9, 0, s, nullptr, nullptr, false, b2);
  }

  int decode(const float* lat, int B, int h, int w, float scaling, float shift, uint8_t* out, hipStream_t s) override {
    VHIPCHK(hipSetDevice(dev));
    if (!finalized) return fail(ECAD_E_NOT_READY, "VAE weights not finalized");
    if (B < 1 || h < 4 || w < 4 || (h * w) % 128 || (w & 1)) return fail(ECAD_E_INVALID, "latent %dx%dx%d unsupported (h*w must be a multiple of 128, w even)", B, h, w);
    int rc = ensure_acts(B, h, w);
    if (rc) return rc;
    conv_flops = conv_flops_exec = 0; conv_launches = conv3s_launches = conv4_launches = 0;
    stats_of = nullptr;
    T *x = bufA, *y = bufB, *t = bufC;
    int H = h, Wd = w;
    {
      const float* pqw = use_pq ? W_<float>("post_quant_conv.weight") : nullptr;
      const float* pqb = use_pq ? W_<float>("post_quant_conv.bias") : nullptr;
      const dim3 grid((H * Wd + CIN_PIX - 1) / CIN_PIX, B);
      if (latent_ch == 4)
        hipLaunchKernelGGL((vae_conv_in_kernel<T, 4>), grid, dim3(256), 0, s, lat, pqw, pqb, W_<float>("decoder.conv_in.weight"), W_<float>("decoder.conv_in.bias"), x, B, H, Wd, 512, 1.0f / scaling, shift);
      else
        hipLaunchKernelGGL((vae_conv_in_kernel<T, 16>), grid, dim3(256), 0, s, lat, pqw, pqb, W_<float>("decoder.conv_in.weight"), W_<float>("decoder.conv_in.bias"), x, B, H, Wd, 512, 1.0f / scaling, shift);
    }
    // ---- mid block
    if ((rc = resnet("decoder.mid_block.resnets.0.", x, y, t, bufD, B, H, Wd, 512, 512, s))) return rc;
    {  // attention: one head of 512 channels over H*W tokens; result back into x
      const std::string a = "decoder.mid_block.attentions.0.";
      const int HW = H * Wd, C = 512;
      group_norm(y, t, a + "group_norm", B, HW, C, false, s);
      T* q = x;                          // [B*HW, C]
      T* k = bufC + (size_t)B * HW * C;  // second region of bufC (t occupies the first B*HW*C elements)
      T* v = k + (size_t)B * HW * C;
      T* vt = v + (size_t)B * HW * C;
      if ((rc = conv(t, a + "to_q", q, nullptr, B, HW, 1, C, C, 1, 0, s))) return rc;
      if ((rc = conv(t, a + "to_k", k, nullptr, B, HW, 1, C, C, 1, 0, s))) return rc;
      if ((rc = conv(t, a + "to_v", v, nullptr, B, HW, 1, C, C, 1, 0, s))) return rc;
      hipLaunchKernelGGL((transpose_kernel<T>), dim3((C + 31) / 32, (HW + 31) / 32, B), dim3(32, 8), 0, s, v, vt, HW, C);
      T* o = t;  // attention output [B*HW, C] (group-normed input no longer needed once q,k,v exist)
      // score matrices of as many images at once as bufD holds ([nb][HW][HW]; all 16 of the bench geometry): the three launches per image of rounds 1-5 were 0.6 ms
      // of launch-bound work per decode
      const size_t cap_imgs = bufD_elems / ((size_t)HW * HW);
      for (int b = 0; b < B;) {
        const int nb = (int)((size_t)(B - b) < cap_imgs ? (size_t)(B - b) : cap_imgs);
        T* S = bufD;
        if ((rc = gemm_batched(q + (size_t)b * HW * C, k + (size_t)b * HW * C, S, nb, HW, C, HW, (size_t)HW * C, (size_t)HW * C, (size_t)HW * HW, s))) return rc;
        hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3(HW * nb), dim3(256), 0, s, S, HW, 1.0f / sqrtf((float)C));
        if ((rc = gemm_batched(S, vt + (size_t)b * HW * C, o + (size_t)b * HW * C, nb, HW, HW, C, (size_t)HW * HW, (size_t)HW * C, (size_t)HW * C, s))) return rc;
        b += nb;
      }
      if ((rc = conv(o, a + "to_out.0", x, y, B, HW, 1, C, C, 1, 0, s))) return rc;  // + residual (input of the attention)
    }
    if ((rc = resnet("decoder.mid_block.resnets.1.", x, y, t, bufD, B, H, Wd, 512, 512, s))) return rc;
    // ---- up blocks (ping-pong x <-> y)
    const int chans[4] = {512, 512, 256, 128};
    int prev = 512;
    T* cur = y;
    T* oth = x;
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 3; ++j) {
        const std::string p = "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j) + ".";
        if ((rc = resnet(p, cur, oth, t, bufD, B, H, Wd, j == 0 ? prev : chans[i], chans[i], s))) return rc;
        std::swap(cur, oth);
      }
      if (i < 3) {
        H *= 2; Wd *= 2;
        const std::string un = "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
        const int Ml = B * (H / 2) * (Wd / 2), c = chans[i];
        if (!up_nine && !conv_tile128 && !conv_gather && Ml % 256 == 0 && c % 256 == 0 && (4 * c / 64) % G4Period<T, 256>::value == 0 && (Ml / 256) * (c / 256) * 4 >= num_cus &&
            (double)Ml * 1.13 * c * sizeof(T) < 4294967296.0) {   // 32-bit gathered offsets, as conv4_bn
          // four 2 x 2 convolutions over the bordered low-resolution image, one per output phase (conv4.h): 4/9 of the nine-tap form's multiplications
          const size_t chunks = (size_t)B * (H / 2 + 2) * (Wd / 2 + 2) * (c / 8);
          const unsigned grid = (unsigned)((chunks + 255) / 256 < 65536 ? (chunks + 255) / 256 : 65536);
          hipLaunchKernelGGL((upsample_pad_kernel<T>), dim3(grid), dim3(256), 0, s, cur, bufP, chunks, H / 2, Wd / 2, c, 0);
          ConvArgs a{};
          a.in = bufP; a.w = korder_tap ? phase_w.at(un + ".weight") : phase_chunk_w.at(un + ".weight"); a.korder = korder_tap ? 1 : 0; a.bias = W_<float>(un + ".bias"); a.out = oth; a.zero_page = zero_page;
          a.B = B; a.H = H; a.W = Wd; a.Cin = c; a.Cout = c; a.taps = 9; a.up = 2;
          a.tile2d = !tile_linear && (H / 2) % 16 == 0 && (Wd / 2) % 16 == 0;
          if (stats_of == oth) stats_of = nullptr;
          const int HWl = (H / 2) * (Wd / 2);
          if (!gn_separate && HWl % 256 == 0 && 4 * (HWl / 256) <= gn_blocks_cap) { a.gn_part = gn_sums; stats_of = oth; stats_nblk = 4 * (HWl / 256); }
          conv_flops += 2.0 * B * H * Wd * (double)c * 9 * c;
          conv_flops_exec += 2.0 * B * H * Wd * (double)c * 4 * c;
          ++conv_launches; ++conv4_launches;
          hipLaunchKernelGGL((conv4_kernel<T, 256, false>), dim3((Ml / 256) * (c / 256) * 4), dim3(256), conv4_lds_bytes<256>(), s, a);
        } else if (conv4_bn(B * H * Wd, chans[i], chans[i], 9)) {  // the upsampled image is written out (bordered) and the convolution runs on conv4_kernel
          const size_t chunks = (size_t)B * (H + 2) * (Wd + 2) * (chans[i] / 8);
          const unsigned grid = (unsigned)((chunks + 255) / 256 < 65536 ? (chunks + 255) / 256 : 65536);
          hipLaunchKernelGGL((upsample_pad_kernel<T>), dim3(grid), dim3(256), 0, s, cur, bufP, chunks, H, Wd, chans[i], 1);
          if ((rc = conv(bufP, un, oth, nullptr, B, H, Wd, chans[i], chans[i], 9, 0, s, nullptr, nullptr, false, true))) return rc;
        } else if ((rc = conv(cur, un, oth, nullptr, B, H, Wd, chans[i], chans[i], 9, 1, s))) return rc;
        std::swap(cur, oth);
      }
      prev = chans[i];
    }
    // conv_norm_out's apply + SiLU is fused into the conv_out kernel's halo staging
    group_norm_stats(cur, "decoder.conv_norm_out", B, H * Wd, 128, s);
    hipLaunchKernelGGL((vae_conv_out_kernel<T>), dim3((H / CO_TH) * (Wd / CO_TW), B), dim3(256), 0, s, cur, gn_scale, gn_shift, W_<T>("decoder.conv_out.weight"),
                       W_<float>("decoder.conv_out.bias"), out, B, H, Wd);
    VHIPCHK(hipGetLastError());
    if (async_failed) { async_failed = false; return fail(ECAD_E_INVALID, "internal: a GroupNorm stage needs more partial-sum blocks than ensure_acts sized for"); }
    return ECAD_OK;
  }
};

}  // namespace ecad

using namespace ecad;
struct EcadVae { VaeBase* impl; };

extern "C" {
int ecad_vae_create_ex(int device, int dtype, int latent_channels, int use_post_quant_conv, EcadVae** out);
int ecad_vae_create(int device, int dtype, EcadVae** out) { return ecad_vae_create_ex(device, dtype, 4, 1, out); }
int ecad_vae_create_ex(int device, int dtype, int latent_channels, int use_post_quant_conv, EcadVae** out) {
  if (!out) return ECAD_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_vae_create_error = "no HIP device visible"; return ECAD_E_HIP; }
  VaeBase* impl = nullptr;
  if (dtype == ECAD_DTYPE_F16) impl = new VaeT<f16_t>();
  else if (dtype == ECAD_DTYPE_BF16) impl = new VaeT<bf16_t>();
  else { g_vae_create_error = "dtype must be f16 or bf16"; return ECAD_E_INVALID; }
  impl->dev = device;
  impl->latent_ch = latent_channels;
  impl->use_pq = use_post_quant_conv != 0;
  int rc = impl->init();
  if (rc) { g_vae_create_error = impl->err; delete impl; return rc; }
  *out = new EcadVae{impl};
  return ECAD_OK;
}
void ecad_vae_destroy(EcadVae* v) { if (v) { delete v->impl; delete v; } }
const char* ecad_vae_last_error(const EcadVae* v) { return v ? v->impl->err.c_str() : g_vae_create_error.c_str(); }
int ecad_vae_load_weights(EcadVae* v, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) {
  if (!v || !name || !ptr || !shape) return ECAD_E_INVALID;
  return v->impl->load(name, ptr, dtype, shape, ndim);
}
int ecad_vae_finalize_weights(EcadVae* v) { return v ? v->impl->finalize() : ECAD_E_INVALID; }
int ecad_vae_decode(EcadVae* v, const float* latents_dev, int B, int h, int w, float scaling_factor, uint8_t* out_u8_dev, void* stream) {
  if (!v || !latents_dev || !out_u8_dev) return ECAD_E_INVALID;
  return v->impl->decode(latents_dev, B, h, w, scaling_factor, 0.f, out_u8_dev, (hipStream_t)stream);
}
int ecad_vae_decode_stats(const EcadVae* v, double* conv_flops, int* conv_launches, int* conv3s_launches) {
  if (!v) return ECAD_E_INVALID;
  if (conv_flops) *conv_flops = v->impl->conv_flops;
  if (conv_launches) *conv_launches = v->impl->conv_launches;
  if (conv3s_launches) *conv3s_launches = v->impl->conv3s_launches;
  return ECAD_OK;
}
int ecad_vae_decode_stats_ex(const EcadVae* v, double* conv_flops, int* conv_launches, int* conv3s_launches, int* conv4_launches, double* conv_flops_executed) {
  if (!v) return ECAD_E_INVALID;
  if (conv4_launches) *conv4_launches = v->impl->conv4_launches;
  if (conv_flops_executed) *conv_flops_executed = v->impl->conv_flops_exec;
  return ecad_vae_decode_stats(v, conv_flops, conv_launches, conv3s_launches);
}
int ecad_vae_decode_ex(EcadVae* v, const float* latents_dev, int B, int h, int w, float scaling_factor, float shift_factor, uint8_t* out_u8_dev, void* stream) {
  if (!v || !latents_dev || !out_u8_dev) return ECAD_E_INVALID;
  return v->impl->decode(latents_dev, B, h, w, scaling_factor, shift_factor, out_u8_dev, (hipStream_t)stream);
}
}
