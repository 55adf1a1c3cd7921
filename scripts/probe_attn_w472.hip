// probe_attn_w472.hip -- stand-alone check + timing of attention_w472.h (d_h 72 on one wave per SIMD: 20 + 24 MFMAs per 64-key tile, row sums out of the matrix pipe)
// against a CPU fp64 reference and against attention_dma.h's four-fragment kernel (what the engine ran at N = 4096 until round 6), without the Python stack.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only scripts/probe_attn_w472.hip -o build_tmp/probe_attn_w472
// run on the GPU box: build_tmp/probe_attn_w472 [reps]      (PROBE_TIME_ONLY=1: no checks)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../ecad_amd/csrc/attention_dma.h"
#include "../ecad_amd/csrc/attention_w472.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> static unsigned short to_bits(float f) {
  if (std::is_same<T, f16_t>::value) { _Float16 hf = (_Float16)f; unsigned short v; std::memcpy(&v, &hf, 2); return v; }
  unsigned u; std::memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
template <typename T> static float from_bits(unsigned short v) {
  if (std::is_same<T, f16_t>::value) { _Float16 hf; std::memcpy(&hf, &v, 2); return (float)hf; }
  unsigned u = (unsigned)v << 16; float f; std::memcpy(&f, &u, 4); return f;
}

struct Shape { int B, H, N, seg_t; };

template <typename T> static std::vector<unsigned short> make_qkv(const Shape& s, float amp, int spike, unsigned seed) {
  const size_t D = (size_t)s.H * 72, rows = (size_t)s.B * s.N;
  std::vector<unsigned short> h(rows * 3 * D);
  unsigned x = seed;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    v = to_bits<T>(((int)(x >> 8) % 2001 - 1000) * 0.001f * amp);
  }
  if (spike) {  // a few keys far along the sequence line up with query rows: the running maximum jumps late, past any threshold
    for (int b = 0; b < s.B; ++b)
      for (int hd = 0; hd < s.H; ++hd)
        for (int j = 0; j < 3; ++j) {
          const int key = (s.N * (3 + j)) / 7 + 5 * j, qr = 17 + 70 * j + hd;
          for (int d = 0; d < 72; ++d) {
            const float qv = from_bits<T>(h[((size_t)b * s.N + qr) * 3 * D + hd * 72 + d]);
            h[((size_t)b * s.N + key) * 3 * D + D + hd * 72 + d] = to_bits<T>(qv * 6.0f);
          }
        }
  }
  return h;
}

// position r of sample b -> row of the [text | image] layout (attention.h att_row)
static size_t row_of(const Shape& s, int b, int r) {
  if (s.seg_t == 0) return (size_t)b * s.N + r;
  return r < s.seg_t ? (size_t)b * s.seg_t + r : (size_t)s.B * s.seg_t + (size_t)b * (s.N - s.seg_t) + (r - s.seg_t);
}

template <typename T> static void cpu_ref(const Shape& s, const std::vector<unsigned short>& h, std::vector<float>& out, int q_step) {
  const size_t D = (size_t)s.H * 72;
  out.assign((size_t)s.B * s.N * D, 0.f);
  std::vector<double> sc(s.N);
  const double scale = 1.0 / std::sqrt(72.0);
  for (int b = 0; b < s.B; ++b)
    for (int hd = 0; hd < s.H; ++hd)
      for (int q = 0; q < s.N; q += q_step) {
        const unsigned short* qp = &h[row_of(s, b, q) * 3 * D + hd * 72];
        double mx = -1e300;
        for (int k = 0; k < s.N; ++k) {
          const unsigned short* kp = &h[row_of(s, b, k) * 3 * D + D + hd * 72];
          double a = 0;
          for (int d = 0; d < 72; ++d) a += (double)from_bits<T>(qp[d]) * from_bits<T>(kp[d]);
          sc[k] = a * scale;
          mx = std::max(mx, sc[k]);
        }
        double den = 0;
        for (int k = 0; k < s.N; ++k) { sc[k] = std::exp(sc[k] - mx); den += sc[k]; }
        float* op = &out[row_of(s, b, q) * D + hd * 72];
        for (int d = 0; d < 72; ++d) {
          double a = 0;
          for (int k = 0; k < s.N; ++k) a += sc[k] * from_bits<T>(h[row_of(s, b, k) * 3 * D + 2 * D + hd * 72 + d]);
          op[d] = (float)(a / den);
        }
      }
}

template <typename T> static AttnArgs make_args(const Shape& s, unsigned short* qkv, unsigned short* out) {
  const size_t D = (size_t)s.H * 72;
  AttnArgs a{};
  a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.out = out; a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
  a.B = s.B; a.H = s.H; a.Nq = s.N; a.Nk = s.N; a.dh = 72; a.scale_log2e = 1.4426950408889634f / sqrtf(72.f);
  a.seg_t = s.seg_t; a.seg_bt = s.B * s.seg_t;
  return a;
}

template <typename T> static void launch_w472(const AttnArgs& a_in, int cus) {
  AttnArgs a = a_in;
  const void* k = (const void*)attn_w472_kernel<T>;
  static bool attr = false;
  if (!attr) { CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES)); attr = true; }
  const int items = a.B * a.H * (a.Nq / 256);
  void* p1[] = {&a};
  CK(hipLaunchKernel(k, dim3(items < cus ? items : cus), dim3(256), p1, W4_LDS_BYTES, 0));
}
template <typename T> static void launch_dma(const AttnArgs& a_in, int cus) {   // engine.hip launch_attn(wide): four query fragments per wave, 512 query rows per work item
  AttnArgs a = a_in;
  const int lds = attn_dma_lds_bytes<72>();
  const void* k = (const void*)attn_dma_kernel<T, 72, 4>;
  static bool attr = false;
  if (!attr) { CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
  const int items = a.B * a.H * (a.Nq / 512);
  void* p1[] = {&a};
  CK(hipLaunchKernel(k, dim3(items < cus ? items : cus), dim3(512), p1, lds, 0));
}

template <typename T> static bool check(const char* name, const Shape& s, float amp, int spike, int q_step, int cus) {
  const size_t D = (size_t)s.H * 72, rows = (size_t)s.B * s.N;
  auto h = make_qkv<T>(s, amp, spike, 777u + spike);
  unsigned short *qkv, *out;
  CK(hipMalloc(&qkv, h.size() * 2)); CK(hipMalloc(&out, rows * D * 2));
  CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> ref;
  cpu_ref<T>(s, h, ref, q_step);
  bool ok = true;
  for (int which = 0; which < 2; ++which) {
    if (which == 1 && s.N % 512) continue;
    CK(hipMemset(out, 0xFF, rows * D * 2));
    AttnArgs a = make_args<T>(s, qkv, out);
    if (which == 0) launch_w472<T>(a, cus); else launch_dma<T>(a, cus);
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> o(rows * D);
    CK(hipMemcpy(o.data(), out, o.size() * 2, hipMemcpyDeviceToHost));
    double num = 0, den = 0, worst = 0; size_t bad = 0;
    for (int b = 0; b < s.B; ++b)
      for (int q = 0; q < s.N; q += q_step)
        for (size_t c = 0; c < D; ++c) {
          const size_t i = row_of(s, b, q) * D + c;
          const double g = from_bits<T>(o[i]), e = ref[i];
          if (!(std::fabs(g) < 1e30)) ++bad;
          num += (g - e) * (g - e); den += e * e; worst = std::max(worst, std::fabs(g - e));
        }
    const double rel = std::sqrt(num / den);
    const double tol = std::is_same<T, f16_t>::value ? 1e-3 : 8e-3;
    printf("check %-28s %-4s rel-L2 %.3e  max|d| %.3e  nonfinite %zu  %s\n", name, which == 0 ? "w472" : "dma", rel, worst, bad, (rel < tol && bad == 0) ? "ok" : "FAIL");
    if (which == 0 && !(rel < tol && bad == 0)) ok = false;
  }
  CK(hipFree(qkv)); CK(hipFree(out));
  return ok;
}

template <typename T> static void timeit(const char* name, const Shape& s, int reps, int cus) {
  const size_t D = (size_t)s.H * 72, rows = (size_t)s.B * s.N;
  auto h = make_qkv<T>(s, 1.0f, 0, 12345u);
  unsigned short *qkv, *out;
  CK(hipMalloc(&qkv, h.size() * 2)); CK(hipMalloc(&out, rows * D * 2));
  CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  AttnArgs a = make_args<T>(s, qkv, out);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flops = 4.0 * s.B * s.H * (double)s.N * s.N * 72;
  float best[2] = {1e9f, 1e9f};
  for (int round = 0; round < 4; ++round)
    for (int which = 0; which < 2; ++which) {
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) { if (which == 0) launch_w472<T>(a, cus); else launch_dma<T>(a, cus); }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (round) best[which] = std::min(best[which], ms / reps);
    }
  printf("time  %-28s w472 %8.1f us %6.0f TF   dma %8.1f us %6.0f TF\n", name, best[0] * 1e3, flops / (best[0] * 1e-3) / 1e12, best[1] * 1e3, flops / (best[1] * 1e-3) / 1e12);
  CK(hipFree(qkv)); CK(hipFree(out));
}

// the asm kernel must give the same bits on every launch (a race between the DMA ring and the fragment reads shows up here long before it breaks a tolerance)
template <typename T> static bool repro(const char* name, const Shape& s, int launches, int cus) {
  const size_t D = (size_t)s.H * 72, rows = (size_t)s.B * s.N;
  auto h = make_qkv<T>(s, 1.0f, 0, 4242u);
  unsigned short *qkv, *out;
  CK(hipMalloc(&qkv, h.size() * 2)); CK(hipMalloc(&out, rows * D * 2));
  CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  AttnArgs a = make_args<T>(s, qkv, out);
  std::vector<unsigned short> first(rows * D), cur(rows * D);
  size_t diffs = 0;
  for (int i = 0; i < launches; ++i) {
    CK(hipMemset(out, 0xFF, rows * D * 2));
    launch_w472<T>(a, cus);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(i == 0 ? first.data() : cur.data(), out, first.size() * 2, hipMemcpyDeviceToHost));
    if (i > 0) for (size_t k = 0; k < first.size(); ++k) diffs += first[k] != cur[k];
  }
  printf("repro %-28s %d launches, elements that differ from the first launch: %zu  %s\n", name, launches, diffs, diffs == 0 ? "ok" : "FAIL");
  CK(hipFree(qkv)); CK(hipFree(out));
  return diffs == 0;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);
  bool ok = true;
  if (!getenv("PROBE_TIME_ONLY")) {
  ok &= check<f16_t>("f16 B1 H2 N256", {1, 2, 256, 0}, 1.0f, 0, 1, cus);
  ok &= check<f16_t>("f16 B1 H2 N512", {1, 2, 512, 0}, 1.0f, 0, 1, cus);
  ok &= check<bf16_t>("bf16 B1 H2 N512", {1, 2, 512, 0}, 1.0f, 0, 1, cus);
  ok &= check<f16_t>("f16 B1 H16 N1024 spikes", {1, 16, 1024, 0}, 1.0f, 1, 3, cus);
  ok &= check<bf16_t>("bf16 B1 H2 N1024 spikes", {1, 2, 1024, 0}, 1.0f, 1, 1, cus);
  ok &= check<f16_t>("f16 B1 H2 N1024 amp 3", {1, 2, 1024, 0}, 3.0f, 0, 1, cus);
  ok &= check<f16_t>("f16 B2 H16 N4096 sampled", {2, 16, 4096, 0}, 1.0f, 0, 41, cus);
  ok &= repro<f16_t>("B8 H16 N4096 f16", {8, 16, 4096, 0}, 4, cus);
  printf(ok ? "ALL CHECKS OK\n" : "CHECK FAILURES\n");
  }
  timeit<f16_t>("Sigma-1024 B8 H16 N4096 f16", {8, 16, 4096, 0}, reps, cus);
  timeit<bf16_t>("Sigma-1024 B8 H16 N4096 bf16", {8, 16, 4096, 0}, reps, cus);
  timeit<f16_t>("alpha-512 B32 H16 N1024 f16", {32, 16, 1024, 0}, reps, cus);
  return ok ? 0 : 1;
}
