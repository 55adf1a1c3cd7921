// probe_fp8_handover.hip -- the fp8 FF1 -> FF2 hand-over of gemm3s8.h in isolation: gemm2w8_kernel<.., EMIT> (e4m3 rows + one E8M0 byte per (row, 288-column tile))
// feeding gemm3s8_kernel<.., BS> (the bytes as MFMA block scales), against the 16-bit output + row quantiser + plain gemm3s8 path, with fp64 host references:
//   (1) producer: every emitted element decodes to the 16-bit value within e4m3's half-step, the scale is the smallest power of two that fits the tile row;
//   (2) consumer: its output equals the fp64 product of the DEQUANTISED operands (the hardware applies the block scales exactly);
//   (3) accuracy: both fp8 paths against the fp64 product of the 16-bit activations and the dequantised weights.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only scripts/probe_fp8_handover.hip -o build_tmp/probe_fp8_handover
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "../ecad_amd/csrc/gemm3s8.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned rng = 3u;
static float rnd() { rng = rng * 1664525u + 1013904223u; return ((int)(rng >> 8) % 2001 - 1000) * 0.001f; }
static float gauss() { float s = 0; for (int i = 0; i < 6; ++i) s += rnd(); return s * 0.7071f; }
static double e4m3_decode(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  double v = e == 0 ? std::ldexp((double)m, -9) : std::ldexp(1.0 + m / 8.0, e - 7);
  return s ? -v : v;
}
static double f16_decode(unsigned short u) { _Float16 h; std::memcpy(&h, &u, 2); return (double)h; }

int main() {
  const int M = 512, D = 1152, F = 4608;  // FF1: [M, D] x [F, D]^T -> [M, F]; FF2: [M, F] x [D, F]^T -> [M, D]
  // operands of FF1 as e4m3 bytes with unit row scales (what matters here is what FF1 PRODUCES): moderate magnitudes so that gelu(.) spans a few units
  std::vector<uint8_t> a8((size_t)M * D), w1((size_t)F * D), w2((size_t)D * F);
  auto enc = [](float v) {  // nearest e4m3 of v (host, crude: search)
    uint8_t best = 0; double bd = 1e30;
    for (int b = 0; b < 256; ++b) { if ((b & 0x7f) == 0x7f) continue; const double d = std::fabs(e4m3_decode((uint8_t)b) - v); if (d < bd) { bd = d; best = (uint8_t)b; } }
    return best;
  };
  uint8_t lut[2048];
  for (int i = 0; i < 2048; ++i) lut[i] = enc((i - 1024) / 128.0f);
  auto q = [&](float v) { int i = (int)std::lround(v * 128.f) + 1024; i = i < 0 ? 0 : (i > 2047 ? 2047 : i); return lut[i]; };
  for (auto& v : a8) v = q(gauss() * 2.0f);
  for (auto& v : w1) v = q(gauss() * 1.0f);
  for (auto& v : w2) v = q(gauss() * 1.0f);
  std::vector<float> sa(M, 1.0f), sw1(F, 0.02f), sw2(D, 0.02f), b1(F), b2(D, 0.f);
  for (auto& v : b1) v = rnd() * 0.5f;
  uint8_t *dA, *dW1, *dW2, *dQ8, *dE8, *dA8b; float *dSa, *dSw1, *dSw2, *dB1, *dB2, *dSq; unsigned short *dC, *dO[2], *dR[2];
  CK(hipMalloc(&dA, a8.size())); CK(hipMalloc(&dW1, w1.size())); CK(hipMalloc(&dW2, w2.size())); CK(hipMalloc(&dQ8, (size_t)M * F)); CK(hipMalloc(&dE8, (size_t)M * 16)); CK(hipMalloc(&dA8b, (size_t)M * F));
  CK(hipMalloc(&dSa, M * 4)); CK(hipMalloc(&dSw1, F * 4)); CK(hipMalloc(&dSw2, D * 4)); CK(hipMalloc(&dB1, F * 4)); CK(hipMalloc(&dB2, D * 4)); CK(hipMalloc(&dSq, M * 4));
  CK(hipMalloc(&dC, (size_t)M * F * 2));
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&dO[i], (size_t)M * D * 2)); CK(hipMalloc(&dR[i], (size_t)M * D * 2)); CK(hipMemset(dR[i], 0, (size_t)M * D * 2)); }
  CK(hipMemcpy(dA, a8.data(), a8.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dW1, w1.data(), w1.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dW2, w2.data(), w2.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dSa, sa.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dSw1, sw1.data(), F * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dSw2, sw2.data(), D * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB1, b1.data(), F * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB2, b2.data(), D * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)gemm2w8_kernel<f16_t, EPI_BIAS_GELU, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)gemm2w8_kernel<f16_t, EPI_BIAS_GELU, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)gemm3s8_kernel<f16_t, EPI_BIAS_CACHE_RESID, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)gemm3s8_kernel<f16_t, EPI_BIAS_CACHE_RESID, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES + G3S8_TAB_BYTES));
  // ---- FF1 both ways
  Gemm8Args<f16_t> f1{};
  f1.A8 = dA; f1.W8 = dW1; f1.sa = dSa; f1.sw = dSw1; f1.lda8 = D;
  f1.e.M = M; f1.e.N = F; f1.e.K = D; f1.e.bias = dB1; f1.e.C = (f16_t*)dC; f1.e.ldc = F; f1.e.rows_per_batch = 256;
  hipLaunchKernelGGL((gemm2w8_kernel<f16_t, EPI_BIAS_GELU, false>), dim3((M / 256) * (F / 288)), dim3(512), G2_LDS_BYTES, 0, f1);
  f1.q8_out = dQ8; f1.e8_out = dE8; f1.ldq8 = F;
  hipLaunchKernelGGL((gemm2w8_kernel<f16_t, EPI_BIAS_GELU, true>), dim3((M / 256) * (F / 288)), dim3(512), G2_LDS_BYTES, 0, f1);
  CK(hipDeviceSynchronize());
  std::vector<unsigned short> c((size_t)M * F);
  std::vector<uint8_t> q8((size_t)M * F), e8((size_t)M * 16);
  CK(hipMemcpy(c.data(), dC, c.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(q8.data(), dQ8, q8.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(e8.data(), dE8, e8.size(), hipMemcpyDeviceToHost));
  {
    size_t bad_scale = 0, bad_val = 0;
    double worst = 0;
    for (int m = 0; m < M; ++m)
      for (int t = 0; t < 16; ++t) {
        double amax = 0;
        for (int n = 0; n < 288; ++n) amax = std::max(amax, std::fabs(f16_decode(c[(size_t)m * F + t * 288 + n])));
        const int eb = e8[(size_t)m * 16 + t];
        const double sc = std::ldexp(1.0, eb - 127);
        if (amax > 0 && !(sc * 448.0 >= amax && sc * 224.0 < amax * (1 + 1e-6))) ++bad_scale;
        for (int n = 0; n < 288; ++n) {
          const double x = f16_decode(c[(size_t)m * F + t * 288 + n]), y = e4m3_decode(q8[(size_t)m * F + t * 288 + n]) * sc;
          const double tol = std::max(std::fabs(x) / 16.0, sc * std::ldexp(1.0, -10)) * 1.0001;  // half a step of a 3-bit mantissa, or half a subnormal step
          if (std::fabs(x - y) > tol) { if (bad_val < 5) printf("  row %d col %d: 16-bit %g, emitted %g (scale 2^%d)\n", m, t * 288 + n, x, y, eb - 127); ++bad_val; }
          if (x != 0) worst = std::max(worst, std::fabs(x - y) / std::max(std::fabs(x), sc * 0.0156));
        }
      }
    printf("producer: %zu of %d (row, tile) scales are not the smallest power of two that fits, %zu of %zu elements off by more than half an e4m3 step (worst relative %.3f)\n", bad_scale, M * 16, bad_val,
           (size_t)M * F, worst);
  }
  // ---- FF2 both ways: (0) block scales from the producer, (1) row quantiser on the 16-bit output
  launch_quant_rows_fp8<f16_t>((const f16_t*)dC, (int64_t)F, F, (const f16_t*)nullptr, (int64_t)0, 0, dA8b, (int64_t)F, dSq, M, (hipStream_t)0);
  Gemm8Args<f16_t> f2{};
  f2.W8 = dW2; f2.sw = dSw2; f2.lda8 = F;
  f2.e.M = M; f2.e.N = D; f2.e.K = F; f2.e.bias = dB2; f2.e.ldc = D; f2.e.ldr = D; f2.e.rows_per_batch = 256;
  f2.A8 = dQ8; f2.sa = dSq; f2.e8_in = dE8; f2.e8_ld = 16; f2.e.C = (f16_t*)dO[0]; f2.e.resid = (f16_t*)dR[0];
  hipLaunchKernelGGL((gemm3s8_kernel<f16_t, EPI_BIAS_CACHE_RESID, true>), dim3((M / 256) * (D / 144)), dim3(512), G3_LDS_BYTES + G3S8_TAB_BYTES, 0, f2);
  f2.A8 = dA8b; f2.e8_in = nullptr; f2.e.C = (f16_t*)dO[1]; f2.e.resid = (f16_t*)dR[1];
  hipLaunchKernelGGL((gemm3s8_kernel<f16_t, EPI_BIAS_CACHE_RESID, false>), dim3((M / 256) * (D / 144)), dim3(512), G3_LDS_BYTES, 0, f2);
  CK(hipDeviceSynchronize());
  std::vector<unsigned short> o0((size_t)M * D), o1((size_t)M * D);
  std::vector<uint8_t> a8b((size_t)M * F);
  std::vector<float> sq(M);
  CK(hipMemcpy(o0.data(), dO[0], o0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), dO[1], o1.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(a8b.data(), dA8b, a8b.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(sq.data(), dSq, M * 4, hipMemcpyDeviceToHost));
  double n_exact = 0, e_self[2] = {0, 0}, e_true[2] = {0, 0}, n_true = 0;
  std::vector<double> wd((size_t)D * F);
  for (size_t i = 0; i < wd.size(); ++i) wd[i] = e4m3_decode(w2[i]) * 0.02;
  for (int m = 0; m < 64; ++m) {  // a sample of rows (fp64 on the host)
    std::vector<double> x16(F), xs(F), xr(F);
    for (int k = 0; k < F; ++k) {
      x16[k] = f16_decode(c[(size_t)m * F + k]);
      xs[k] = e4m3_decode(q8[(size_t)m * F + k]) * std::ldexp(1.0, e8[(size_t)m * 16 + k / 288] - 127);
      xr[k] = e4m3_decode(a8b[(size_t)m * F + k]) * sq[m];
    }
    for (int n = 0; n < D; ++n) {
      double t = 0, s0 = 0, s1 = 0;
      const double* wrow = &wd[(size_t)n * F];
      for (int k = 0; k < F; ++k) { t += x16[k] * wrow[k]; s0 += xs[k] * wrow[k]; s1 += xr[k] * wrow[k]; }
      const double g0 = f16_decode(o0[(size_t)m * D + n]), g1 = f16_decode(o1[(size_t)m * D + n]);
      e_self[0] += (g0 - s0) * (g0 - s0); e_self[1] += (g1 - s1) * (g1 - s1);
      e_true[0] += (g0 - t) * (g0 - t); e_true[1] += (g1 - t) * (g1 - t);
      n_true += t * t; n_exact += s0 * s0;
    }
  }
  printf("consumer: kernel output against the fp64 product of ITS OWN dequantised operands: block-scaled %.3e, row-scaled %.3e (rel-L2; both should be one 16-bit rounding, ~3e-4)\n",
         std::sqrt(e_self[0] / n_exact), std::sqrt(e_self[1] / n_exact));
  printf("accuracy: against the fp64 product of the 16-bit activations: block-scaled hand-over %.3e, row quantiser %.3e\n", std::sqrt(e_true[0] / n_true), std::sqrt(e_true[1] / n_true));
  {  // time of the two consumers at the batch-16 shape (M = 8192; operands of the first 512 rows repeated: timing only)
    const int MB = 8192;
    uint8_t *bA, *bE; float* bS; unsigned short *bO, *bR;
    CK(hipMalloc(&bA, (size_t)MB * F)); CK(hipMalloc(&bE, (size_t)MB * 16)); CK(hipMalloc(&bS, MB * 4)); CK(hipMalloc(&bO, (size_t)MB * D * 2)); CK(hipMalloc(&bR, (size_t)MB * D * 2));
    for (int r = 0; r < MB / M; ++r) {
      CK(hipMemcpy(bA + (size_t)r * M * F, dQ8, (size_t)M * F, hipMemcpyDeviceToDevice)); CK(hipMemcpy(bE + (size_t)r * M * 16, dE8, (size_t)M * 16, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(bS + (size_t)r * M, dSq, M * 4, hipMemcpyDeviceToDevice));
    }
    CK(hipMemset(bR, 0, (size_t)MB * D * 2));
    Gemm8Args<f16_t> g2 = f2;
    g2.e.M = MB; g2.A8 = bA; g2.sa = bS; g2.e.C = (f16_t*)bO; g2.e.resid = (f16_t*)bR;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
      g2.e8_in = which == 0 ? bE : nullptr; g2.e8_ld = 16;
      float best = 1e9f;
      for (int round = 0; round < 4; ++round) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) {
          if (which == 0) hipLaunchKernelGGL((gemm3s8_kernel<f16_t, EPI_BIAS_CACHE_RESID, true>), dim3((MB / 256) * (D / 144)), dim3(512), G3_LDS_BYTES + G3S8_TAB_BYTES, 0, g2);
          else hipLaunchKernelGGL((gemm3s8_kernel<f16_t, EPI_BIAS_CACHE_RESID, false>), dim3((MB / 256) * (D / 144)), dim3(512), G3_LDS_BYTES, 0, g2);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (round) best = std::min(best, ms / 50);
      }
      printf("FF2 at M = 8192 on gemm3s8, %s: %.1f us per launch (%.0f TF)\n", which == 0 ? "block scales from the table" : "row scales", best * 1e3, 2.0 * MB * D * F / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
