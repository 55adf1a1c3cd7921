// probe_gemm3s.hip -- the product GEMM kernels of gemm3s.h / gemm2w.h on the model's shapes without the Python stack: microseconds per launch (HIP events over
// back-to-back launches on random data) and the in-kernel stamps of a -DG_STAMP build (cycles of pipeline fill / K loop / epilogue per workgroup, in-kernel clock).
// build (one binary per set of experiment macros):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only -DECAD_DIAG -DG_STAMP scripts/probe_gemm3s.hip -o build_tmp/probe_gemm3s[_x]
// run on the GPU box: build_tmp/probe_gemm3s [reps]
// (Round 4: streams created with hipExtStreamCreateWithCUMask -- each owning a share of every XCD's CUs -- HUNG the first launch on this pool; not pursued.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../ecad_amd/csrc/gemm3s.h"
#include "../ecad_amd/csrc/gemm2w.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#ifndef PROBE_VARIANT
#define PROBE_VARIANT 1
#endif

template <typename F> static void run_case(const char* name, F launch_on, int M, int N, int K, int epi, int grid_wgs, int reps) {
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<_Float16> ha((size_t)M * K), hw((size_t)N * K), hr((size_t)M * N);
  for (auto& v : ha) v = (_Float16)nd(rng);
  for (auto& v : hw) v = (_Float16)(0.03f * nd(rng));
  for (auto& v : hr) v = (_Float16)nd(rng);
  std::vector<float> hb(N);
  for (auto& v : hb) v = 0.1f * nd(rng);
  _Float16 *A, *W, *C, *R;
  float* bias;
  unsigned long long* dbg;
  CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&W, hw.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&R, (size_t)M * N * 2));
  CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&dbg, 8 * 8192 * 8));
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(R, hr.data(), hr.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dbg, 0, 8 * 8192 * 8));
  GemmArgs<f16_t> g{};
  g.A = A; g.W = W; g.bias = bias; g.C = C; g.resid = epi == 3 ? R : nullptr; g.gate_table = nullptr; g.gate_mod = (const f16_t*)dbg; g.rows_per_batch = 256;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N; g.variant = PROBE_VARIANT; g.epi_direct = 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](const GemmArgs<f16_t>& a) { launch_on(a, (hipStream_t)0); };
  for (int i = 0; i < 50; ++i) launch(g);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch(g);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(8 * 8192);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> fill, loop, epi_c, wall, clk;
  for (int b = 0; b < 8192; ++b) {
    const unsigned long long* d = &h[8 * b];
    if (!d[3]) continue;
    fill.push_back((double)d[0]); loop.push_back((double)d[1]); epi_c.push_back((double)d[2]); wall.push_back(d[3] / 100.0); clk.push_back((double)d[5] / (d[3] / 100.0) / 1e3);
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  // the same launches dealt to NS streams in turn (independent: same inputs, outputs racing on the same buffer -- timing only): what part of the gap between
  // "microseconds per launch" and the stamped tile time (launch latency + the drain of the epilogue's stores) another stream's kernel can fill
  double us_ms[3] = {0, 0, 0};
  {
    hipStream_t st[4];
    for (auto& q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    int k = 0;
    for (int ns : {2, 3, 4}) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < ns; ++q) CK(hipStreamWaitEvent(st[q], e0, 0));
      for (int i = 0; i < reps; ++i) launch_on(g, st[i % ns]);
      hipEvent_t ev[4];
      for (int q = 0; q < ns; ++q) { CK(hipEventCreate(&ev[q])); CK(hipEventRecord(ev[q], st[q])); CK(hipStreamWaitEvent(0, ev[q], 0)); }
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float m2 = 0;
      CK(hipEventElapsedTime(&m2, e0, e1));
      us_ms[k++] = m2 * 1e3 / reps;
      for (int q = 0; q < ns; ++q) CK(hipEventDestroy(ev[q]));
    }
    for (auto& q : st) CK(hipStreamDestroy(q));
  }
  const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / us / 1e6;
  const int steps = K / 64;
  printf("%-28s M=%d N=%d K=%d epi=%d: %7.2f us/launch = %6.0f TF (on 2 / 3 / 4 streams: %.2f / %.2f / %.2f us) | tiles %zu: fill %.0f  loop %.0f (%.0f/step)  epilogue %.0f cycles, tile wall %.2f us, clock %.3f GHz\n", name, M, N, K, epi, us, tf, us_ms[0], us_ms[1], us_ms[2],
         fill.size(), med(fill), med(loop), med(loop) / steps, med(epi_c), med(wall), med(clk));
  fflush(stdout);
  CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(R)); CK(hipFree(bias)); CK(hipFree(dbg));
  (void)grid_wgs;
}

template <int EPI, int TAG> static void g3(const GemmArgs<f16_t>& g, hipStream_t st) {
  static bool attr = false;
  auto k = gemm3s_kernel<f16_t, EPI, TAG>;
  if (!attr) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES)); attr = true; }
  hipLaunchKernelGGL(k, dim3((g.M / G3_BM) * (g.N / G3_BN)), dim3(G3_NW * 64), G3_LDS_BYTES, st, g);
}
template <int EPI, int TAG> static void g2(const GemmArgs<f16_t>& g, hipStream_t st) {
  static bool attr = false;
  auto k = gemm2w_kernel<f16_t, EPI, TAG>;
  if (!attr) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES)); attr = true; }
  const int tiles = (g.M / G2_BM) * (g.N / G2_BN);
  hipLaunchKernelGGL(k, dim3(tiles > 256 && tiles % 256 == 0 && EPI == EPI_BIAS_GELU ? 256 : tiles), dim3(G2_NW * 64), G2_LDS_BYTES, st, g);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 400;
  run_case("gemm3s plain (q)", g3<EPI_BIAS, 0>, 8192, 1152, 1152, 0, 256, reps);
  run_case("gemm3s rmw (attn1_out)", g3<EPI_BIAS_CACHE_RESID, 1>, 8192, 1152, 1152, 3, 256, reps);
  run_case("gemm3s rmw (ff2)", g3<EPI_BIAS_CACHE_RESID, 5>, 8192, 1152, 4608, 3, 256, reps / 2);
  run_case("gemm2w gelu (ff1)", g2<EPI_BIAS_GELU, 4>, 8192, 4608, 1152, 1, 256, reps / 2);
  run_case("gemm2w plain (kv)", g2<EPI_BIAS, 9>, 8192, 2304, 1152, 0, 256, reps);
  return 0;
}
