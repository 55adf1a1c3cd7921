// probe_fewrows.hip -- the 64-row latency tiles of gemm3s.h at batch-1 rows (M = 512) with weights that rotate through more bytes than the Infinity Cache holds, as the
// 28 layers of a step do: microseconds per launch and the -DG_STAMP cycles of fill / K loop / epilogue, for the 64-row tile on four waves (the product's) and a 128-row tile on eight (more waves issuing LDS-DMA per tile; the six-stage ring of round 5 is described in DESIGN 5f).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only -DECAD_DIAG -DG_STAMP scripts/probe_fewrows.hip -o build_tmp/probe_fewrows
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../ecad_amd/csrc/gemm3s.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int EPI, int TAG, int BM, int NW, int NSTG> static void lat(const GemmArgs<f16_t>& g, hipStream_t st) {
  static bool attr = false;
  auto k = gemm3s_kernel<f16_t, EPI, TAG, BM, NW, NSTG>;
  const int lds = NSTG * (BM + G3_BN) * 128;
  if (!attr) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
  hipLaunchKernelGGL(k, dim3(((g.M + BM - 1) / BM) * (g.N / G3_BN)), dim3(NW * 64), lds, st, g);
}

template <typename F> static void run_case(const char* name, F launch_on, int M, int N, int K, int epi, int reps) {
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  const int nsets = (int)(400e6 / ((double)N * K * 2)) + 1;
  std::vector<_Float16> ha((size_t)M * K), hw((size_t)N * K), hr((size_t)M * N);
  for (auto& v : ha) v = (_Float16)nd(rng);
  for (auto& v : hw) v = (_Float16)(0.03f * nd(rng));
  for (auto& v : hr) v = (_Float16)nd(rng);
  std::vector<float> hb(N);
  for (auto& v : hb) v = 0.1f * nd(rng);
  _Float16 *A, *W, *C, *R;
  float* bias;
  unsigned long long* dbg;
  CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&W, hw.size() * 2 * nsets)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&R, (size_t)M * N * 2));
  CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&dbg, 8 * 8192 * 8));
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  for (int s = 0; s < nsets; ++s) CK(hipMemcpy(W + (size_t)s * N * K, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(R, hr.data(), hr.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dbg, 0, 8 * 8192 * 8));
  GemmArgs<f16_t> g{};
  g.A = A; g.W = W; g.bias = bias; g.C = C; g.resid = epi == 3 ? R : nullptr; g.gate_table = nullptr; g.gate_mod = (const f16_t*)dbg; g.rows_per_batch = 256;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N; g.variant = 1; g.epi_direct = 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int set = 0;
  auto launch = [&]() { GemmArgs<f16_t> a = g; a.W = W + (size_t)set * N * K; set = (set + 1) % nsets; launch_on(a, (hipStream_t)0); };
  for (int i = 0; i < 50; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
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
  const double us = ms * 1e3 / reps;
  const int steps = K / 64;
  printf("%-34s M=%d N=%d K=%d epi=%d (%d weight sets): %7.2f us/launch | tiles %zu: fill %.0f  loop %.0f (%.0f/step)  epilogue %.0f cycles, tile wall %.2f us, clock %.3f GHz\n", name, M, N, K, epi, nsets, us,
         fill.size(), med(fill), med(loop), med(loop) / steps, med(epi_c), med(wall), med(clk));
  fflush(stdout);
  CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(R)); CK(hipFree(bias)); CK(hipFree(dbg));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 600;
  for (int M : {512, 2048}) {
    run_case("plain 64x144 3-stage", lat<EPI_BIAS, 0, 64, 4, 3>, M, 1152, 1152, 0, reps);
    run_case("plain 128x144 8 waves", lat<EPI_BIAS, 0, 128, 8, 3>, M, 1152, 1152, 0, reps);
    run_case("rmw 64x144 3-stage", lat<EPI_BIAS_CACHE_RESID, 1, 64, 4, 3>, M, 1152, 1152, 3, reps);
    run_case("rmw 128x144 8 waves", lat<EPI_BIAS_CACHE_RESID, 1, 128, 8, 3>, M, 1152, 1152, 3, reps);
    run_case("gelu N=4608 64x144 3-stage", lat<EPI_BIAS_GELU, 4, 64, 4, 3>, M, 4608, 1152, 1, reps);
    run_case("gelu N=4608 128x144 8 waves", lat<EPI_BIAS_GELU, 4, 128, 8, 3>, M, 4608, 1152, 1, reps);
    run_case("plain 256x144 8 waves", lat<EPI_BIAS, 0, 256, 8, 3>, M, 1152, 1152, 0, reps);
  }
  return 0;
}
