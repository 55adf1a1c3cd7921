// probe_gemm2w4.hip -- gemm2w4.h (four waves, hand-placed K loop) against gemm2w_kernel on the PixArt shapes: same bits, and the time of both.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only scripts/probe_gemm2w4.hip -o build_tmp/probe_gemm2w4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "../ecad_amd/csrc/gemm2w4.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned rng = 7u;
static float rnd() { rng = rng * 1664525u + 1013904223u; return ((int)(rng >> 8) % 2001 - 1000) * 0.001f; }
static unsigned short f16_bits(float f) { _Float16 h = (_Float16)f; unsigned short v; std::memcpy(&v, &h, 2); return v; }

template <int EPI, int TAG> static void run(const char* name, int M, int N, int K, int direct, bool fold, int reps, int grid_cus) {
  const int nsets = 6;  // rotate operands: > 256 MiB in flight, as inside the pipeline
  std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
  for (auto& v : ha) v = f16_bits(rnd());
  for (auto& v : hw) v = f16_bits(rnd() * 0.03f);
  std::vector<float> hb(N), hc1(N), hst((size_t)8 * M * 2);
  for (auto& v : hb) v = rnd();
  for (auto& v : hc1) v = rnd();
  for (size_t i = 0; i < hst.size(); i += 2) { hst[i] = rnd() * 4.f; hst[i + 1] = 20.f + rnd(); }
  unsigned short *A[nsets], *W[nsets], *C[2]; float *B, *C1, *ST;
  for (int s = 0; s < nsets; ++s) { CK(hipMalloc(&A[s], ha.size() * 2)); CK(hipMalloc(&W[s], hw.size() * 2)); CK(hipMemcpy(A[s], ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W[s], hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); }
  for (int i = 0; i < 2; ++i) CK(hipMalloc(&C[i], (size_t)M * N * 2));
  CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&C1, N * 4)); CK(hipMalloc(&ST, hst.size() * 4));
  CK(hipMemcpy(B, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(C1, hc1.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ST, hst.data(), hst.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)gemm2w4_kernel<f16_t, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G2W4_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)gemm2w_kernel<f16_t, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
  const int tiles = (M / 256) * (N / 288);
  auto args = [&](int which, int set) {
    GemmArgs<f16_t> g{};
    g.A = (const f16_t*)A[set]; g.W = (const f16_t*)W[set]; g.bias = B; g.C = (f16_t*)C[which]; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.rows_per_batch = 256;
    g.epi_direct = direct;
    if (fold) { g.ln_stat = ST; g.ln_c1 = C1; g.ln_parts = 8; g.ln_eps = 1e-6f; g.ln_inv_d = 1.0f / 1152.f; }
    return g;
  };
  int set = 0;
  const int walk = (direct > (EPI == EPI_BIAS_GELU ? 0 : 1)) && tiles > grid_cus && tiles % grid_cus == 0;
  auto l4 = [&]() { hipLaunchKernelGGL((gemm2w4_kernel<f16_t, EPI, TAG>), dim3(tiles > grid_cus && tiles % grid_cus == 0 ? grid_cus : tiles), dim3(256), G2W4_LDS_BYTES, 0, args(0, set)); set = (set + 1) % nsets; };
  auto l8 = [&]() { hipLaunchKernelGGL((gemm2w_kernel<f16_t, EPI, TAG>), dim3(walk ? grid_cus : tiles), dim3(512), G2_LDS_BYTES, 0, args(1, set)); set = (set + 1) % nsets; };
  for (int i = 0; i < 2; ++i) CK(hipMemset(C[i], 0xFF, (size_t)M * N * 2));
  set = 0; l4(); set = 0; l8();
  CK(hipDeviceSynchronize());
  std::vector<unsigned short> c0((size_t)M * N), c1((size_t)M * N);
  CK(hipMemcpy(c0.data(), C[0], c0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), C[1], c1.size() * 2, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < c0.size(); ++i) diff += c0[i] != c1[i];
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best[2] = {1e9f, 1e9f};
  for (int round = 0; round < 4; ++round)
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) { if (w == 0) l4(); else l8(); }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (round) best[w] = std::min(best[w], ms / reps);
    }
#ifdef G2W4_STAMP
  {
    for (int r = 0; r < 200; ++r) l4();
    CK(hipDeviceSynchronize());
    const int nb = tiles > grid_cus && tiles % grid_cus == 0 ? grid_cus : tiles;
    std::vector<unsigned long long> st(4 * nb);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g2w4_stamp), st.size() * 8));
    std::vector<double> a, b, c, f;
    for (int i = 0; i < nb; ++i) { a.push_back((double)st[4 * i]); b.push_back((double)st[4 * i + 1]); c.push_back((double)st[4 * i + 2]); f.push_back((double)st[4 * i + 2] / (double)st[4 * i + 3] * 0.1); }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const int per = (tiles + nb - 1) / nb;
    printf("    stamps (median over workgroups, %d tiles each): K loop %.0f cycles/tile, epilogue %.0f cycles/tile, workgroup %.0f cycles, clock %.2f GHz\n", per, med(a) / per, med(b) / per, med(c), med(f));
  }
#endif
  const double fl = 2.0 * M * N * K;
  printf("%-30s M %5d N %5d K %5d: four waves %7.1f us %5.0f TF | gemm2w %7.1f us %5.0f TF | words differing %zu%s\n", name, M, N, K, best[0] * 1e3, fl / (best[0] * 1e-3) / 1e12, best[1] * 1e3,
         fl / (best[1] * 1e-3) / 1e12, diff, diff ? "  FAIL" : "");
  for (int s = 0; s < nsets; ++s) { CK(hipFree(A[s])); CK(hipFree(W[s])); }
  for (int i = 0; i < 2; ++i) CK(hipFree(C[i]));
  CK(hipFree(B)); CK(hipFree(C1)); CK(hipFree(ST));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 24;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  run<EPI_BIAS_GELU, 4>("FF1 + GELU, direct", 8192, 4608, 1152, 1, false, reps, cus);
  run<EPI_BIAS_GELU, 4>("FF1 + GELU, direct, LN fold", 8192, 4608, 1152, 1, true, reps, cus);
  run<EPI_BIAS, 9>("K,V wide, staged", 8192, 2304, 1152, 1, false, reps, cus);
  run<EPI_BIAS, 9>("K,V wide, staged, LN fold", 8192, 2304, 1152, 1, true, reps, cus);
  run<EPI_BIAS, 9>("K,V wide, direct", 8192, 2304, 1152, 2, false, reps, cus);
  run<EPI_BIAS_GELU, 4>("FF1 + GELU, staged", 8192, 4608, 1152, 0, false, reps, cus);
  run<EPI_BIAS_GELU, 4>("FF1 Sigma-1024 rows", 32768, 4608, 1152, 1, true, reps / 4 + 1, cus);
  return 0;
}
