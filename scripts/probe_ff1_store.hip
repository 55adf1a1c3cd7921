// probe_ff1_store.hip -- FF1 + GELU (gemm2w4_kernel, direct epilogue, LayerNorm fold) under the store policies of g2w4_store16: does writing the first tile of a
// workgroup's walk through to memory (sc1 / nt) while its second tile multiplies shorten the end-of-kernel write-back?  Timing only; every policy stores the same bytes
// (checked against policy 0).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only -DG2W4_STORE_PROBE scripts/probe_ff1_store.hip -o build_tmp/probe_ff1_store
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

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  const int M = 8192, N = 4608, K = 1152, nsets = 6;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
  for (auto& v : ha) v = f16_bits(rnd());
  for (auto& v : hw) v = f16_bits(rnd() * 0.03f);
  std::vector<float> hb(N), hc1(N), hst((size_t)8 * M * 2);
  for (auto& v : hb) v = rnd();
  for (auto& v : hc1) v = rnd();
  for (size_t i = 0; i < hst.size(); i += 2) { hst[i] = rnd() * 4.f; hst[i + 1] = 20.f + rnd(); }
  unsigned short *A[nsets], *W[nsets], *C[nsets + 1]; float *B, *C1, *ST;
  for (int s = 0; s < nsets; ++s) { CK(hipMalloc(&A[s], ha.size() * 2)); CK(hipMalloc(&W[s], hw.size() * 2)); CK(hipMemcpy(A[s], ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W[s], hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); }
  for (int i = 0; i <= nsets; ++i) CK(hipMalloc(&C[i], (size_t)M * N * 2));
  CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&C1, N * 4)); CK(hipMalloc(&ST, hst.size() * 4));
  CK(hipMemcpy(B, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(C1, hc1.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ST, hst.data(), hst.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)gemm2w4_kernel<f16_t, EPI_BIAS_GELU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, G2W4_LDS_BYTES));
  const int tiles = (M / 256) * (N / 288);
  auto launch = [&](int pol, int set, unsigned short* out) {
    GemmArgs<f16_t> g{};
    g.A = (const f16_t*)A[set]; g.W = (const f16_t*)W[set]; g.bias = B; g.C = (f16_t*)out; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.rows_per_batch = 256;
    g.epi_direct = 1; g.variant = 1 | (pol << 8);
    g.ln_stat = ST; g.ln_c1 = C1; g.ln_parts = 8; g.ln_eps = 1e-6f; g.ln_inv_d = 1.0f / 1152.f;
    hipLaunchKernelGGL((gemm2w4_kernel<f16_t, EPI_BIAS_GELU, 4>), dim3(tiles % cus == 0 ? cus : tiles), dim3(256), G2W4_LDS_BYTES, 0, g);
  };
  // policy code: low two bits = tiles before the last of a walk, next two = the last tile
  const int pols[] = {0x0, 0x1, 0x2, 0x3, 0x5, 0xA, 0xF, 0x4, 0x8};
  const char* names[] = {"write-back (product)", "first tile nt", "first tile sc1", "first tile sc0 sc1", "both nt", "both sc1", "both sc0 sc1", "last tile nt", "last tile sc1"};
  std::vector<unsigned short> ref((size_t)M * N), got((size_t)M * N);
  CK(hipMemset(C[nsets], 0xFF, (size_t)M * N * 2));
  launch(0, 0, C[nsets]);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), C[nsets], ref.size() * 2, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int np = sizeof(pols) / sizeof(pols[0]);
  std::vector<float> best(np, 1e9f);
  std::vector<size_t> diffs(np, 0);
  for (int i = 0; i < np; ++i) {
    CK(hipMemset(C[0], 0xFF, (size_t)M * N * 2));
    launch(pols[i], 0, C[0]);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), C[0], got.size() * 2, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < got.size(); ++k) diffs[i] += got[k] != ref[k];
  }
  for (int round = 0; round < 5; ++round)
    for (int i = 0; i < np; ++i) {
      int set = 0;
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) { launch(pols[i], set, C[set]); set = (set + 1) % nsets; }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (round) best[i] = std::min(best[i], ms / reps);
    }
  const double fl = 2.0 * M * N * K;
  for (int i = 0; i < np; ++i)
    printf("FF1 + GELU, LN fold, stores %-22s (0x%X): %7.2f us per launch %5.0f TF | words differing from write-back %zu%s\n", names[i], pols[i], best[i] * 1e3, fl / (best[i] * 1e-3) / 1e12, diffs[i],
           diffs[i] ? "  FAIL" : "");
  return 0;
}
