// probe_quant_rows.hip -- the fp8 row quantiser (fgemm8.h) alone, at the shapes the two engines launch it with: GB/s moved (read 2 B + write 1 B per element).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only scripts/probe_quant_rows.hip -o build_tmp/probe_quant_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../ecad_amd/csrc/fgemm8.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <typename T> static void run(const char* name, int rows, int K1, int K2, int nsets) {
  T* s1[8]; T* s2[8]; uint8_t* d[8]; float* sc;
  for (int i = 0; i < nsets; ++i) {
    CK(hipMalloc(&s1[i], (size_t)rows * K1 * 2)); CK(hipMemset(s1[i], 0x3c, (size_t)rows * K1 * 2));
    s2[i] = nullptr;
    if (K2) { CK(hipMalloc(&s2[i], (size_t)rows * K2 * 2)); CK(hipMemset(s2[i], 0x3c, (size_t)rows * K2 * 2)); }
    CK(hipMalloc(&d[i], (size_t)rows * (K1 + K2)));
  }
  CK(hipMalloc(&sc, rows * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  const int reps = 20;
  for (int round = 0; round < 4; ++round) {
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch_quant_rows_fp8<T>(s1[r % nsets], (int64_t)K1, K1, s2[r % nsets], (int64_t)K2, K2, d[r % nsets], (int64_t)(K1 + K2), sc, rows, (hipStream_t)0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (round) best = std::min(best, ms / reps);
  }
  const double bytes = (double)rows * (K1 + K2) * 3.0;
  printf("%-44s rows %6d K %5d + %5d: %8.1f us  %6.2f TB/s\n", name, rows, K1, K2, best * 1e3, bytes / (best * 1e-3) / 1e12);
  for (int i = 0; i < nsets; ++i) { CK(hipFree(s1[i])); if (s2[i]) CK(hipFree(s2[i])); CK(hipFree(d[i])); }
  CK(hipFree(sc));
}
int main() {
  run<f16_t>("alpha-256 attention output / residual stream", 8192, 1152, 0, 6);
  run<f16_t>("alpha-256 gelu(ff mid)", 8192, 4608, 0, 6);
  run<bf16_t>("FLUX attention output (joint + single rows)", 9216, 3072, 0, 4);
  run<bf16_t>("FLUX joint FF2 input, image rows", 8192, 12288, 0, 3);
  run<bf16_t>("FLUX proj_out input [attn | gelu(mlp)]", 9216, 3072, 12288, 3);
  return 0;
}
