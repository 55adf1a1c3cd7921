// probe_mfma_valu_overlap.hip -- do the MFMAs of one wave overlap the VALU work of the OTHER wave on the same SIMD (gfx950)?
// One workgroup of 8 waves per CU (wave w and w + 4 share a SIMD).  mode bit 0: waves 0-3 run N iterations of 16 independent v_mfma_f32_16x16x32_bf16;
// bit 1: waves 4-7 run N iterations of a VALU block: kind 0 = 64 v_fma_f32, kind 1 = 16 v_exp_f32 + 16 v_fma_f32, kind 2 = 64 v_pk_fma_f32.
// Prints cycles per iteration (s_memtime, wave 0 / wave 4 of workgroup 0) for MFMA alone, VALU alone, both.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe_mfma_valu_overlap.hip -o build_tmp/probe_mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND> __global__ __launch_bounds__(512) void probe(int mode, int n, float* out, unsigned long long* dbg) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (mode & 1) {
      f32x4 acc[16];
      for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
      for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
      float s = 0.f;
      for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    }
  } else if (mode & 2) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.01f * (lane + i);
    for (int it = 0; it < n; ++it) {
      if (KIND == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
      } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[i], 0.5f, -1.0f));
      } else {
        f32x2* v2 = reinterpret_cast<f32x2*>(v);
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) v2[i] = __builtin_elementwise_fma(v2[i], (f32x2){0.999f, 0.999f}, (f32x2){0.001f, 0.001f});
      }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) dbg[wave >> 2] = t1 - t0;
}

template <int KIND> void run(const char* what, int n) {
  float* out; unsigned long long* dbg;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&dbg, 16));
  for (int mode = 1; mode <= 3; ++mode) {
    CK(hipMemset(dbg, 0, 16));
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(512), 0, 0, mode, n, out, dbg);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, dbg, 16, hipMemcpyDeviceToHost));
    printf("%-34s mode %d (%s): MFMA wave %.1f cycles/iter (16 MFMAs = 256 pipe cycles), VALU wave %.1f cycles/iter\n", what, mode,
           mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", (double)h[0] / n, (double)h[1] / n);
  }
  CK(hipFree(out)); CK(hipFree(dbg));
}

int main() {
  const int n = 20000;
  run<0>("64 v_fma_f32 per iteration", n);
  run<1>("16 v_exp_f32 + 16 v_fma_f32", n);
  run<2>("64 v_pk_fma_f32", n);
  return 0;
}
