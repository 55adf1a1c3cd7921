// probe_mfma_shape_power.hip -- does the MFMA tile shape change what the chip sustains under its power cap?
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe_mfma_shape_power.hip -o build_tmp/probe_mfma_shape_power ; run on the GPU box.
//
// Register-only loops (no LDS, no global traffic): every wave keeps ACC independent accumulator chains and issues MFMAs back to back for ~0.5 s.
//   16x16x32 f16: 16 384 flop, 2 x 16 B of operands per lane and instruction  -> 2 KB of operand reads per 16 Kflop
//   32x32x16 f16: 32 768 flop, 2 x 16 B of operands per lane and instruction  -> half the register-file reads per flop
//   16x16x16 f16 (printed as 1616x1616): the K = 16 instruction of earlier generations, 8 192 flop -- same issue time as K = 32, i.e. half the rate
// Reported: TFLOP/s over the whole chip (wall clock) and the in-kernel clock (s_memtime cycles / s_memrealtime), with 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SHAPE> __global__ __launch_bounds__(512) void spin(float* out, unsigned long long* clk, int iters, int data) {
  h8 a, b;
  // operands: pseudo-random in [-1, 1) per lane and element (DATA 1) or tiny constants (DATA 0) -- switching power depends on the bits
  for (int i = 0; i < 8; ++i) {
    unsigned x = (threadIdx.x * 8u + i + 1u) * 2654435761u + blockIdx.x * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    const float r = (float)(x & 0xffff) / 32768.0f - 1.0f, q = (float)(x >> 16) / 32768.0f - 1.0f;
    a[i] = data ? (_Float16)r : (_Float16)(0.001f * (threadIdx.x + i));
    b[i] = data ? (_Float16)(q * 0.05f) : (_Float16)(0.002f * (i + 1));
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float keep = 0.f;
  if (SHAPE == 16) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) keep += acc[i][0];
  } else if (SHAPE == 1616) {  // the K = 16 instruction of earlier CDNA generations: 8 192 flop
    f4 acc[8];
    const h4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) keep += acc[i][0];
  } else {
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) keep += acc[i][0];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  if (keep == 123.456f) out[0] = keep;
}

template <int SHAPE> void run(int threads, int iters, int data) {
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, 4)); CK(hipMalloc(&clk, 256 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((spin<SHAPE>), dim3(256), dim3(threads), 0, 0, out, clk, iters, data);  // warm: clocks settle
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((spin<SHAPE>), dim3(256), dim3(threads), 0, 0, out, clk, iters, data);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[512]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  const double per_wave = (SHAPE == 16 ? 32.0 * 16384.0 : SHAPE == 1616 ? 32.0 * 8192.0 : 16.0 * 32768.0) * iters;  // flop per wave
  const double flop = per_wave * (threads / 64) * 256;
  printf("%dx%d %s %d waves/SIMD: %7.1f TFLOP/s  in-kernel clock %.2f GHz  (%.0f ms)\n", SHAPE, SHAPE, data ? "random" : "tiny  ", threads / 256, flop / (ms * 1e-3) / 1e12,
         (double)h[0] / ((double)h[1] / 100.0) / 1e3, ms);
}

int main() {
  const int iters = 4000000;  // ~0.9 s per launch at full rate: the power governor has settled
  for (int data = 0; data < 2; ++data) {
    run<16>(256, iters, data); run<32>(256, iters, data);
    run<16>(512, iters / 2, data); run<32>(512, iters / 2, data);
    run<1616>(256, iters, data);  // printed as 1616x1616: v_mfma_f32_16x16x16_f16
  }
  return 0;
}
