// probe_gemm_energy_budget.hip -- what the ingredients of a GEMM K loop cost under the power cap (companion of probe_mfma_shape_power.hip).
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe_gemm_energy_budget.hip -o build_tmp/probe_gemm_energy_budget ; run on the GPU box.
//
// Every wave of 256 workgroups x 8 waves runs, for ~1 s, iterations of 36 MFMAs 16x16x32 f16 on a 4 x 9 grid of operand fragments (the wave tile of
// gemm2w: 64 x 144 per 32-deep k step), all pseudo-random fp16.  Modes add what a real K loop adds:
//   regs     operands stay in registers (a different pair per MFMA: the operand buses toggle, unlike probe_mfma_shape_power's single pair)
//   lds13    + 13 ds_read_b128 per iteration feeding the operands (gemm2w: 0.36 fragment reads per MFMA)
//   lds22    + 22 ds_read_b128 (gemm3s, 32 x 144 wave tile: 0.61 per MFMA)
//   lds13dma + 13 reads + 4 global_load_lds_dwordx4 (1 KB each: gemm2w's L2 -> LDS share per wave and k step is 4.25 KB) from an L2-resident buffer
//   lds22dma + 22 reads + 3 global_load_lds_dwordx4 (gemm3s: 3.1 KB)
// Output: TFLOP/s over the chip and the in-kernel clock.  The LDS / DMA modes use each fragment right after its read (no prefetch distance), so their
// TFLOP/s is latency-bound and says nothing about energy; what they show is that the clock stays where the MFMA-only mode puts it.  The cycle count is
// workgroup 0's OLDEST wave: in `regs` it is half the kernel time -- the SIMD arbitrates oldest-first and the younger wave runs after it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; return x; }

template <int NREAD, int NDMA> __global__ __launch_bounds__(512) void spin(float* out, unsigned long long* clk, const char* gbuf, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 64 KB of fragments + 32 KB DMA landing zone
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 4; i += 512) {
    const unsigned x = hash(i + blockIdx.x * 7919u);
    _Float16 lo = (_Float16)((float)(x & 0xffff) / 32768.0f - 1.0f), hi = (_Float16)(((float)(x >> 16) / 32768.0f - 1.0f) * 0.05f);
    reinterpret_cast<_Float16*>(smem)[2 * i] = lo;
    reinterpret_cast<_Float16*>(smem)[2 * i + 1] = hi;
  }
  __syncthreads();
  h8 a[4], b[9];
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(smem + ((wave * 13 + i) * 1024 + lane * 16) % 65536);
  for (int j = 0; j < 9; ++j) b[j] = *reinterpret_cast<const h8*>(smem + ((wave * 13 + 4 + j) * 1024 + lane * 16) % 65536);
  f4 acc[4][9];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 9; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  const char* gsrc = gbuf + ((size_t)blockIdx.x * 8 + wave) * 65536 + lane * 16;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned off = wave * 1024u;
  for (int it = 0; it < iters; ++it) {
    // fragment reads of this iteration (their values replace operand registers: the MFMAs below see fresh bits every time)
#pragma unroll
    for (int r = 0; r < NREAD; ++r) {
      const h8 v = *reinterpret_cast<const h8*>(smem + ((off + r * 1024u) & 65535u) + lane * 16);
      if (r < 4) a[r] = v;
      else if (r < 13) b[r - 4] = v;
      else a[(r - 13) & 3] = v;  // gemm3s re-reads: more reads per MFMA, same operand count
    }
#pragma unroll
    for (int d = 0; d < NDMA; ++d)
      __builtin_amdgcn_global_load_lds((glb_void*)(gsrc + ((it * NDMA + d) & 63) * 1024), (lds_void*)(smem + 65536 + (wave * 4 + d) * 1024), 16, 0, 0);
    off += 13 * 1024u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (NDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float keep = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 9; ++j) keep += acc[i][j][0];
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  if (keep == 123.456f) out[0] = keep;
}

template <int NREAD, int NDMA> void run(const char* name, const char* gbuf, int iters) {
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, 4)); CK(hipMalloc(&clk, 256 * 16));
  const int lds = 65536 + 32768;
  CK(hipFuncSetAttribute((const void*)spin<NREAD, NDMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((spin<NREAD, NDMA>), dim3(256), dim3(512), lds, 0, out, clk, gbuf, iters / 4);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((spin<NREAD, NDMA>), dim3(256), dim3(512), lds, 0, out, clk, gbuf, iters);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  const double flop = 36.0 * 16384.0 * iters * 8 * 256;
  printf("%-9s %7.1f TFLOP/s  in-kernel clock %.2f GHz  %.0f cycles per 36 MFMAs (576 = back to back)  (%.0f ms)\n", name, flop / (ms * 1e-3) / 1e12,
         (double)h[0] / ((double)h[1] / 100.0) / 1e3, (double)h[0] / iters, ms);
}

int main() {
  char* gbuf; CK(hipMalloc(&gbuf, (size_t)256 * 8 * 65536)); CK(hipMemset(gbuf, 0x5a, (size_t)256 * 8 * 65536));
  const int iters = 1000000;
  run<0, 0>("regs", gbuf, iters);
  run<13, 0>("lds13", gbuf, iters);
  run<22, 0>("lds22", gbuf, iters);
  run<13, 4>("lds13dma", gbuf, iters);
  run<22, 3>("lds22dma", gbuf, iters);
  return 0;
}
