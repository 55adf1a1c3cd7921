// probe_ring_depth.hip -- does the operand stream of a one-wave-per-SIMD GEMM tile (gemm2w4.h: 256 x 288, four waves) run faster with more bytes in flight?
// Every workgroup (256 threads, one per CU) walks the K loop of its tiles as the GEMM does -- LDS-DMA pieces of 1 KiB, one barrier per stage, counted vmcnt -- but issues
// no MFMA and no fragment read.  Variants: SLOTS ring slots of RB bytes of K per row each (RB 128: the product's stages, 68 KiB; RB 64: half stages, 34 KiB), AHEAD stages
// in flight while one is "consumed".  Output: GB/s per CU (wall clock) with every CU running, FF1's shape (8192 x 4608 x 1152, operands L2 / Infinity-Cache resident).
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe_ring_depth.hip -o build_tmp/probe_ring_depth
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Args { const char* A; const char* W; int M, N, K; unsigned long long* dbg; int spin; };

template <int RB, int SLOTS, int AHEAD, int NB> __global__ __launch_bounds__(256) void probe(const Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = 256 + NB, STAGE = ROWS * RB, RPP = 1024 / RB, PIECES = ROWS / RPP, PER_WAVE = (PIECES + 3) / 4, CPR = RB / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / NB, nwg = (p.M / 256) * tiles_n;
  const int nk = p.K * 2 / RB;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  double bytes = 0;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
    const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const char* src[PER_WAVE];
    int dst[PER_WAVE];
    bool on[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int piece = wave * PER_WAVE + i;
      on[i] = piece < PIECES;
      const int pc = on[i] ? piece : PIECES - 1;
      const int rr = pc * RPP + lane / CPR;
      const int c = (lane % CPR) ^ ((rr >> 1) & (CPR - 1));
      const size_t row = rr < 256 ? (size_t)tm * 256 + rr : (size_t)tn * NB + (rr - 256);
      src[i] = (rr < 256 ? p.A : p.W) + row * (size_t)p.K * 2 + c * 16;
      dst[i] = pc * 1024;
    }
    auto issue = [&](int kt, int slot) {
#pragma unroll
      for (int i = 0; i < PER_WAVE; ++i)
        if (on[i]) __builtin_amdgcn_global_load_lds((glb_void*)(src[i] + (size_t)kt * RB), (lds_void*)(smem + slot * STAGE + dst[i]), 16, 0, 0);
    };
    for (int a = 0; a < AHEAD && a < nk; ++a) issue(a, a % SLOTS);
    for (int kt = 0; kt < nk; ++kt) {
      // stage kt has landed when at most the min(AHEAD - 1, nk - 1 - kt) newer stages are outstanding (every wave issues PER_WAVE pieces per stage here: PIECES % 4 == 0)
      const int newer = nk - 1 - kt;
      if (AHEAD == 1 || newer <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (AHEAD == 2 || newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
      else if (AHEAD == 3 || newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER_WAVE) : "memory");
      __builtin_amdgcn_s_barrier();
      for (int s = 0; s < p.spin; ++s) asm volatile("s_nop 15");  // stand-in for the stage's MFMA time (spin x 16 cycles)
      __builtin_amdgcn_s_barrier();   // the slot refilled below was read by everybody
      if (kt + AHEAD < nk) issue(kt + AHEAD, (kt + AHEAD) % SLOTS);
    }
    bytes += (double)STAGE * nk;
    __syncthreads();
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { p.dbg[blockIdx.x * 2] = r1 - r0; p.dbg[blockIdx.x * 2 + 1] = (unsigned long long)bytes; }
}

template <int RB, int SLOTS, int AHEAD, int NB> void run(const char* name, Args a, int grid) {
  static_assert(AHEAD < SLOTS || AHEAD == SLOTS - 0, "");
  const int lds = SLOTS * (256 + NB) * RB;
  CK(hipFuncSetAttribute((const void*)probe<RB, SLOTS, AHEAD, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int w = 0; w < 4; ++w) hipLaunchKernelGGL((probe<RB, SLOTS, AHEAD, NB>), dim3(grid), dim3(256), lds, 0, a);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(grid * 2);
  CK(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> gbs;
  for (int i = 0; i < grid; ++i) gbs.push_back((double)h[2 * i + 1] / ((double)h[2 * i] * 10.0));  // bytes / ns
  std::sort(gbs.begin(), gbs.end());
  printf("%-44s spin %3d: %6.1f GB/s per CU (median), %5.1f TB/s over the chip, LDS %d KiB\n", name, a.spin, gbs[grid / 2], gbs[grid / 2] * grid / 1e3, lds / 1024);
}

int main(int argc, char** argv) {
  const int M = 8192, N = 4608, K = 1152;
  char *A, *W; unsigned long long* dbg;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&dbg, 1024 * 16));
  CK(hipMemset(A, 1, (size_t)M * K * 2)); CK(hipMemset(W, 1, (size_t)N * K * 2));
  for (int spin : {0, 60, 120}) {
    Args a{A, W, M, N, K, dbg, spin};
    run<128, 2, 1, 288>("2 slots x 68 KiB, 1 stage in flight (product)", a, 256);
    run<64, 4, 3, 288>("4 slots x 34 KiB, 3 half stages in flight", a, 256);
    run<64, 4, 2, 288>("4 slots x 34 KiB, 2 half stages in flight", a, 256);
    run<128, 2, 1, 256>("256 x 256: 2 slots x 64 KiB, 1 in flight", a, 256);
    run<64, 4, 3, 256>("256 x 256: 4 slots x 32 KiB, 3 in flight", a, 256);
    run<128, 3, 2, 144>("256 x 144: 3 slots x 50 KiB, 2 in flight", a, 256);
  }
  return 0;
}
