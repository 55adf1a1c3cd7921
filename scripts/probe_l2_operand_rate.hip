// probe_l2_operand_rate.hip -- how fast can one CU pull the operand stream of the 256x144 GEMM tile (gemm3s.h) out of its XCD's L2?
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe_l2_operand_rate.hip -o build_tmp/probe_l2_operand_rate ; run on the GPU box.
//
// Every workgroup (512 threads, one per CU) walks the K loop of its tile exactly as the GEMM does -- per K-step 256 A rows + NB W rows
// of 128 bytes each, rows 2*K bytes apart, the same 4 x 8 tile patches per XCD -- but does no MFMA.  Variants of the load:
//   dma     global_load_lds_dwordx4 into a 3-stage LDS ring, counted vmcnt, one barrier per step (the GEMM's own staging)
//   dma_nb  the same without the per-step barrier (each wave only waits for its own pieces)
//   reg     global_load_dwordx4 to VGPRs, kept alive by an empty asm (no LDS at all): the ceiling of the L2 -> CU path for plain loads
//   regw    global_load_dwordx4 to VGPRs + ds_write_b128 into the ring (register staging)
// Output: bytes per clock per CU (s_memtime) and GB/s per CU (wall), median over workgroups, with every CU running.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args { const char* A; const char* W; int M, N, K, NB; unsigned long long* dbg; int reps; };

__device__ __forceinline__ void tile_of(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  if ((tiles_m & 3) == 0 && (tiles_n & 7) == 0) {
    const int patch = bid >> 5, within = bid & 31, pn = tiles_n >> 3;
    const int pm = patch / pn, pq = patch - pm * pn;
    tm = pm * 4 + (within >> 3); tn = pq * 8 + (within & 7);
  } else { tm = bid / tiles_n; tn = bid - tm * tiles_n; }
}

template <int MODE, int NB> __global__ __launch_bounds__(512) void probe(const Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = (256 + NB) * 128, PIECES = (256 + NB) / 8;  // 1-KiB pieces (8 rows x 128 B) per stage
  constexpr int PER_WAVE = (PIECES + 7) / 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  tile_of(blockIdx.x, p.M / 256, p.N / NB, tm, tn);
  const char* src[PER_WAVE];
  int dst[PER_WAVE];
  bool on[PER_WAVE];
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int piece = wave + i * 8;
    on[i] = piece < PIECES;
    const int pc = on[i] ? piece : PIECES - 1;
    const int r = pc * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const size_t row = r < 256 ? (size_t)tm * 256 + r : (size_t)tn * NB + (r - 256);
    src[i] = (r < 256 ? p.A : p.W) + row * (size_t)p.K * 2 + c * 16;
    dst[i] = pc * 1024;
  }
  const int nk = p.K / 64;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  u32x4 sink = {0, 0, 0, 0};
  for (int rep = 0; rep < p.reps; ++rep) {
    if (MODE == 0 || MODE == 1) {
      auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i)
          if (on[i]) __builtin_amdgcn_global_load_lds((glb_void*)(src[i] + (size_t)kt * 128), (lds_void*)(smem + stage * STAGE + dst[i]), 16, 0, 0);
      };
      issue(0, 0);
      issue(1, 1);
      int stage = 0;
      for (int kt = 0; kt < nk; ++kt) {
        // all but the newest stage's pieces have landed (PER_WAVE pieces per stage; the odd last piece only on some waves: wait conservatively)
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE - 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 0) __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2, (stage + 2) % 3);
        stage = (stage + 1) % 3;
      }
    } else {
      // register staging, 2 steps in flight
      u32x4 v[2][PER_WAVE];
      auto load = [&](int kt, int b) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) v[b][i] = *reinterpret_cast<const u32x4*>(src[i] + (size_t)kt * 128);
      };
      load(0, 0);
      load(1, 1);
      for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i)
              if (on[i]) *reinterpret_cast<u32x4*>(smem + ((kt + b) % 3) * STAGE + dst[i] + lane * 16) = v[b][i];
            __builtin_amdgcn_s_barrier();
          } else {
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) asm volatile("" ::"v"(v[b][i]));
          }
          if (kt + b + 2 < nk) load(kt + b + 2, b);
        }
      }
    }
    __syncthreads();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { p.dbg[blockIdx.x * 2] = c1 - c0; p.dbg[blockIdx.x * 2 + 1] = r1 - r0; }
  if (sink.x == 0x1234567) p.dbg[0] = 0;
}

template <int MODE, int NB> void run(const char* name, Args a, int grid, int lds) {
  CK(hipFuncSetAttribute((const void*)probe<MODE, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, NB>), dim3(grid), dim3(512), lds, 0, a);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(grid * 2);
  CK(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, wall;
  for (int i = 0; i < grid; ++i) { cyc.push_back((double)h[2 * i]); wall.push_back((double)h[2 * i + 1] / 100.0); }
  std::sort(cyc.begin(), cyc.end());
  std::sort(wall.begin(), wall.end());
  const double bytes = (double)(256 + NB) * 128.0 * (a.K / 64) * a.reps;
  const double c = cyc[grid / 2], us = wall[grid / 2];
  printf("%-7s NB=%3d grid=%3d: %6.1f B/clk/CU  %6.1f GB/s/CU  (%.0f cycles/K-step, clock %.2f GHz, chip %.1f TB/s)\n", name, NB, grid, bytes / c, bytes / us / 1e3,
         c / ((double)(a.K / 64) * a.reps), c / us / 1e3, bytes / us * grid / 1e6);
}

int main(int argc, char** argv) {
  const int M = 8192, K = argc > 1 ? atoi(argv[1]) : 1152;
  const int Nmax = 4608;
  char *A, *W;
  unsigned long long* dbg;
  CK(hipMalloc(&A, (size_t)M * K * 2));
  CK(hipMalloc(&W, (size_t)Nmax * K * 2));
  CK(hipMalloc(&dbg, 4096 * 16));
  CK(hipMemset(A, 1, (size_t)M * K * 2));
  CK(hipMemset(W, 2, (size_t)Nmax * K * 2));
  for (int grid : {8, 256}) {
    Args a{A, W, M, 1152, K, 144, dbg, 40};
    const int lds144 = 3 * (256 + 144) * 128;
    if (grid == 8) a.M = 256;
    run<0, 144>("dma", a, grid, lds144);
    run<1, 144>("dma_nb", a, grid, lds144);
    run<2, 144>("reg", a, grid, lds144);
    run<3, 144>("regw", a, grid, lds144);
    Args b{A, W, M, 2304, K, 288, dbg, 40};
    if (grid == 8) b.M = 256;
    const int lds288 = 2 * (256 + 288) * 128;  // the ring of gemm2w holds 2 stages: the probe's 3-stage index wraps inside it only for the timing pattern
    (void)lds288;
  }
  return 0;
}
