// probe_gemm4c.hip -- round 4, the review's candidate (b) for the one-round N = K = 1152 class: workgroups small enough that TWO fit a CU, so that the tile of a
// kernel on another stream (another batch in flight) can run its K loop while this one sits in its pipeline fill, epilogue and store drain.  Same 256 x 144 tile
// as gemm3s.h (same L2->LDS bytes per flop), but FOUR waves of 64 x 144 (0.36 fragment reads per MFMA instead of 0.61), BK = 32, a 3-stage ring of 25 600-byte
// stages: 76 800 B of ring / 77 824 B with the C tile, two workgroups per CU, 256 registers per wave.  LDS rows are 64 B (4 chunks of 16 B), chunk c of row r lives
// at chunk position c ^ G[(r >> 2) & 3], G = {0, 2, 3, 1}: conflict-free for the four 16-lane groups of ds_read_b128.
// Plain epilogue (bias, stores through a C tile) and the read-modify-write one (cache slot + h += out, no gate) -- enough to time; outputs are compared with gemm3s.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only scripts/probe_gemm4c.hip -o build_tmp/probe_gemm4c
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../ecad_amd/csrc/gemm3s.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int C4_BM = 256, C4_BN = 144, C4_BK = 32, C4_NW = 4, C4_NSTG = 3;
constexpr int C4_A_BYTES = C4_BM * 64, C4_STAGE = (C4_BM + C4_BN) * 64;
constexpr int C4_CS = C4_BN + 8;
constexpr int C4_LDS = C4_BM * C4_CS * 2 > C4_NSTG * C4_STAGE ? C4_BM * C4_CS * 2 : C4_NSTG * C4_STAGE;
static_assert(2 * C4_LDS <= 160 * 1024, "two workgroups per CU");
#ifndef C4_PN
#define C4_PN 4  // fragment columns of a step whose MFMAs run after the NEXT barrier
#endif

template <typename T, int EPI>
__global__ __launch_bounds__(C4_NW * 64, 2) void gemm4c_kernel(const GemmArgs<T> p) {
  constexpr int FM = 4, FN = 9, PN = C4_PN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int tiles_n = p.N / C4_BN, tiles_m = p.M / C4_BM;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  tile_coords(bid, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * C4_BM, n0 = tn * C4_BN;
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  // LDS-DMA pieces: 1 KiB = 16 rows x 64 B.  A: 16 pieces, wave w issues 4w .. 4w + 3 (its own 64 rows); W: 9 pieces, wave w issues w, w + 4 and wave 0 also piece 8.
  // sources as 32-bit byte offsets from a wave-uniform base that advances with k (SGPR base + VGPR offset addressing: 7 registers instead of 14)
  uint32_t a_o[4], b_o[3];
  {
    const int r = lane >> 2, cp = lane & 3;
    const int c = cp ^ ((0x1320 >> (4 * ((r >> 2) & 3))) & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) a_o[i] = (uint32_t)(((4 * wave + i) * 16 + r) * p.lda + c * 8) * (uint32_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int piece = wave + 4 * i;
      piece = piece < 9 ? piece : 8;
      b_o[i] = (uint32_t)((piece * 16 + r) * p.ldw + c * 8) * (uint32_t)sizeof(T);
    }
  }
  const char* a_k = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
  const char* b_k = reinterpret_cast<const char*>(p.W + (size_t)n0 * p.ldw);
  auto issue = [&](int stage) {
    char* abase = smem + stage * C4_STAGE;
    char* bbase = abase + C4_A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_k + a_o[i], abase + (4 * wave + i) * 1024);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int piece = wave + 4 * i;
      if (piece < 9) glds16(b_k + b_o[i], bbase + piece * 1024);
    }
    a_k += C4_BK * sizeof(T);
    b_k += C4_BK * sizeof(T);
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frag_off = l15 * 64 + ((g ^ ((0x1320 >> (4 * (l15 >> 2))) & 3)) * 16);
  const int a_off = wave * 4 * 1024 + frag_off;
  const int b_off = C4_A_BYTES + frag_off;
  const int nk = p.K / C4_BK;
  issue(0);
  if (nk > 1) issue(1);
  f32x4 bvs[FN];
  auto step_top = [&](int kt) {
    if (kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // the newest stage's pieces (6 per wave, 7 on wave 0) may still be in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // one K-step: fragments of stage kt into (afN, wfN); the MFMAs of the previous step's last PN fragment columns (afC, wfC) run first, while the new fragments travel
  auto step = [&](int kt, Pack8 (&afN)[FM], Pack8 (&wfN)[FN], const Pack8 (&afC)[FM], const Pack8 (&wfC)[FN]) {
    int stage = kt % C4_NSTG;
    step_top(kt);
    const char* sb = smem + stage * C4_STAGE;
#pragma unroll
    for (int i = 0; i < FM; ++i) afN[i] = lds_read16(sb + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < FN; ++j) wfN[j] = lds_read16(sb + b_off + j * 1024);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nk) {
      int ns = stage + 2;
      ns = ns >= C4_NSTG ? ns - C4_NSTG : ns;
      issue(ns);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt > 0) {
#pragma unroll
      for (int j = FN - PN; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wfC[j], afC[i], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FN - PN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wfN[j], afN[i], acc[i][j]);
  };
  Pack8 afA[FM], wfA[FN], afB[FM], wfB[FN];
  {
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      step(kt, afA, wfA, afB, wfB);
      step(kt + 1, afB, wfB, afA, wfA);
    }
    if (kt < nk) {  // odd step count
      step(kt, afA, wfA, afB, wfB);
#pragma unroll
      for (int j = FN - PN; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wfA[j], afA[i], acc[i][j]);
    } else {
#pragma unroll
      for (int j = FN - PN; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wfB[j], afB[i], acc[i][j]);
    }
  }
  const unsigned long long st_c2 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    bvs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bvs[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * j + 4 * g);
  }
  constexpr int CPR = C4_BN / 8, NCH = 64 * CPR / 64;  // 18 chunks of 16 B per lane
  __syncthreads();  // the C tile covers the whole ring: every wave has left the K loop
  T* ctile = reinterpret_cast<T*>(smem);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = 16 * j + 4 * g;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ml = wave * 64 + 16 * i + l15;
      Pack4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) float_to_elem4<T>(o, r, acc[i][j][r] + bvs[j][r]);
      *reinterpret_cast<u32x2*>(ctile + ml * C4_CS + nl) = o.u;
    }
  }
  // residual chunks: requested once the accumulators are dead (the other workgroup of the CU covers their latency)
  Pack8 hpre[EPI == EPI_BIAS_CACHE_RESID ? NCH : 1];
  if constexpr (EPI == EPI_BIAS_CACHE_RESID) {
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int id = lane + it * 64;
      const int rl = id / CPR, cchunk = id - rl * CPR;
      hpre[it] = gload16(p.resid + (size_t)(m0 + wave * 64 + rl) * p.ldr + n0 + cchunk * 8);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int id = lane + it * 64;
    const int rl = id / CPR, cchunk = id - rl * CPR;
    const int ml = wave * 64 + rl;
    const int m = m0 + ml, n = n0 + cchunk * 8;
    Pack8 v = lds_read16(ctile + ml * C4_CS + cchunk * 8);
    if constexpr (EPI == EPI_BIAS_CACHE_RESID) {
      if (p.C) gstore16_nt(p.C + (size_t)m * p.ldc + n, v);
      float o[8], h[8];
      unpack8<T>(v, o);
      unpack8<T>(hpre[it], h);
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] += o[e];
      gstore16(p.resid + (size_t)m * p.ldr + n, pack8<T>(h));
    } else {
      gstore16(p.C + (size_t)m * p.ldc + n, v);
    }
  }
  if (!p.gate_table && p.gate_mod && wave == 0 && lane == 0) {  // stamps (the buffer rides in gate_mod as in the diagnostic build of the product kernels)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_c3 = __builtin_amdgcn_s_memtime(), st_r3 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<T*>(p.gate_mod)) + 4 * blockIdx.x;
    dbg[0] = st_c2 - st_c0; dbg[1] = st_c3 - st_c2; dbg[2] = st_r3 - st_r0; dbg[3] = st_c3 - st_c0;
  }
}

template <int EPI> static void g4(const GemmArgs<f16_t>& g, hipStream_t st) {
  static bool attr = false;
  auto k = gemm4c_kernel<f16_t, EPI>;
  if (!attr) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C4_LDS)); attr = true; }
  hipLaunchKernelGGL(k, dim3((g.M / C4_BM) * (g.N / C4_BN)), dim3(C4_NW * 64), C4_LDS, st, g);
}
template <int EPI, int TAG> static void g3(const GemmArgs<f16_t>& g, hipStream_t st) {
  static bool attr = false;
  auto k = gemm3s_kernel<f16_t, EPI, TAG>;
  if (!attr) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES)); attr = true; }
  hipLaunchKernelGGL(k, dim3((g.M / G3_BM) * (g.N / G3_BN)), dim3(G3_NW * 64), G3_LDS_BYTES, st, g);
}

struct Case {
  int M, N, K, epi;
  _Float16 *A, *W, *C, *R, *R0;
  float* bias;
  GemmArgs<f16_t> g;
};
static Case make_case(int M, int N, int K, int epi) {
  Case c{M, N, K, epi};
  std::mt19937 rng(7 + K + epi);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<_Float16> ha((size_t)M * K), hw((size_t)N * K), hr((size_t)M * N);
  for (auto& v : ha) v = (_Float16)nd(rng);
  for (auto& v : hw) v = (_Float16)(0.03f * nd(rng));
  for (auto& v : hr) v = (_Float16)nd(rng);
  std::vector<float> hb(N);
  for (auto& v : hb) v = 0.1f * nd(rng);
  CK(hipMalloc(&c.A, ha.size() * 2)); CK(hipMalloc(&c.W, hw.size() * 2)); CK(hipMalloc(&c.C, (size_t)M * N * 2)); CK(hipMalloc(&c.R, (size_t)M * N * 2)); CK(hipMalloc(&c.R0, (size_t)M * N * 2));
  CK(hipMalloc(&c.bias, N * 4));
  CK(hipMemcpy(c.A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(c.W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(c.R0, hr.data(), hr.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(c.bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  GemmArgs<f16_t> g{};
  g.A = c.A; g.W = c.W; g.bias = c.bias; g.C = c.C; g.resid = epi == 3 ? c.R : nullptr; g.rows_per_batch = 256;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N; g.variant = 1; g.epi_direct = 1;
  c.g = g;
  return c;
}
typedef void (*Launch)(const GemmArgs<f16_t>&, hipStream_t);

static void check(Case& c, Launch ref, Launch cand) {
  const size_t n = (size_t)c.M * c.N;
  std::vector<_Float16> c_ref(n), c_new(n), r_ref(n), r_new(n);
  CK(hipMemcpy(c.R, c.R0, n * 2, hipMemcpyDeviceToDevice));
  CK(hipMemset(c.C, 0, n * 2));
  ref(c.g, 0);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(c_ref.data(), c.C, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r_ref.data(), c.R, n * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c.R, c.R0, n * 2, hipMemcpyDeviceToDevice));
  CK(hipMemset(c.C, 0, n * 2));
  cand(c.g, 0);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(c_new.data(), c.C, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r_new.data(), c.R, n * 2, hipMemcpyDeviceToHost));
  size_t dc = 0, dr = 0;
  for (size_t i = 0; i < n; ++i) { dc += memcmp(&c_ref[i], &c_new[i], 2) != 0; dr += memcmp(&r_ref[i], &r_new[i], 2) != 0; }
  printf("  check M=%d N=%d K=%d epi=%d: %zu of %zu outputs differ from gemm3s (C), %zu (h)\n", c.M, c.N, c.K, c.epi, dc, n, c.epi == 3 ? dr : (size_t)0);
}

// `reps` launches dealt to `ns` streams in turn, microseconds per launch (aggregate)
static Case* g_shift[4] = {nullptr, nullptr, nullptr, nullptr};  // phase shifters: one short launch in front of stream q's chain (K = q / ns of the timed kernels')
static Launch g_shift_launch = nullptr;
static double time_streams(const std::vector<std::pair<Case*, Launch>>& mix, int ns, int reps) {
  static hipStream_t st[4];
  static bool made = false;
  if (!made) { for (auto& q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking)); made = true; }
  hipEvent_t e0, e1, ev[4];
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& q : ev) CK(hipEventCreate(&q));
  double best = 1e30;
  for (int rnd = 0; rnd < 3; ++rnd) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int q = 0; q < ns; ++q) CK(hipStreamWaitEvent(st[q], e0, 0));
    // stream q runs the mix in order, rotated by q: neighbouring streams are at different kernels at any time (what batches in flight look like)
    if (g_shift_launch)
      for (int q = 1; q < ns; ++q) g_shift_launch(g_shift[ns == 2 ? 2 : (ns == 3 ? (q == 1 ? 1 : 3) : q)]->g, st[q]);
    for (int i = 0; i < reps; ++i)
      for (int q = 0; q < ns; ++q) {
        const auto& m = mix[(i + q) % mix.size()];
        m.second(m.first->g, st[q]);
      }
    for (int q = 0; q < ns; ++q) { CK(hipEventRecord(ev[q], st[q])); CK(hipStreamWaitEvent(0, ev[q], 0)); }
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms * 1e3 / (reps * ns));
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  for (auto& q : ev) CK(hipEventDestroy(q));
  return best;
}

// in-kernel clock and loop / epilogue cycles of gemm4c while `ns` streams run it back to back (median over the workgroups of the last launches)
static void stamped(Case& c, Launch cand, int ns) {
  static hipStream_t st[4];
  static bool made = false;
  if (!made) { for (auto& q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking)); made = true; }
  unsigned long long* dbg;
  const int nwg = (c.M / 256) * (c.N / 144);
  CK(hipMalloc(&dbg, (size_t)nwg * 4 * 8 * 4));
  CK(hipMemset(dbg, 0, (size_t)nwg * 4 * 8 * 4));
  GemmArgs<f16_t> gs[4];
  for (int q = 0; q < 4; ++q) { gs[q] = c.g; gs[q].gate_mod = (const f16_t*)(dbg + (size_t)q * nwg * 4); }
  CK(hipDeviceSynchronize());
  for (int i = 0; i < 600; ++i) for (int q = 0; q < ns; ++q) cand(gs[q], st[q]);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)nwg * 4 * ns);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> loop, epi, wall, clk;
  for (size_t b = 0; b < (size_t)nwg * ns; ++b) {
    const unsigned long long* d = &h[4 * b];
    if (!d[2]) continue;
    loop.push_back((double)d[0]); epi.push_back((double)d[1]); wall.push_back(d[2] / 100.0); clk.push_back((double)d[3] / (d[2] / 100.0) / 1e3);
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("    gemm4c stamps, %d stream(s): fill + loop %.0f cycles (%.0f per k32 step), epilogue %.0f, workgroup wall %.2f us, in-kernel clock %.3f GHz\n", ns, med(loop), med(loop) / (c.K / 32), med(epi), med(wall), med(clk));
  CK(hipFree(dbg));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 300;
  Case q = make_case(8192, 1152, 1152, 0), o1 = make_case(8192, 1152, 1152, 3), f2 = make_case(8192, 1152, 4608, 3), s4 = make_case(32768, 1152, 1152, 3);
  check(q, g3<EPI_BIAS, 0>, g4<EPI_BIAS>);
  check(o1, g3<EPI_BIAS_CACHE_RESID, 1>, g4<EPI_BIAS_CACHE_RESID>);
  check(f2, g3<EPI_BIAS_CACHE_RESID, 5>, g4<EPI_BIAS_CACHE_RESID>);
  struct Row { const char* name; std::vector<std::pair<Case*, Launch>> ref, cand; double flops; };
  std::vector<Row> rows;
  rows.push_back({"plain 8192x1152x1152", {{&q, g3<EPI_BIAS, 0>}}, {{&q, g4<EPI_BIAS>}}, 2.0 * 8192 * 1152 * 1152});
  rows.push_back({"rmw   8192x1152x1152", {{&o1, g3<EPI_BIAS_CACHE_RESID, 1>}}, {{&o1, g4<EPI_BIAS_CACHE_RESID>}}, 2.0 * 8192 * 1152 * 1152});
  rows.push_back({"rmw   8192x1152x4608", {{&f2, g3<EPI_BIAS_CACHE_RESID, 5>}}, {{&f2, g4<EPI_BIAS_CACHE_RESID>}}, 2.0 * 8192 * 1152 * 4608});
  rows.push_back({"rmw  32768x1152x1152", {{&s4, g3<EPI_BIAS_CACHE_RESID, 1>}}, {{&s4, g4<EPI_BIAS_CACHE_RESID>}}, 2.0 * 32768 * 1152 * 1152});
  rows.push_back({"mix plain, rmw, rmw K=4608", {{&q, g3<EPI_BIAS, 0>}, {&o1, g3<EPI_BIAS_CACHE_RESID, 1>}, {&f2, g3<EPI_BIAS_CACHE_RESID, 5>}},
                  {{&q, g4<EPI_BIAS>}, {&o1, g4<EPI_BIAS_CACHE_RESID>}, {&f2, g4<EPI_BIAS_CACHE_RESID>}}, 2.0 * 8192 * 1152 * (1152 + 1152 + 4608) / 3.0});
  for (int ns : {1, 2, 3}) stamped(q, g4<EPI_BIAS>, ns);
  for (int ns : {1, 2}) stamped(o1, g4<EPI_BIAS_CACHE_RESID>, ns);
  for (int ns : {1, 2}) stamped(f2, g4<EPI_BIAS_CACHE_RESID>, ns);
  Case sh1 = make_case(8192, 1152, 288, 0), sh2 = make_case(8192, 1152, 576, 0), sh3 = make_case(8192, 1152, 864, 0);
  g_shift[1] = &sh1; g_shift[2] = &sh2; g_shift[3] = &sh3;
  for (int pass = 0; pass < 1; ++pass)
  for (auto& r : rows) {
    if (pass == 1 && r.ref.size() > 1) continue;
    printf("%s%-28s", pass ? "[streams phase-shifted by a short launch in front] " : "", r.name);
    for (int which = 0; which < 2; ++which) {
      const auto& mix = which ? r.cand : r.ref;
      g_shift_launch = pass ? (which ? (Launch)g4<EPI_BIAS> : (Launch)g3<EPI_BIAS, 0>) : nullptr;
      time_streams(mix, 1, reps / 4);  // warm
      printf(" | %s:", which ? "gemm4c (2 WG/CU)" : "gemm3s");
      for (int ns : {1, 2, 3, 4}) {
        const double us = time_streams(mix, ns, reps / ns);
        printf(" %6.2f us (%4.0f TF)", us, r.flops / us / 1e6);
      }
    }
    printf("   [1 / 2 / 3 / 4 streams, us per launch aggregate]\n");
    fflush(stdout);
  }
  return 0;
}
