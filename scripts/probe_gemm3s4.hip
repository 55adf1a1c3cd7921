// probe_gemm3s4.hip -- gemm3s4.h (256 x 144 tile on four waves, hand-placed K loop, read-modify-write epilogue) against gemm3s_kernel: cache slot, hidden states and row
// statistics bit for bit -- with and without a gate, with 0 / 1 / 2 replayed cached components, a bias or none -- and the time of both on 1 / 2 / 3 streams.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only scripts/probe_gemm3s4.hip -o build_tmp/probe_gemm3s4      (-DG3S4_STAMP: stamps)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "../ecad_amd/csrc/gemm3s4.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned rng = 11u;
static float rnd() { rng = rng * 1664525u + 1013904223u; return ((int)(rng >> 8) % 2001 - 1000) * 0.001f; }
static unsigned short f16_bits(float f) { _Float16 h = (_Float16)f; unsigned short v; std::memcpy(&v, &h, 2); return v; }
static unsigned short bf16_bits(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
template <typename T> static unsigned short bits_of(float f) { return std::is_same<T, bf16_t>::value ? bf16_bits(f) : f16_bits(f); }

static int failures = 0;

template <typename T, int TAG> static void rmw_case(const char* name, int M, int N, int K, bool gate, bool bias, int nx, int reps, int rows_per_batch = 256) {
  std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K), hr((size_t)M * N), hm((size_t)(M / rows_per_batch) * 6 * N), hx((size_t)M * N);
  for (auto& v : ha) v = bits_of<T>(rnd());
  for (auto& v : hw) v = bits_of<T>(rnd() * 0.03f);
  for (auto& v : hr) v = bits_of<T>(rnd());
  for (auto& v : hm) v = bits_of<T>(rnd());
  for (auto& v : hx) v = bits_of<T>(rnd());
  std::vector<float> hb(N), ht(N), ht2(N);
  for (auto& v : hb) v = rnd();
  for (auto& v : ht) v = rnd();
  for (auto& v : ht2) v = rnd();
  unsigned short *A, *W, *C[2], *R[2], *MOD, *X[2]; float *B, *TAB, *TAB2, *ST[2]; SrcDesc* XS;
  CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&W, hw.size() * 2)); CK(hipMalloc(&MOD, hm.size() * 2)); CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&TAB, N * 4)); CK(hipMalloc(&TAB2, N * 4));
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(MOD, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(TAB, ht.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(TAB2, ht2.data(), N * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&C[i], (size_t)M * N * 2)); CK(hipMalloc(&R[i], (size_t)M * N * 2)); CK(hipMalloc(&ST[i], (size_t)(N / 144) * M * 2 * 4)); CK(hipMalloc(&X[i], (size_t)M * N * 2));
    for (auto& v : hx) v = bits_of<T>(rnd());
    CK(hipMemcpy(X[i], hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  }
  {  // replayed components: the first gated (table + modulation row 4 of the sample), the second plain
    SrcDesc d[2] = {{X[0], TAB2, MOD + 4 * N}, {X[1], nullptr, nullptr}};
    CK(hipMalloc(&XS, sizeof(d))); CK(hipMemcpy(XS, d, sizeof(d), hipMemcpyHostToDevice));
  }
  CK(hipFuncSetAttribute((const void*)gemm3s4_kernel<T, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G3S4_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)gemm3s_kernel<T, EPI_BIAS_CACHE_RESID, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
  auto args = [&](int which) {
    GemmArgs<T> g{};
    g.A = (const T*)A; g.W = (const T*)W; g.bias = bias ? B : nullptr; g.C = (T*)C[which]; g.resid = (T*)R[which]; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N;
    g.rows_per_batch = rows_per_batch; g.variant = 1; g.epi_direct = 1; g.stat_out = ST[which];
    if (gate) { g.gate_table = TAB; g.gate_mod = (const T*)MOD + 2 * N; g.gate_mod_stride = 6 * N; }
    g.xsrc = XS; g.n_xsrc = nx; g.xsrc_mod_stride = 6 * N;
    return g;
  };
  const int tiles = (M / 256) * (N / 144);
  auto launch = [&](int which, hipStream_t st) {
    const GemmArgs<T> g = args(which);
    if (which == 0) {
      if (!g3s4_ok(g)) { printf("g3s4_ok refuses %s\n", name); exit(1); }
      hipLaunchKernelGGL((gemm3s4_kernel<T, TAG>), dim3(tiles), dim3(256), G3S4_LDS_BYTES, st, g);
    } else hipLaunchKernelGGL((gemm3s_kernel<T, EPI_BIAS_CACHE_RESID, TAG>), dim3(tiles), dim3(512), G3_LDS_BYTES, st, g);
  };
  for (int i = 0; i < 2; ++i) { CK(hipMemcpy(R[i], hr.data(), hr.size() * 2, hipMemcpyHostToDevice)); CK(hipMemset(C[i], 0xFF, (size_t)M * N * 2)); CK(hipMemset(ST[i], 0xFF, (size_t)(N / 144) * M * 8)); launch(i, 0); }
  CK(hipDeviceSynchronize());
  size_t diff[3] = {0, 0, 0};
  {
    std::vector<unsigned short> x((size_t)M * N), y((size_t)M * N);
    CK(hipMemcpy(x.data(), C[0], x.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), C[1], y.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < x.size(); ++i) {
      if (x[i] != y[i] && diff[0] < 4) printf("      cache[%zu, %zu]: four waves 0x%04x, gemm3s 0x%04x\n", i / N, i % N, x[i], y[i]);
      diff[0] += x[i] != y[i];
    }
    if (diff[0]) {
      size_t hist[4][9] = {};
      for (size_t i = 0; i < x.size(); ++i) if (x[i] != y[i]) hist[((i / N) % 256) / 64][((i % N) % 144) / 16]++;
      for (int q = 0; q < 4; ++q) { printf("      differing, rows %3d.. of the tile, by 16-column fragment:", 64 * q); for (int j = 0; j < 9; ++j) printf(" %zu", hist[q][j]); printf("\n"); }
    }
    CK(hipMemcpy(x.data(), R[0], x.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), R[1], y.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < x.size(); ++i) {
      if (x[i] != y[i] && diff[1] < 4) printf("      hidden[%zu, %zu]: four waves 0x%04x, gemm3s 0x%04x\n", i / N, i % N, x[i], y[i]);
      diff[1] += x[i] != y[i];
    }
    std::vector<unsigned> sa((size_t)(N / 144) * M * 2), sb(sa.size());
    CK(hipMemcpy(sa.data(), ST[0], sa.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), ST[1], sb.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < sa.size(); ++i) diff[2] += sa[i] != sb[i];
  }
#ifdef G3S4_STAMP
  if (reps > 0) {
    for (int r = 0; r < 50; ++r) launch(0, 0);
    CK(hipDeviceSynchronize());
    const int nb = std::min(tiles, 1024);
    std::vector<unsigned long long> sv(4 * nb);
    CK(hipMemcpyFromSymbol(sv.data(), HIP_SYMBOL(g3s4_stamp), sv.size() * 8));
    std::vector<double> a, b, c, f;
    for (int i = 0; i < nb; ++i) { a.push_back((double)sv[4 * i]); b.push_back((double)sv[4 * i + 1]); c.push_back((double)sv[4 * i + 2]); f.push_back((double)sv[4 * i + 2] / (double)sv[4 * i + 3] * 0.1); }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("    stamps (median over workgroups, wave 0): statement (fill + %d K-steps + C-tile staging) %.0f cycles = %.0f per K-step, row pass %.0f cycles, workgroup %.0f cycles, clock %.2f GHz\n", K / 64, med(a),
           med(a) / (K / 64), med(b), med(c), med(f));
  }
#endif
  hipStream_t st[3];
  for (auto& q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double us[2][3] = {};
  if (reps > 0)
    for (int which = 0; which < 2; ++which)
      for (int ns = 1; ns <= 3; ++ns) {
        float best = 1e9f;
        for (int round = 0; round < 3; ++round) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, 0));
          for (int q = 0; q < ns; ++q) CK(hipStreamWaitEvent(st[q], e0, 0));
          for (int i = 0; i < reps; ++i) launch(which, st[i % ns]);
          hipEvent_t ev[3];
          for (int q = 0; q < ns; ++q) { CK(hipEventCreate(&ev[q])); CK(hipEventRecord(ev[q], st[q])); CK(hipStreamWaitEvent(0, ev[q], 0)); }
          CK(hipEventRecord(e1, 0));
          CK(hipDeviceSynchronize());
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (round) best = std::min(best, ms * 1e3f / reps);
          for (int q = 0; q < ns; ++q) CK(hipEventDestroy(ev[q]));
        }
        us[which][ns - 1] = best;
      }
  const bool bad = diff[0] + diff[1] + diff[2] != 0;
  failures += bad;
  const double fl = 2.0 * M * N * K;
  printf("%-30s %s M %5d N %5d K %5d gate %d bias %d replays %d: words differing cache %zu hidden %zu statistics %zu%s | us per launch on 1 / 2 / 3 streams: four waves %6.1f %6.1f %6.1f (%4.0f TF) | gemm3s %6.1f %6.1f %6.1f (%4.0f TF)\n",
         name, std::is_same<T, bf16_t>::value ? "bf16" : "f16 ", M, N, K, (int)gate, (int)bias, nx, diff[0], diff[1], diff[2], bad ? "  FAIL" : "", us[0][0], us[0][1], us[0][2], us[0][0] > 0 ? fl / us[0][0] / 1e6 : 0.0, us[1][0],
         us[1][1], us[1][2], us[1][0] > 0 ? fl / us[1][0] / 1e6 : 0.0);
  for (auto& q : st) CK(hipStreamDestroy(q));
  CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(MOD)); CK(hipFree(B)); CK(hipFree(TAB)); CK(hipFree(TAB2)); CK(hipFree(XS));
  for (int i = 0; i < 2; ++i) { CK(hipFree(C[i])); CK(hipFree(R[i])); CK(hipFree(ST[i])); CK(hipFree(X[i])); }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 24;
  // small first: a wrong kernel fails here before anything long runs
  rmw_case<f16_t, 1>("one tile row, K = 192", 256, 1152, 192, true, true, 0, 0);
  rmw_case<f16_t, 1>("one tile row, K = 384", 256, 1152, 384, true, true, 0, 0);
  rmw_case<f16_t, 1>("two tile rows, K = 1152", 512, 1152, 1152, true, true, 0, 0);
  rmw_case<f16_t, 1>("attn1-out", 8192, 1152, 1152, true, true, 0, 8 * reps);
  rmw_case<f16_t, 3>("attn2-out, no gate", 8192, 1152, 1152, false, true, 0, 8 * reps);
  rmw_case<f16_t, 3>("no gate, no bias", 8192, 1152, 1152, false, false, 0, 0);
  rmw_case<f16_t, 1>("attn1-out + 1 replay", 8192, 1152, 1152, true, true, 1, 8 * reps);
  rmw_case<f16_t, 1>("attn1-out + 2 replays", 8192, 1152, 1152, true, true, 2, 8 * reps);
  rmw_case<f16_t, 5>("FF2", 8192, 1152, 4608, true, true, 0, 4 * reps);
  rmw_case<f16_t, 5>("FF2 + 2 replays", 8192, 1152, 4608, true, true, 2, 4 * reps);
  rmw_case<bf16_t, 1>("attn1-out", 8192, 1152, 1152, true, true, 1, 8 * reps);
  rmw_case<bf16_t, 5>("FF2", 8192, 1152, 4608, true, true, 0, 4 * reps);
  rmw_case<f16_t, 1>("attn1-out, 1024-token samples", 8192, 1152, 1152, true, true, 1, 0, 1024);
  rmw_case<f16_t, 1>("attn1-out, Sigma rows", 32768, 1152, 1152, true, true, 0, 2 * reps, 4096);
  rmw_case<f16_t, 5>("FF2, Sigma rows", 32768, 1152, 4608, true, true, 0, reps, 4096);
  printf(failures ? "%d cases FAILED\n" : "all cases bit-identical\n", failures);
  return failures ? 1 : 0;
}
