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

// Half-chip launches on several streams: an N = 1152 GEMM at M = 8192 is 128 wide tiles -- one per CU on half the chip.  Does the hardware run two such launches of
// different streams side by side (aggregate microseconds per launch on 1 / 2 / 3 / 4 streams)?  Compared with the same GEMM on 256 narrow tiles (gemm3s.h, one round).
#include "../ecad_amd/csrc/gemm3s.h"
template <int EPI, int TAG> static void streams_case(const char* name, int M, int N, int K, int reps) {
  std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
  for (auto& v : ha) v = f16_bits(rnd());
  for (auto& v : hw) v = f16_bits(rnd() * 0.03f);
  std::vector<float> hb(N);
  for (auto& v : hb) v = rnd();
  const int nsets = 4;
  unsigned short *A[nsets], *W[nsets], *C[nsets]; float* B;
  for (int s = 0; s < nsets; ++s) {
    CK(hipMalloc(&A[s], ha.size() * 2)); CK(hipMalloc(&W[s], hw.size() * 2)); CK(hipMalloc(&C[s], (size_t)M * N * 2));
    CK(hipMemcpy(A[s], ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W[s], hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&B, N * 4)); CK(hipMemcpy(B, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)gemm2w4_kernel<f16_t, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G2W4_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)gemm3s_kernel<f16_t, EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
  auto args = [&](int set) {
    GemmArgs<f16_t> g{};
    g.A = (const f16_t*)A[set]; g.W = (const f16_t*)W[set]; g.bias = B; g.C = (f16_t*)C[set]; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.rows_per_batch = 256; g.epi_direct = 1; g.variant = 1;
    return g;
  };
  hipStream_t st[4];
  for (auto& q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double us[2][4];
  for (int which = 0; which < 2; ++which)
    for (int ns = 1; ns <= 4; ++ns) {
      float best = 1e9f;
      for (int round = 0; round < 3; ++round) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int q = 0; q < ns; ++q) CK(hipStreamWaitEvent(st[q], e0, 0));
        for (int i = 0; i < reps; ++i) {
          const GemmArgs<f16_t> g = args(i % ns);
          if (which == 0) hipLaunchKernelGGL((gemm2w4_kernel<f16_t, EPI, TAG>), dim3((M / 256) * (N / 288)), dim3(256), G2W4_LDS_BYTES, st[i % ns], g);
          else hipLaunchKernelGGL((gemm3s_kernel<f16_t, EPI, TAG>), dim3((M / 256) * (N / 144)), dim3(512), G3_LDS_BYTES, st[i % ns], g);
        }
        hipEvent_t ev[4];
        for (int q = 0; q < ns; ++q) { CK(hipEventCreate(&ev[q])); CK(hipEventRecord(ev[q], st[q])); CK(hipStreamWaitEvent(0, ev[q], 0)); }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (round) best = std::min(best, ms * 1e3f / reps);
        for (int q = 0; q < ns; ++q) CK(hipEventDestroy(ev[q]));
      }
      us[which][ns - 1] = best;
    }
  const double fl = 2.0 * M * N * K;
  printf("%-22s M %5d N %5d K %5d, us per launch on 1 / 2 / 3 / 4 streams: wide tiles on four waves (%d workgroups) %6.1f %6.1f %6.1f %6.1f (%4.0f TF) | gemm3s (%d workgroups) %6.1f %6.1f %6.1f %6.1f (%4.0f TF)\n", name, M, N, K,
         (M / 256) * (N / 288), us[0][0], us[0][1], us[0][2], us[0][3], fl / us[0][3] / 1e6, (M / 256) * (N / 144), us[1][0], us[1][1], us[1][2], us[1][3], fl / us[1][3] / 1e6);
  for (auto& q : st) CK(hipStreamDestroy(q));
  for (int s = 0; s < nsets; ++s) { CK(hipFree(A[s])); CK(hipFree(W[s])); CK(hipFree(C[s])); }
  CK(hipFree(B));
}

// read-modify-write epilogue (cache slot, gated residual, row statistics): gemm2w4 against gemm3s -- three buffers, bit for bit; and the time on 1 / 2 / 3 streams
template <int TAG> static void rmw_case(const char* name, int M, int N, int K, bool gate, int reps) {
  std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K), hr((size_t)M * N), hm((size_t)(M / 256) * 6 * N);
  for (auto& v : ha) v = f16_bits(rnd());
  for (auto& v : hw) v = f16_bits(rnd() * 0.03f);
  for (auto& v : hr) v = f16_bits(rnd());
  for (auto& v : hm) v = f16_bits(rnd());
  std::vector<float> hb(N), ht(N);
  for (auto& v : hb) v = rnd();
  for (auto& v : ht) v = rnd();
  unsigned short *A, *W, *C[2], *R[2], *MOD; float *B, *TAB, *ST[2];
  CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&W, hw.size() * 2)); CK(hipMalloc(&MOD, hm.size() * 2)); CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&TAB, N * 4));
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(MOD, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(TAB, ht.data(), N * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&C[i], (size_t)M * N * 2)); CK(hipMalloc(&R[i], (size_t)M * N * 2)); CK(hipMalloc(&ST[i], (size_t)(N / 144) * M * 2 * 4)); }
  CK(hipFuncSetAttribute((const void*)gemm2w4_kernel<f16_t, EPI_BIAS_CACHE_RESID, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G2W4_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)gemm3s_kernel<f16_t, EPI_BIAS_CACHE_RESID, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES));
  auto args = [&](int which) {
    GemmArgs<f16_t> g{};
    g.A = (const f16_t*)A; g.W = (const f16_t*)W; g.bias = B; g.C = (f16_t*)C[which]; g.resid = (f16_t*)R[which]; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N;
    g.rows_per_batch = 256; g.variant = 1; g.epi_direct = 1; g.stat_out = ST[which];
    if (gate) { g.gate_table = TAB; g.gate_mod = (const f16_t*)MOD + 2 * N; g.gate_mod_stride = 6 * N; }
    return g;
  };
  const int wide = (M / 256) * (N / 288), narrow = (M / 256) * (N / 144);
  const int cus = 256;
  auto launch = [&](int which, hipStream_t st) {
    const GemmArgs<f16_t> g = args(which);
    if (which == 0) hipLaunchKernelGGL((gemm2w4_kernel<f16_t, EPI_BIAS_CACHE_RESID, TAG>), dim3(wide > cus && wide % cus == 0 ? cus : wide), dim3(256), G2W4_LDS_BYTES, st, g);
    else hipLaunchKernelGGL((gemm3s_kernel<f16_t, EPI_BIAS_CACHE_RESID, TAG>), dim3(narrow), dim3(512), G3_LDS_BYTES, st, g);
  };
  for (int i = 0; i < 2; ++i) { CK(hipMemcpy(R[i], hr.data(), hr.size() * 2, hipMemcpyHostToDevice)); CK(hipMemset(C[i], 0xFF, (size_t)M * N * 2)); CK(hipMemset(ST[i], 0xFF, (size_t)(N / 144) * M * 8)); launch(i, 0); }
  CK(hipDeviceSynchronize());
  size_t diff[3] = {0, 0, 0};
  {
    std::vector<unsigned short> x((size_t)M * N), y((size_t)M * N);
    CK(hipMemcpy(x.data(), C[0], x.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), C[1], y.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < x.size(); ++i) {
      if (x[i] != y[i] && diff[0] < 6) { _Float16 a, b; std::memcpy(&a, &x[i], 2); std::memcpy(&b, &y[i], 2); printf("      cache[%zu, %zu]: four waves %g, gemm3s %g\n", i / N, i % N, (double)a, (double)b); }
      diff[0] += x[i] != y[i];
    }
    {
      size_t hist[4][2] = {};
      for (size_t i = 0; i < x.size(); ++i) if (x[i] != y[i]) hist[((i / N) % 256) / 64][((i % N) % 288) / 144]++;
      printf("      differing by (row quarter of the tile, column half): %zu %zu | %zu %zu | %zu %zu | %zu %zu\n", hist[0][0], hist[0][1], hist[1][0], hist[1][1], hist[2][0], hist[2][1], hist[3][0], hist[3][1]);
      double ref[4];
      for (int n = 0; n < 4; ++n) { double a = hb[n]; for (int k = 0; k < K; ++k) { _Float16 u, w; std::memcpy(&u, &ha[k], 2); std::memcpy(&w, &hw[(size_t)n * K + k], 2); a += (double)u * (double)w; } ref[n] = a; }
      printf("      fp64 reference cache[0, 0..3]: %g %g %g %g\n", ref[0], ref[1], ref[2], ref[3]);
    }
    CK(hipMemcpy(x.data(), R[0], x.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), R[1], y.size() * 2, hipMemcpyDeviceToHost));
    {
      std::vector<unsigned short> cc((size_t)M * N);
      CK(hipMemcpy(cc.data(), C[1], cc.size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < x.size(); ++i) {
        if (x[i] != y[i] && diff[1] < 4) {
          auto f = [](unsigned short u) { _Float16 a; std::memcpy(&a, &u, 2); return (double)a; };
          const size_t m = i / N, n = i % N;
          const float h0 = (float)f(hr[i]), o = (float)f(cc[i]), t = ht[n], md = (float)f(hm[(m / 256) * 6 * N + 2 * N + n]);
          const float g = t + md;
          volatile float prod = g * o;
          const float c_fma = std::fmaf(g, o, h0), c_mul = prod + h0, c_two = std::fmaf(md, o, std::fmaf(t, o, h0)), c_two2 = std::fmaf(t, o, std::fmaf(md, o, h0));
          printf("      hidden[%zu, %zu]: four waves %.9g gemm3s %.9g | fp32: fma %.9g  mul+add %.9g  two fmas %.9g / %.9g | h0 %.9g o %.9g table %.9g mod %.9g g %.9g\n", m, n, f(x[i]), f(y[i]), (double)c_fma, (double)c_mul,
                 (double)c_two, (double)c_two2, (double)h0, (double)o, (double)t, (double)md, (double)g);
        }
        diff[1] += x[i] != y[i];
      }
    }
    std::vector<unsigned> sa((size_t)(N / 144) * M * 2), sb(sa.size());
    CK(hipMemcpy(sa.data(), ST[0], sa.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), ST[1], sb.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < sa.size(); ++i) diff[2] += sa[i] != sb[i];
  }
#ifdef G2W4_STAMP
  {
    for (int r = 0; r < 100; ++r) launch(0, 0);
    CK(hipDeviceSynchronize());
    const int nb = wide > cus && wide % cus == 0 ? cus : wide;
    std::vector<unsigned long long> sv(4 * nb);
    CK(hipMemcpyFromSymbol(sv.data(), HIP_SYMBOL(g2w4_stamp), sv.size() * 8));
    std::vector<double> a, b, c, f;
    for (int i = 0; i < nb; ++i) { a.push_back((double)sv[4 * i]); b.push_back((double)sv[4 * i + 1]); c.push_back((double)sv[4 * i + 2]); f.push_back((double)sv[4 * i + 2] / (double)sv[4 * i + 3] * 0.1); }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const int per = (wide + nb - 1) / nb;
    printf("    stamps (%d tiles per workgroup): K loop + C-tile staging %.0f cycles/tile, row pass %.0f cycles/tile, workgroup %.0f cycles, clock %.2f GHz\n", per, med(a) / per, med(b) / per, med(c), med(f));
  }
#endif
  hipStream_t st[3];
  for (auto& q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double us[2][3];
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
  printf("%-26s M %5d N %5d K %5d gate %d: words differing cache %zu hidden %zu statistics %zu%s | us per launch on 1 / 2 / 3 streams: four waves %6.1f %6.1f %6.1f | gemm3s %6.1f %6.1f %6.1f\n", name, M, N, K, (int)gate,
         diff[0], diff[1], diff[2], diff[0] + diff[1] + diff[2] ? "  FAIL" : "", us[0][0], us[0][1], us[0][2], us[1][0], us[1][1], us[1][2]);
  for (auto& q : st) CK(hipStreamDestroy(q));
  CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(MOD)); CK(hipFree(B)); CK(hipFree(TAB));
  for (int i = 0; i < 2; ++i) { CK(hipFree(C[i])); CK(hipFree(R[i])); CK(hipFree(ST[i])); }
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
  rmw_case<1>("rmw attn1-out", 8192, 1152, 1152, true, 8 * reps);
  rmw_case<3>("rmw attn2-out, no gate", 8192, 1152, 1152, false, 8 * reps);
  rmw_case<5>("rmw FF2", 8192, 1152, 4608, true, 4 * reps);
  rmw_case<1>("rmw attn1-out, Sigma rows", 32768, 1152, 1152, true, 2 * reps);
  rmw_case<5>("rmw FF2, Sigma rows", 32768, 1152, 4608, true, reps);
  streams_case<EPI_BIAS, 0>("plain N=K=1152", 8192, 1152, 1152, 8 * reps);
  streams_case<EPI_BIAS, 5>("plain N=1152 K=4608", 8192, 1152, 4608, 4 * reps);
  streams_case<EPI_BIAS, 0>("plain, Sigma rows", 32768, 1152, 1152, 2 * reps);
  streams_case<EPI_BIAS, 5>("K=4608, Sigma rows", 32768, 1152, 4608, reps);
  return 0;
}
