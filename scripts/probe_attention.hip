// probe_attention.hip -- stand-alone timing of the attention kernels (attention_dma.h / scripts/retired_attention_pingpong.h) on the FLUX joint shape (B 2, 24 heads x 128,
// S 4608, bf16) and the Sigma-1024 self-attention shape (Bt 8, 16 heads x 72, N 4096, fp16), with the ablation macros of attention_dma.h.
// build (one binary per ablation): hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only -DECAD_DIAG -DATT_ABL=<bits> scripts/probe_attention.hip -o build_tmp/probe_attention_<bits>
// run on the GPU box: build_tmp/probe_attention_<bits> [reps]; prints microseconds per launch (min of 3 timed loops) for every kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../ecad_amd/csrc/attention_dma.h"
#if __has_include("retired_attention_pingpong.h")
#include "retired_attention_pingpong.h"
#define HAVE_PINGPONG 1
#endif
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> static float run(const char* name, const void* fn, int lds, int B, int H, int N, int dh, int rows_per_item, int reps, int threads = 512) {
  const size_t D = (size_t)H * dh, rows = (size_t)B * N;
  std::vector<unsigned short> h(rows * 3 * D);
  unsigned x = 12345u;
  for (auto& v : h) {  // values in (-1, 1): sign + exponent below 1.0 + random mantissa bits, in either 16-bit format
    x = x * 1664525u + 1013904223u;
    const float f = ((int)(x >> 8) % 2001 - 1000) * 0.001f;
    if (sizeof(T) == 2 && std::is_same<T, f16_t>::value) { _Float16 hf = (_Float16)f; v = *reinterpret_cast<unsigned short*>(&hf); }
    else { unsigned u; std::memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  }
  unsigned short *qkv, *out;
  CK(hipMalloc(&qkv, h.size() * 2)); CK(hipMalloc(&out, rows * D * 2));
  CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  AttnArgs a{};
  a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.out = out; a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
  a.B = B; a.H = H; a.Nq = N; a.Nk = N; a.dh = dh; a.scale_log2e = 1.4426950408889634f / sqrtf((float)dh);
  unsigned long long* dbg;
  CK(hipMalloc(&dbg, 128)); CK(hipMemset(dbg, 0, 128));
#if ATT_STAMP
  a.bias = reinterpret_cast<const float*>(dbg);
#endif
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int items = B * H * ((N + rows_per_item - 1) / rows_per_item);
  int cus = 256;
  void* params[] = {&a};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) CK(hipLaunchKernel(fn, dim3(items < cus ? items : cus), dim3(threads), params, lds, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it) best = ms / reps < best ? ms / reps : best;
  }
  const double flops = 4.0 * B * H * (double)N * N * dh;
  printf("ABL %3d  %-34s %8.1f us  %6.0f TF\n", ATT_ABL, name, best * 1e3, flops / (best * 1e-3) / 1e12);
#if ATT_STAMP
  unsigned long long hd[16];
  CK(hipMemcpy(hd, dbg, 128, hipMemcpyDeviceToHost));
  for (int gq = 0; gq < 2; ++gq)
    if (hd[8 + 4 * gq + 3]) printf("         ping-pong, wave %d, cycles per 64-key unit: M phase %.0f  V phase %.0f  waiting at the barriers %.0f\n", 4 * gq,
                                  (double)hd[8 + 4 * gq] / hd[8 + 4 * gq + 3], (double)hd[8 + 4 * gq + 1] / hd[8 + 4 * gq + 3], (double)hd[8 + 4 * gq + 2] / hd[8 + 4 * gq + 3]);
  if (hd[5]) printf("         cycles per 64-key unit (wave 0 of workgroup 0, first item): wait+barrier %.0f  dma issue %.0f  S %.0f  max+rescale %.0f  exp+PV %.0f  (units %llu)\n",
                    (double)hd[0] / hd[5], (double)hd[1] / hd[5], (double)hd[2] / hd[5], (double)hd[3] / hd[5], (double)hd[4] / hd[5], hd[5]);
#endif
  CK(hipFree(qkv)); CK(hipFree(out)); CK(hipFree(dbg));
  return best;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  run<bf16_t>("flux  dma  d128 R2 (B2 H24 S4608)", (const void*)attn_dma_kernel<bf16_t, 128, 2>, attn_dma_lds_bytes<128>(), 2, 24, 4608, 128, 256, reps);
  run<f16_t>("sigma dma  d72  R2 (B8 H16 N4096)", (const void*)attn_dma_kernel<f16_t, 72, 2>, attn_dma_lds_bytes<72>(), 8, 16, 4096, 72, 256, reps);
  run<f16_t>("sigma dma  d72  R4", (const void*)attn_dma_kernel<f16_t, 72, 4>, attn_dma_lds_bytes<72>(), 8, 16, 4096, 72, 512, reps);
#ifdef ATT_TRY_NW12
  run<f16_t>("sigma dma  d72  R2, 12 waves", (const void*)attn_dma_kernel<f16_t, 72, 2, 12>, attn_dma_lds_bytes<72>(), 8, 16, 4096, 72, 384, reps, 768);
  run<f16_t>("sigma dma  d72  R2, 16 waves", (const void*)attn_dma_kernel<f16_t, 72, 2, 16>, attn_dma_lds_bytes<72>(), 8, 16, 4096, 72, 512, reps, 1024);
  run<f16_t>("sigma dma  d72  R1, 16 waves", (const void*)attn_dma_kernel<f16_t, 72, 1, 16>, attn_dma_lds_bytes<72>(), 8, 16, 4096, 72, 256, reps, 1024);
  run<bf16_t>("flux  dma  d128 R1, 16 waves", (const void*)attn_dma_kernel<bf16_t, 128, 1, 16>, attn_dma_lds_bytes<128>(), 2, 24, 4608, 128, 256, reps, 1024);
#endif
#ifdef ATT_TRY_NW4
  run<f16_t>("sigma dma  d72  R8, 4 waves", (const void*)attn_dma_kernel<f16_t, 72, 8, 4>, attn_dma_lds_bytes<72>(), 8, 16, 4096, 72, 512, reps, 256);
  run<f16_t>("sigma dma  d72  R4, 4 waves", (const void*)attn_dma_kernel<f16_t, 72, 4, 4>, attn_dma_lds_bytes<72>(), 8, 16, 4096, 72, 256, reps, 256);
  run<bf16_t>("flux  dma  d128 R4, 4 waves", (const void*)attn_dma_kernel<bf16_t, 128, 4, 4>, attn_dma_lds_bytes<128>(), 2, 24, 4608, 128, 256, reps, 256);
#endif
#if (ATT_ABL == 0 || ATT_STAMP) && defined(HAVE_PINGPONG)
  run<bf16_t>("flux  pipe d128 R2", (const void*)attn_pipe_kernel<bf16_t, 128, 2>, attn_pipe_lds_bytes<128>(), 2, 24, 4608, 128, 256, reps);
  run<f16_t>("sigma pipe d72  R2", (const void*)attn_pipe_kernel<f16_t, 72, 2>, attn_pipe_lds_bytes<72>(), 8, 16, 4096, 72, 256, reps);
#endif
  return 0;
}
