// probe_conv4.hip -- time of conv4_kernel (ecad_amd/csrc/conv4.h) on the decoder's shapes, and of its phases: build with -DCONV4_PROBE_NO_EPILOGUE (K loop + prologue only)
// or -DCONV4_PROBE_NO_LOOP (prologue + epilogue only; the results are garbage either way).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffast-math -fno-finite-math-only scripts/probe_conv4.hip -o build_tmp/probe_conv4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ecad_amd/csrc/conv4.h"
using namespace ecad;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static int g_tile2d = 0, g_korder = 0;
template <int BN, bool RES> static void run(const char* name, int B, int H, int W, int cin, int cout, int taps, int up, bool stats) {
  const int Hs = up == 2 ? H / 2 : H, Ws = up == 2 ? W / 2 : W;
  const size_t in_elems = taps == 9 ? (size_t)B * (Hs + 2) * (Ws + 2) * cin : (size_t)B * H * W * cin;
  const int K = (up == 2 ? 4 : taps) * cin;
  const size_t w_elems = (size_t)(up == 2 ? 4 : 1) * cout * K, out_elems = (size_t)B * H * W * cout;
  f16_t *in, *w, *out, *res; float *bias, *part;
  CK(hipMalloc(&in, in_elems * 2)); CK(hipMalloc(&w, w_elems * 2)); CK(hipMalloc(&out, out_elems * 2)); CK(hipMalloc(&res, out_elems * 2));
  CK(hipMalloc(&bias, cout * 4)); CK(hipMalloc(&part, (size_t)B * H * W / 256 * 64 * 4 * 4));
  CK(hipMemset(in, 0x11, in_elems * 2)); CK(hipMemset(w, 0x12, w_elems * 2)); CK(hipMemset(res, 0x13, out_elems * 2)); CK(hipMemset(bias, 0, cout * 4));
  ConvArgs a{};
  a.in = in; a.w = w; a.bias = bias; a.resid = RES ? res : nullptr; a.out = out; a.B = B; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout; a.taps = taps; a.up = up;
  a.gn_part = stats ? part : nullptr;
  a.tile2d = g_tile2d;
  a.korder = g_korder;
  CK(hipFuncSetAttribute((const void*)conv4_kernel<f16_t, BN, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, conv4_lds_bytes<BN>()));
  const int M = B * Hs * Ws;
  const dim3 grid((M / 256) * (cout / BN) * (up == 2 ? 4 : 1));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv4_kernel<f16_t, BN, RES>), grid, dim3(256), conv4_lds_bytes<BN>(), 0, a);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((conv4_kernel<f16_t, BN, RES>), grid, dim3(256), conv4_lds_bytes<BN>(), 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, fl = 2.0 * B * H * W * (double)cout * K;
  printf("%-44s %8.1f us  %7.0f TFLOP/s   (%u x %u workgroups, %.1f us per tile and CU)\n", name, us, fl / us / 1e6, grid.x, grid.y, us / (grid.x * grid.y / 256.0));
  CK(hipFree(in)); CK(hipFree(w)); CK(hipFree(out)); CK(hipFree(res)); CK(hipFree(bias)); CK(hipFree(part));
}

int main(int argc, char** argv) {
  g_tile2d = argc > 1 ? atoi(argv[1]) : 0;
  g_korder = argc > 2 ? atoi(argv[2]) : 0;
  printf("tile2d %d, K order %s\n", g_tile2d, g_korder ? "kernel row, tap, channel (conv3s)" : "chunk of 64 channels, kernel row, tap");
  run<128, false>("256^2 128->128 (BN 128)", 16, 256, 256, 128, 128, 9, 0, true);
  run<128, true>("256^2 128->128 + residual", 16, 256, 256, 128, 128, 9, 0, true);
  run<128, false>("256^2 256->128", 16, 256, 256, 256, 128, 9, 0, true);
  run<256, false>("128^2 256->256 (BN 256)", 16, 128, 128, 256, 256, 9, 0, true);
  run<256, true>("128^2 256->256 + residual", 16, 128, 128, 256, 256, 9, 0, true);
  run<256, false>("128^2 512->256", 16, 128, 128, 512, 256, 9, 0, true);
  run<256, false>("64^2 512->512", 16, 64, 64, 512, 512, 9, 0, true);
  run<128, false>("32^2 512->512 (BN 128)", 16, 32, 32, 512, 512, 9, 0, true);
  run<256, false>("256^2 upsampler 256 (four phases)", 16, 256, 256, 256, 256, 9, 2, true);
  run<256, false>("128^2 upsampler 512 (four phases)", 16, 128, 128, 512, 512, 9, 2, true);
  return 0;
}
