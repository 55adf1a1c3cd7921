// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 block-scale semantics on gfx950: hipcc --offload-arch=gfx950 scripts/probe_mfma_scale.hip -o /tmp/p && /tmp/p
// Result (MI355X): lane l`s scale byte (opsel selects which of the 4) scales row/col l & 15, LOGICAL K-block l >> 4, for both operands -- which registers of which
// lanes make up that block is settled by scripts/probe_mfma_scale2.hip (low 16 bytes of lane group g = K-chunk g, high 16 bytes = chunk 4 + g; block b = chunks 2b, 2b+1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OPSEL>
__global__ void k(const int* sa, const int* sb, f32x4* c) {
  const int ones = 0x38383838;  // four e4m3 1.0
  v8i av = {ones, ones, ones, ones, ones, ones, ones, ones}, bv = av;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, OPSEL, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}
int main() {
  int hsa[64], hsb[64];
  int *dsa, *dsb; f32x4* dc; f32x4 hc[64];
  hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 64 * 16);
  auto run = [&](const char* name, int opsel) {
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    if (opsel == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dsa, dsb, dc); else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dsa, dsb, dc);
    hipMemcpy(hc, dc, 64 * 16, hipMemcpyDeviceToHost);
    printf("%s\n", name);
    for (int l : {0, 1, 5, 16, 17, 37, 63}) printf("  lane %2d: %g %g %g %g\n", l, hc[l][0], hc[l][1], hc[l][2], hc[l][3]);
  };
  for (int l = 0; l < 64; ++l) { hsa[l] = 0x7F7F7F7F; hsb[l] = 0x7F7F7F7F; }
  run("all scales 1 (expect 128)", 0);
  for (int l = 0; l < 64; ++l) hsa[l] = 0x7F7F7F00 | (127 + (l & 3));
  run("first-operand scale byte0 = 127 + (lane & 3)  [lane-own rows => element r = 128 * 2^r]", 0);
  for (int l = 0; l < 64; ++l) hsa[l] = 0x7F7F7F00 | (127 + (l >> 4));
  run("first-operand scale byte0 = 127 + (lane >> 4)  [lane-own k-block => 32 * 15 = 480 everywhere]", 0);
  for (int l = 0; l < 64; ++l) hsa[l] = 0x7F7F7F00 | (127 + (l & 15) % 5);
  run("first-operand scale byte0 = 127 + (lane & 15) % 5", 0);
  for (int l = 0; l < 64; ++l) { hsa[l] = 0x7F7F7F7F; hsb[l] = 0x7F7F7F00 | (127 + (l & 15) % 3); }
  run("second-operand scale byte0 = 127 + (lane & 15) % 3  [column l15 => all 4 elements = 128 * 2^(l15 % 3)]", 0);
  for (int l = 0; l < 64; ++l) { hsb[l] = 0x7F7F7F7F; hsa[l] = 0x7F7F807F; }
  run("first-operand scale bytes {0x7F, 0x80, 0x7F, 0x7F}, opsel 0 (expect 128 if byte 0 is used)", 0);
  run("same, opsel 1 (expect 256 if opsel picks byte 1)", 1);
  return 0;
}
