// probe_mfma_scale2.hip -- which lane's E8M0 byte scales which 32-element block of the SECOND operand of v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950)?  The second
// operand is ones in exactly one lane group g0 (lanes 16 g0 .. 16 g0 + 15) and one register half (bytes 0-15 or 16-31 of the lane's 32), zero elsewhere; the first
// operand is all ones with unit scales; lane l of the second operand's scale VGPR holds byte0 = 127 + (l >> 4).  Result / 16 = 2^(group whose scale was applied).
// build: hipcc --offload-arch=gfx950 scripts/probe_mfma_scale2.hip -o build_tmp/probe_mfma_scale2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* sa, const int* sb, int g0, int half, int which, f32x4* c) {
  const int ones = 0x38383838;  // four e4m3 1.0
  const int lane = threadIdx.x;
  v8i full = {ones, ones, ones, ones, ones, ones, ones, ones};
  v8i part = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((lane >> 4) == g0) {
    if (half == 0) part = (v8i){ones, ones, ones, ones, 0, 0, 0, 0};
    else part = (v8i){0, 0, 0, 0, ones, ones, ones, ones};
  }
  f32x4 acc = {0, 0, 0, 0};
  if (which == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(full, part, acc, 0, 0, 0, sa[lane], 0, sb[lane]);   // sparse SECOND operand, its scales vary
  else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(part, full, acc, 0, 0, 0, sb[lane], 0, sa[lane]);            // sparse FIRST operand, its scales vary
  c[lane] = acc;
}
int main() {
  int hsa[64], hsb[64];
  int *dsa, *dsb; f32x4* dc; f32x4 hc[64];
  hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 64 * 16);
  for (int l = 0; l < 64; ++l) { hsa[l] = 0x7F7F7F7F; hsb[l] = 0x7F7F7F00 | (127 + (l >> 4)); }
  hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
  for (int which = 1; which <= 2; ++which)
    for (int g0 = 0; g0 < 4; ++g0)
      for (int half = 0; half < 2; ++half) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dsa, dsb, g0, half, which, dc);
        hipMemcpy(hc, dc, 64 * 16, hipMemcpyDeviceToHost);
        printf("%s operand sparse: ones in lane group %d, bytes %2d-%2d of the lane: D[0][0] = %g (= 16 x 2^%g)  lane 17 elem 0 = %g\n", which == 1 ? "second" : "first ", g0, 16 * half, 16 * half + 15, hc[0][0],
               log2(hc[0][0] / 16.0), hc[17][0]);
      }
  // and with the scale varying by (lane & 15) for the second operand: column dependence
  for (int l = 0; l < 64; ++l) hsb[l] = 0x7F7F7F00 | (127 + ((l & 15) == 3 ? 2 : 0));
  hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dsa, dsb, 1, 0, 1, dc);
  hipMemcpy(hc, dc, 64 * 16, hipMemcpyDeviceToHost);
  printf("second operand, scale 2^2 only in lanes with (lane & 15) == 3, ones in group 1: lane 3 elems %g %g %g %g | lane 2 elems %g | lane 19 elems %g\n", hc[3][0], hc[3][1], hc[3][2], hc[3][3], hc[2][0], hc[19][0]);
  return 0;
}
