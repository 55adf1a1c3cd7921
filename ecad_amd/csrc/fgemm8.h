// fgemm8.h -- OCP fp8 (e4m3) variant of fgemm.h for the FLUX GEMMs (BASELINE config 5 names fp8 MFMA; opt-in: the reference computes in
// bf16, so bf16 stays the default and the measured precision of record).
//
//   C[M,N] = epilogue( (A8[M,K] @ W8[N,K]^T) * sa[m] * sw[n] + bias[n] )
//
// A8 / W8 hold x / scale rounded to e4m3; sa is a per-ROW activation scale written by the producer (amax / 448 of the row), sw a
// per-OUTPUT-CHANNEL weight scale fixed at load.  gfx950 only reaches 2x the bf16 rate through the block-scaled instruction
// (v_mfma_scale_f32_16x16x128_f8f6f4, 32 fp8 per lane per operand); it is issued with every block scale = 2^0 and the two fp32
// scales are applied to the accumulators before the shared epilogue, so no E8M0 scale tensors exist and the operand / scale lane
// maps of the instruction beyond "lane l owns row (col) l & 15 and K-group l >> 4" never matter (A and B use the same K order).
// Probed on the hardware (scripts/probe_mfma_scale.hip, probe_mfma_scale2.hip): the scale operands ARE per lane -- the selected byte (opsel) of lane l's scale
// VGPR, an E8M0 exponent -- but the 32 elements it multiplies are the instruction's LOGICAL K-block (l >> 4) of row / column l & 15, and the hardware counts K
// as (register half, lane group): a lane's low 16 bytes are K-chunk g, its high 16 bytes K-chunk 4 + g.  (Round 5 read the first probe as "the lane's own 32
// elements"; with ones everywhere it could not tell.)  A kernel that uses block scales therefore loads pieces g and 4 + g of the 128-byte K-step per lane group,
// not 2g and 2g + 1 -- gemm3s8_kernel<.., BS> does, for the FF1 -> FF2 hand-over of the PixArt path (round 6: e4m3 + one E8M0 byte per (row, 288-column tile)
// from FF1's epilogue, no row quantiser in between).  The same fusion in front of FLUX's FF2 / proj_out is the next step for this mode.
// Same 256 x BN tile, wave layout, LDS image (128-byte rows = 128 fp8, 16-byte chunks XOR-swizzled by (row >> 1) & 7), LDS-DMA ring
// and epilogue as fgemm.h; a K-step is 128 elements, so a step moves the same bytes as a bf16 step but carries twice the flops.
#pragma once
#include "common.h"
#include "fgemm.h"

namespace ecad {

typedef int i32x8 __attribute__((ext_vector_type(8)));
constexpr int GEMM8_BK = 128;
constexpr float FP8_MAX = 448.0f;

template <typename T> struct FGemm8Args {
  const uint8_t* A8;  // [M][K] e4m3, row stride lda8 bytes
  const uint8_t* W8;  // [N][K] e4m3, row stride K
  const float* sa;    // [M]
  const float* sw;    // [N]
  int lda8;
  FGemmArgs<T> e;     // M, N, K (elements), bias, outputs and every epilogue field; A / W / lda / ldw unused
};

// acc * (sa[m] * sw[n]) with exactly these two roundings, whatever -ffast-math would like to reassociate or contract around them: fgemm8_kernel and gemm4.h's
// fgemm48_kernel must hand the shared epilogue the same numbers to stay bit-identical
ECAD_DEV float fp8_descale(float acc, float sa, float sw) {
  float sc = sa * sw;
  asm volatile("" : "+v"(sc));
  float v = acc * sc;
  asm volatile("" : "+v"(v));  // nor may the product be contracted into the epilogue's bias add in one kernel and not in the other
  return v;
}

ECAD_DEV uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
}

// dst8[r][:] = e4m3(cat(src1[r], src2[r]) / s), s = max|row| / 448 (1 for an all-zero row); one wave per row, 16-byte loads.
// Weights at load; activations right before each fp8 GEMM (src2: the single-stream proj_out reads [attn | gelu(mlp)] from two cache
// slots -- the concatenation materialises only in fp8, inside this pass).  K1, K2 multiples of 8.
// NR = 16-byte chunks a lane keeps between the maximum and the conversion (rows of up to 512 NR elements): the buffer is what bounds the waves per SIMD, and with
// them the bytes in flight of a pass that is nothing but latency -- launch_quant_rows_fp8 picks the smallest instantiation that holds the row (round 6: the
// 1152- and 4608-wide rows of the PixArt path ran on the 32-chunk buffer sized for FLUX's 15360, three waves per SIMD).  Same arithmetic for every NR.
template <typename TS, int NR_ = 32>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const TS* __restrict__ src1, int64_t ld1, int K1, const TS* __restrict__ src2, int64_t ld2, int K2,
                                                            uint8_t* __restrict__ dst, int64_t ldd, float* __restrict__ scale, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TS* s1 = src1 + (size_t)row * ld1;
  const TS* s2 = src2 ? src2 + (size_t)row * ld2 : nullptr;
  const int c1 = K1 >> 3, ctot = c1 + (K2 >> 3);
  u32x2* d = reinterpret_cast<u32x2*>(dst + (size_t)row * ldd);
  // Rows of up to 16384 elements (every row of both models: K <= 5 D = 15360) stay in the wave's registers between the maximum and the conversion -- 32 chunks of
  // 16 bytes per lane -- so the row is read once.  The two-pass form below re-read it "from L2", but with every CU holding 32 rows of 30 KB in flight the second
  // pass came from the Infinity Cache / HBM: 5 bytes of traffic per element instead of 3 (round 5: fp8_quant_rows 48 -> 33 ms per FLUX batch).
  constexpr int NR = NR_;
  if (ctot <= 64 * NR) {
    Pack8 buf[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      buf[i].u = (u32x4){0, 0, 0, 0};
      if (64 * i < ctot) {  // wave-uniform
        const int c = lane + 64 * i;
        if (c < ctot) buf[i] = gload16(c < c1 ? s1 + c * 8 : s2 + (c - c1) * 8);
      }
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (64 * i < ctot) {
        float v[8];
        unpack8<TS>(buf[i], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (64 * i < ctot) {
        const int c = lane + 64 * i;
        float v[8];
        unpack8<TS>(buf[i], v);
        if (c < ctot) d[c] = (u32x2){pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv), pack_fp8x4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv)};
      }
    }
    return;
  }
  float amax = 0.f;
  for (int c = lane; c < ctot; c += 64) {
    float v[8];
    unpack8<TS>(gload16(c < c1 ? s1 + c * 8 : s2 + (c - c1) * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  for (int c = lane; c < ctot; c += 64) {  // second pass over the row
    float v[8];
    unpack8<TS>(gload16(c < c1 ? s1 + c * 8 : s2 + (c - c1) * 8), v);
    d[c] = (u32x2){pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv), pack_fp8x4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv)};
  }
}

template <typename TS>
inline void launch_quant_rows_fp8(const TS* src1, int64_t ld1, int K1, const TS* src2, int64_t ld2, int K2, uint8_t* dst, int64_t ldd, float* scale, int rows, hipStream_t s) {
  const int per_lane = (((K1 + K2) >> 3) + 63) / 64;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (per_lane <= 3) hipLaunchKernelGGL((quant_rows_fp8_kernel<TS, 3>), grid, block, 0, s, src1, ld1, K1, src2, ld2, K2, dst, ldd, scale, rows);
  else if (per_lane <= 6) hipLaunchKernelGGL((quant_rows_fp8_kernel<TS, 6>), grid, block, 0, s, src1, ld1, K1, src2, ld2, K2, dst, ldd, scale, rows);
  else if (per_lane <= 9) hipLaunchKernelGGL((quant_rows_fp8_kernel<TS, 9>), grid, block, 0, s, src1, ld1, K1, src2, ld2, K2, dst, ldd, scale, rows);
  else if (per_lane <= 16) hipLaunchKernelGGL((quant_rows_fp8_kernel<TS, 16>), grid, block, 0, s, src1, ld1, K1, src2, ld2, K2, dst, ldd, scale, rows);
  else hipLaunchKernelGGL((quant_rows_fp8_kernel<TS, 32>), grid, block, 0, s, src1, ld1, K1, src2, ld2, K2, dst, ldd, scale, rows);
}

template <typename T, int BN, int EPI>
__global__ __launch_bounds__(512) void fgemm8_kernel(const FGemm8Args<T> q) {
  constexpr int BM = 256, NW = 8, STAGES = fgemm_stages<BN>();
  constexpr int FM = 4, FN = BN / 32;
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
  constexpr int A_INST = BM / 8 / NW, B_INST = BN / 8 / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FGemmArgs<T>& p = q.e;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  int tm, tn;
  fgemm_tile_of(blockIdx.x, (p.M + 255) / 256, p.N / BN, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const uint8_t* a_src[A_INST];
  const uint8_t* b_src[B_INST];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int r = (wave * A_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int row = m0 + r;
    row = row < p.M ? row : p.M - 1;
    a_src[i] = q.A8 + (size_t)row * q.lda8 + c * 16;
  }
#pragma unroll
  for (int i = 0; i < B_INST; ++i) {
    const int r = (wave * B_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[i] = q.W8 + (size_t)(n0 + r) * p.K + c * 16;
  }
  auto issue = [&](int stage) {
    char* abase = smem + stage * STAGE_BYTES;
    char* bbase = abase + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      glds16(a_src[i], abase + (wave * A_INST + i) * 1024);
      a_src[i] += GEMM8_BK;
    }
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
      glds16(b_src[i], bbase + (wave * B_INST + i) * 1024);
      b_src[i] += GEMM8_BK;
    }
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wm * 64 + l15) * 128;
  const int b_row_off = (wn * (BN / 2) + l15) * 128;
  const int c0 = ((2 * g) ^ swz) * 16, c1 = ((2 * g + 1) ^ swz) * 16;  // the two 16-byte chunks holding this lane's 32 K-elements
  const int nk = p.K / GEMM8_BK;
  issue(0);
  if (STAGES == 3 && nk > 1) issue(1);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (STAGES == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INST + B_INST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + STAGES - 1 < nk) {
      int ns = stage + STAGES - 1;
      ns = ns >= STAGES ? ns - STAGES : ns;
      issue(ns);
    }
    const char* abase = smem + stage * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;
    i32x8 af[FM], wf[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const Pack8 lo = lds_read16(abase + a_row_off + i * 2048 + c0), hi = lds_read16(abase + a_row_off + i * 2048 + c1);
      af[i] = (i32x8){(int)lo.u[0], (int)lo.u[1], (int)lo.u[2], (int)lo.u[3], (int)hi.u[0], (int)hi.u[1], (int)hi.u[2], (int)hi.u[3]};
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const Pack8 lo = lds_read16(bbase + b_row_off + j * 2048 + c0), hi = lds_read16(bbase + b_row_off + j * 2048 + c1);
      wf[j] = (i32x8){(int)lo.u[0], (int)lo.u[1], (int)lo.u[2], (int)lo.u[3], (int)hi.u[0], (int)hi.u[1], (int)hi.u[2], (int)hi.u[3]};
    }
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    stage = stage + 1 >= STAGES ? 0 : stage + 1;
  }
  __syncthreads();
  // de-scale: lane (l15, g) holds row l15 of fragment row-block i, columns 4g .. 4g+3 of fragment column-block j
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * 64 + 16 * i + l15;
    const float sam = q.sa[m < p.M ? m : p.M - 1];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const f32x4 sv = *reinterpret_cast<const f32x4*>(q.sw + n0 + wn * (BN / 2) + 16 * j + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = fp8_descale(acc[i][j][r], sam, sv[r]);
    }
  }
  fgemm_epilogue<T, BN, EPI>(p, smem, m0, n0, acc);
}

}  // namespace ecad
