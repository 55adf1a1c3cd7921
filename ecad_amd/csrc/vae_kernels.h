// vae_kernels.h -- kernels of the AutoencoderKL decoder tail (VAE decode + postprocess, inside the
// reference's timed region at ecad/pipelines/pass_through.py:382-396 / pixart_image_generator.py:405-438).
// Activations are NHWC fp16/bf16 so a 3x3 convolution is an implicit GEMM with K-contiguous operands:
//   out[(b,y,x)][co] = sum_{tap, ci} in[(b, y+dy, x+dx)][ci] * w[co][tap][ci]
// and the A tile of one K-step (64 channels of one tap) is gathered by LDS-DMA with a per-lane source address; rows
// that fall into the zero padding point at a zero page.  Nearest-2x upsampling in front of a conv is folded into the
// address generator ((y+dy)>>1, (x+dx)>>1).
#pragma once
#include "common.h"

namespace ecad {

struct ConvArgs {
  const void* in;     // [B, Hin, Win, Cin]
  const void* w;      // [Cout][taps][Cin]
  const float* bias;  // [Cout]
  const void* resid;  // [B, H, W, Cout] or null: out = conv + bias + resid
  void* out;          // [B, H, W, Cout]
  const void* zero_page;  // >= 128 bytes of zeros
  int B, H, W;        // OUTPUT spatial size
  int Cin, Cout, taps;  // taps = 9 (3x3, pad 1) or 1 (1x1 / linear)
  int up;             // 1: input is [B, H/2, W/2, Cin], nearest-upsampled on the fly
  int korder;         // conv4_kernel only: 0 = K runs chunk of 64 channels, kernel row, tap (p.w laid out [Cout][Cin / 64][taps][64]); 1 = kernel row, tap, channel (p.w [Cout][taps][Cin])
  int tile2d;         // conv4_kernel only: 16 x 16 pixel patches as M tiles (needs the M image's H % 16 == W % 16 == 0), see conv4.h
  float* gn_part;     // conv4_kernel only: per-tile GroupNorm partial sums of the OUTPUT, [B H W / 256][32 groups][2] (null: none); needs H W % 256 == 0
  size_t bs_in, bs_w, bs_out;  // conv_gemm_kernel only: element strides of in / w / out per blockIdx.y (batched GEMMs of the mid-block attention; 0 elsewhere)
};

// 128 x 128 x 64 tile, 4 waves (2 x 2, 64 x 64 each), 2-stage LDS ring; same LDS image / swizzle / epilogue as gemm.h.
template <typename T>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvArgs p) {
  constexpr int BM = 128, BN = 128, BK = 64, FM = 4, FN = 4;
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
  constexpr int A_INST = 4, B_INST = 4, CS = BN + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.B * p.H * p.W;
  const int tiles_n = p.Cout / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = p.taps * p.Cin;
  const int Hin = p.up ? p.H >> 1 : p.H, Win = p.up ? p.W >> 1 : p.W;
  const T* in = reinterpret_cast<const T*>(p.in) + blockIdx.y * p.bs_in;
  const T* wgt = reinterpret_cast<const T*>(p.w) + blockIdx.y * p.bs_w;

  // this lane's A rows (output pixels) and its 16-byte chunk inside the 64-channel K-step
  int pb[A_INST], py[A_INST], px[A_INST], pc[A_INST];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int r = (wave * A_INST + i) * 8 + (lane >> 3);
    int m = m0 + r;
    m = m < M ? m : M - 1;
    pb[i] = m / (p.H * p.W);
    const int rem = m - pb[i] * p.H * p.W;
    py[i] = rem / p.W;
    px[i] = rem - py[i] * p.W;
    pc[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
  const T* b_src[B_INST];
#pragma unroll
  for (int i = 0; i < B_INST; ++i) {
    const int r = (wave * B_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[i] = wgt + (size_t)(n0 + r) * K + c * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* abase = smem + buf * STAGE_BYTES;
    char* bbase = abase + A_BYTES;
    const int k0 = kt * BK;
    const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
    const int dy = p.taps == 9 ? tap / 3 - 1 : 0, dx = p.taps == 9 ? tap % 3 - 1 : 0;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      const int sy = p.up ? yy >> 1 : yy, sx = p.up ? xx >> 1 : xx;
      const T* src = ok ? in + ((size_t)(pb[i] * Hin + sy) * Win + sx) * p.Cin + c0 + pc[i] : reinterpret_cast<const T*>(p.zero_page) + (pc[i] & 56);
      glds16(src, abase + (wave * A_INST + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
      glds16(b_src[i], bbase + (wave * B_INST + i) * 1024);
      b_src[i] += BK;
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wm * 64 + l15) * 128, b_row_off = (wn * 64 + l15) * 128;
  const int nk = K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
    const char* abase = smem + buf * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + g) ^ swz) * 16;
      Pack8 af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = lds_read16(abase + a_row_off + i * 2048 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = lds_read16(bbase + b_row_off + j * 2048 + coff);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<T>(wf[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // epilogue: bias in fp32, round, C tile through LDS, 16-byte row stores (+ residual)
  T* ctile = reinterpret_cast<T*>(smem);
  f32x4 bvs[FN];  // one batch of loads (gemm3s.h: hipcc otherwise serializes a load -> wait -> use round trip per column group)
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    bvs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bvs[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 64 + 16 * j + 4 * g);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = wn * 64 + 16 * j + 4 * g;
    const f32x4 bv = bvs[j];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ml = wm * 64 + 16 * i + l15;
      Pack4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) float_to_elem4<T>(o, r, acc[i][j][r] + bv[r]);
      *reinterpret_cast<u32x2*>(ctile + ml * CS + nl) = o.u;
    }
  }
  __syncthreads();
  T* out = reinterpret_cast<T*>(p.out) + blockIdx.y * p.bs_out;
  const T* res = reinterpret_cast<const T*>(p.resid);
  const int cchunk = tid & 15, crow0 = tid >> 4;
#pragma unroll 4
  for (int pass = 0; pass < 8; ++pass) {
    const int ml = pass * 16 + crow0;
    const int m = m0 + ml;
    if (m >= M) break;
    Pack8 v = lds_read16(ctile + ml * CS + cchunk * 8);
    const size_t off = (size_t)m * p.Cout + n0 + cchunk * 8;
    if (res) {
      float a[8], b[8];
      unpack8<T>(v, a);
      unpack8<T>(gload16(res + off), b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
      v = pack8<T>(a);
    }
    gstore16(out + off, v);
  }
}

// conv3s_kernel -- the K loop of gemm3s.h on the implicit-GEMM convolution (round 3): 256 x 128 x 64 tiles, eight waves of 64 x 64 (4 x 2), a 3-stage LDS-DMA
// ring with counted vmcnt (a stage has two K-steps to land), the second fragment set of a step consumed after the next step's barrier, and the six DMA pieces
// of a wave issued one per pair of MFMAs -- waves 0-3 beside the carried set, waves 4-7 (their SIMD partners) beside the step's first set -- instead of in
// one block after the barrier.  The A rows are gathered as in conv_gemm_kernel (per-lane source address, zero page for the padding, nearest-2x upsample in
// the address); a piece's source advances by 64 channels per K-step and is recomputed when the step enters the next tap.  Same accumulation order per output
// as conv_gemm_kernel (K ascending, fp32), same epilogue arithmetic: the two kernels give identical outputs.  The engine launches it when the pixel count
// fills 256-row tiles (env ECAD_VAE_CONV=tile128 keeps the 128 x 128 kernel everywhere).
constexpr int C3_BM = 256, C3_BN = 128, C3_NW = 8, C3_STAGES = 3;
constexpr int C3_STAGE_BYTES = (C3_BM + C3_BN) * 128;        // 49 152
constexpr int C3_LDS_BYTES = C3_STAGES * C3_STAGE_BYTES;     // 147 456 (the C tile, 256 x 136 x 2 = 69 632, reuses the ring)

template <typename T>
__global__ __launch_bounds__(C3_NW * 64, 2) void conv3s_kernel(const ConvArgs p) {
  constexpr int BM = C3_BM, BN = C3_BN, BK = 64, FM = 4, FN = 4, NSTG = C3_STAGES;
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = C3_STAGE_BYTES;
  constexpr int A_INST = BM / 8 / C3_NW, B_INST = BN / 8 / C3_NW;  // 4 + 2 one-KiB pieces per wave and stage
  constexpr int VMW = A_INST + B_INST, CS = BN + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.B * p.H * p.W;
  const int tiles_n = p.Cout / BN;
  const int tiles_m = (M + BM - 1) / BM;
  int bid = blockIdx.x;
  {  // workgroups of one XCD (blockIdx % 8) take a contiguous run of tiles: neighbours in n share their A rows in that XCD's L2
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = p.taps * p.Cin;
  const int Hin = p.up ? p.H >> 1 : p.H, Win = p.up ? p.W >> 1 : p.W;
  const T* in = reinterpret_cast<const T*>(p.in);
  const T* wgt = reinterpret_cast<const T*>(p.w);

  int pb[A_INST], py[A_INST], px[A_INST], pc[A_INST];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int r = (wave * A_INST + i) * 8 + (lane >> 3);
    int m = m0 + r;
    m = m < M ? m : M - 1;
    pb[i] = m / (p.H * p.W);
    const int rem = m - pb[i] * p.H * p.W;
    py[i] = rem / p.W;
    px[i] = rem - py[i] * p.W;
    pc[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
  const T* a_cur[A_INST];
  int a_inc[A_INST];  // elements per K-step: 64 inside the image, 0 on the zero page
  const T* b_src[B_INST];
#pragma unroll
  for (int i = 0; i < B_INST; ++i) {
    const int r = (wave * B_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[i] = wgt + (size_t)(n0 + r) * K + c * 8;
  }
  int tap = 0, c0 = 0;  // of the next stage to issue
  auto set_tap = [&]() {
    const int dy = p.taps == 9 ? tap / 3 - 1 : 0, dx = p.taps == 9 ? tap - (tap / 3) * 3 - 1 : 0;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      const int sy = p.up ? yy >> 1 : yy, sx = p.up ? xx >> 1 : xx;
      a_cur[i] = ok ? in + ((size_t)(pb[i] * Hin + sy) * Win + sx) * p.Cin + pc[i] : reinterpret_cast<const T*>(p.zero_page) + (pc[i] & 56);
      a_inc[i] = ok ? BK : 0;
    }
  };
  auto advance = [&]() {  // the sources of the stage after the one just issued
    c0 += BK;
    if (c0 == p.Cin) {
      c0 = 0;
      ++tap;
      if (tap < p.taps) set_tap();
    } else {
#pragma unroll
      for (int i = 0; i < A_INST; ++i) a_cur[i] += a_inc[i];
    }
#pragma unroll
    for (int i = 0; i < B_INST; ++i) b_src[i] += BK;
  };
  auto issue_one = [&](int stg, auto qc) {
    constexpr int q = decltype(qc)::value;
    char* abase = smem + stg * STAGE_BYTES;
    if constexpr (q < A_INST) glds16(a_cur[q], abase + (wave * A_INST + q) * 1024);
    else glds16(b_src[q - A_INST], abase + A_BYTES + (wave * B_INST + (q - A_INST)) * 1024);
  };
  auto issue_all = [&](int stg) { static_for<VMW>([&](auto qc) { issue_one(stg, qc); }); };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wm * 64 + l15) * 128, b_row_off = (wn * 64 + l15) * 128;
  const int nk = K / BK;
  set_tap();
  issue_all(0);
  advance();
  if (nk > 1) { issue_all(1); advance(); }

  f32x4 bvs[FN];
  auto load_bias = [&]() {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      bvs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias) bvs[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 64 + 16 * j + 4 * g);
    }
  };
  int stage = 0;
  Pack8 afp[FM], wfp[FN];  // second fragment set of a step: multiplied after the NEXT step's barrier
  auto step_top = [&](int kt) {
    if (kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");  // only the newest stage's pieces of this wave may still be in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto read_set = [&](Pack8 (&af)[FM], Pack8 (&wf)[FN], int stg, int half) {
    const char* abase = smem + stg * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;
    const int coff = ((half * 4 + g) ^ swz) * 16;
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = lds_read16(abase + a_row_off + i * 2048 + coff);
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[j] = lds_read16(bbase + b_row_off + j * 2048 + coff);
  };
  auto mfma_all = [&](const Pack8 (&wf)[FN], const Pack8 (&af)[FM]) {
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf[j], af[i], acc[i][j]);
  };
  auto next_stage = [&]() { int ns = stage + NSTG - 1; return ns >= NSTG ? ns - NSTG : ns; };
  auto step_block = [&](int kt) {
    step_top(kt);
    Pack8 af0[FM], wf0[FN];
    read_set(af0, wf0, stage, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NSTG - 1 < nk) { issue_all(next_stage()); advance(); }
    if (kt == nk - 2) load_bias();
    __builtin_amdgcn_sched_barrier(0);
    if (kt > 0) mfma_all(wfp, afp);
    __builtin_amdgcn_sched_barrier(0);
    read_set(afp, wfp, stage, 1);
    mfma_all(wf0, af0);
    stage = stage + 1 >= NSTG ? 0 : stage + 1;
  };
  // 16 MFMAs of a fragment set in 8 pairs, one DMA piece in front of each of the first six pairs on the waves whose turn it is
  auto mfma_dma = [&](const Pack8 (&wf)[FN], const Pack8 (&af)[FM], int ns, bool mine) {
    static_for<8>([&](auto gc) {
      constexpr int gi = decltype(gc)::value;
      if constexpr (gi < VMW) {
        if (mine) issue_one(ns, gc);
      }
      static_for<2>([&](auto tc) {
        constexpr int idx = gi * 2 + decltype(tc)::value;
        constexpr int j = idx / FM, i = idx - j * FM;
        acc[i][j] = mfma16<T>(wf[j], af[i], acc[i][j]);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto step_stagger = [&](int kt) {
    step_top(kt);
    Pack8 af0[FM], wf0[FN];
    read_set(af0, wf0, stage, 0);
    __builtin_amdgcn_sched_barrier(0);
    const int ns = next_stage();
    const bool early = wave < C3_NW / 2;
    mfma_dma(wfp, afp, ns, early);
    read_set(afp, wfp, stage, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_dma(wf0, af0, ns, !early);
    advance();
    stage = stage + 1 >= NSTG ? 0 : stage + 1;
  };
  {
    int kt = 0;
    step_block(kt++);
    for (; kt + 2 < nk; ++kt) step_stagger(kt);
    for (; kt < nk; ++kt) step_block(kt);
  }
  mfma_all(wfp, afp);
  if (nk < 2) load_bias();

  // epilogue: bias in fp32, round, C tile through LDS (the ring: everybody must have left the K loop), 16-byte row stores (+ residual)
  __syncthreads();
  T* ctile = reinterpret_cast<T*>(smem);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = wn * 64 + 16 * j + 4 * g;
    const f32x4 bv = bvs[j];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ml = wm * 64 + 16 * i + l15;
      Pack4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) float_to_elem4<T>(o, r, acc[i][j][r] + bv[r]);
      *reinterpret_cast<u32x2*>(ctile + ml * CS + nl) = o.u;
    }
  }
  __syncthreads();
  T* out = reinterpret_cast<T*>(p.out);
  const T* res = reinterpret_cast<const T*>(p.resid);
  const int cchunk = tid & 15, crow0 = tid >> 4;
#pragma unroll 4
  for (int pass = 0; pass < BM / 32; ++pass) {
    const int ml = pass * 32 + crow0;
    const int m = m0 + ml;
    if (m >= M) break;
    Pack8 v = lds_read16(ctile + ml * CS + cchunk * 8);
    const size_t off = (size_t)m * p.Cout + n0 + cchunk * 8;
    if (res) {
      float a[8], b[8];
      unpack8<T>(v, a);
      unpack8<T>(gload16(res + off), b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
      v = pack8<T>(a);
    }
    gstore16(out + off, v);
  }
}

// ---- GroupNorm statistics over NHWC: sums[b][group][2] += (sum x, sum x^2).  Per-thread partials are folded per group through
// LDS atomics (ds_add_f32), so a block issues only 2 * groups global atomics (the first version issued 16 per thread onto
// C * 2 addresses and ran at 1 ms per call on contention alone).
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ block_sums, int HW, int C, int groups, int pix_per_block) {
  const int b = blockIdx.y;
  const int cpr = C >> 3;  // 16-byte chunks per pixel
  const int cg = C / groups;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  // thread t owns channel chunk t % cpr and walks pixels with stride 256 / cpr
  const int c = threadIdx.x % cpr, pstep = 256 / cpr, pofs = threadIdx.x / cpr;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  if (pofs < pstep) {
    const T* xp = x + ((size_t)b * HW + p0 + pofs) * C + c * 8;
    const size_t stride = (size_t)pstep * C;
    int pix = p0 + pofs;
    for (; pix + 3 * pstep < p1; pix += 4 * pstep, xp += 4 * stride) {  // 4 loads in flight per lane
      const Pack8 r0 = gload16(xp), r1 = gload16(xp + stride), r2 = gload16(xp + 2 * stride), r3 = gload16(xp + 3 * stride);
      float v[8];
#define ECAD_GN_ACC(r)                        \
  unpack8<T>(r, v);                           \
  _Pragma("unroll") for (int e = 0; e < 8; ++e) { \
    s[e] += v[e];                             \
    q[e] = __builtin_fmaf(v[e], v[e], q[e]);  \
  }
      ECAD_GN_ACC(r0) ECAD_GN_ACC(r1) ECAD_GN_ACC(r2) ECAD_GN_ACC(r3)
    }
    for (; pix < p1; pix += pstep, xp += stride) {
      float v[8];
      const Pack8 r0 = gload16(xp);
      ECAD_GN_ACC(r0)
    }
#undef ECAD_GN_ACC
  }
  // fixed-order reduction (round 3: the first version folded the partials with LDS and global atomics, and two decodes of the same latents differed by one
  // level on ~7 % of the bytes): every thread leaves its partial sums in LDS, thread t < 2 * groups adds the 256 / groups (x 2 for 4-channel groups)
  // partials of its (group, sum | sum of squares) in thread order, and the block writes ONE row of per-block partials that gn_finalize_kernel adds in block order.
  __shared__ float part[256][4];
  {
    float lo_s = 0.f, lo_q = 0.f, hi_s = 0.f, hi_q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) lo_s += s[e], lo_q += q[e], hi_s += s[4 + e], hi_q += q[4 + e];
    part[threadIdx.x][0] = lo_s; part[threadIdx.x][1] = lo_q; part[threadIdx.x][2] = hi_s; part[threadIdx.x][3] = hi_q;
  }
  __syncthreads();
  if (threadIdx.x < 2 * groups) {
    const int gi = threadIdx.x >> 1, which = threadIdx.x & 1;
    float tot = 0.f;
    if (cg >= 8) {
      const int c_lo = gi * cg / 8, c_hi = (gi + 1) * cg / 8;
      for (int po = 0; po < pstep; ++po)
        for (int cc = c_lo; cc < c_hi; ++cc) tot += part[po * cpr + cc][which] + part[po * cpr + cc][2 + which];
    } else {  // cg == 4: chunk gi / 2, its lower or upper half
      const int cc = gi >> 1, half = gi & 1;
      for (int po = 0; po < pstep; ++po) tot += part[po * cpr + cc][2 * half + which];
    }
    block_sums[((size_t)b * gridDim.x + blockIdx.x) * groups * 2 + threadIdx.x] = tot;
  }
}

// scale[b][c] = rstd * gamma[c], shift[b][c] = beta[c] - mean * rstd * gamma[c]   (eps 1e-6).  One block per image, 32 groups: thread (q, v) adds every fourth row of
// per-block partials of value v = (group, sum | sum of squares) in block order, the four quarters are added in order -- the same sum on every launch.  (Rounds 1-5
// looped over all rows once per CHANNEL: 30 us at 512 rows, 0.3-0.6 ms per decode.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ block_sums, int nblk, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ scale, float* __restrict__ shift, int C, int HW, int groups, float eps) {
  __shared__ float red[4][64];
  __shared__ float mean_s[32], rstd_s[32];
  const int b = blockIdx.x, v = threadIdx.x & 63, q = threadIdx.x >> 6;
  const float* src = block_sums + (size_t)b * nblk * 64 + v;
  float t = 0.f;
  int k = q;
  for (; k + 12 < nblk; k += 16) {
    const float t0 = src[(size_t)k * 64], t1 = src[(size_t)(k + 4) * 64], t2 = src[(size_t)(k + 8) * 64], t3 = src[(size_t)(k + 12) * 64];
    t += t0; t += t1; t += t2; t += t3;
  }
  for (; k < nblk; k += 4) t += src[(size_t)k * 64];
  red[q][v] = t;
  __syncthreads();
  const int cg = C / groups;
  if (threadIdx.x < groups) {
    const int gi = threadIdx.x;
    const float s = ((red[0][2 * gi] + red[1][2 * gi]) + red[2][2 * gi]) + red[3][2 * gi];
    const float qq = ((red[0][2 * gi + 1] + red[1][2 * gi + 1]) + red[2][2 * gi + 1]) + red[3][2 * gi + 1];
    const float n = (float)HW * (float)cg;
    const float mean = s / n;
    const float var = fmaxf(qq / n - mean * mean, 0.f);
    mean_s[gi] = mean;
    rstd_s[gi] = rsqrtf(var + eps);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int gi = c / cg;
    const float rg = rstd_s[gi] * gamma[c];
    scale[(size_t)b * C + c] = rg;
    shift[(size_t)b * C + c] = beta[c] - mean_s[gi] * rg;
  }
}

// y = act(x * scale[b][c] + shift[b][c]); act = SiLU (resnets, output norm) or identity (attention group_norm)
template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, size_t total_chunks, int HW, int C) {
  const int cpr = C >> 3;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total_chunks; idx += (size_t)gridDim.x * 256) {
    const size_t pix = idx / cpr;
    const int c = (int)(idx - pix * cpr) * 8;
    const int b = (int)(pix / HW);
    float v[8];
    unpack8<T>(gload16(x + idx * 8), v);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c), s1 = *reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + (size_t)b * C + c), h1 = *reinterpret_cast<const f32x4*>(shift + (size_t)b * C + c + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = v[e] * s0[e] + h0[e];
      v[4 + e] = v[4 + e] * s1[e] + h1[e];
    }
    if (SILU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
    }
    gstore16(y + idx * 8, pack8<T>(v));
  }
}

// ---- first layer: latents fp32 NCHW [B,CL,h,w] / scaling + shift -> [post_quant_conv (1x1, CL->CL)] -> conv_in (3x3, CL->C) -> NHWC
// CL = 4 (SD / PixArt VAE, with post_quant_conv) or 16 (FLUX VAE: no post_quant_conv, shift_factor 0.1159).
// A block owns CIN_PIX consecutive pixels of one image; each thread keeps the CL*9 weights of its channel(s) in registers.
constexpr int CIN_PIX = 32;
template <typename T, int CL>
__global__ __launch_bounds__(256) void vae_conv_in_kernel(const float* __restrict__ lat, const float* __restrict__ pq_w, const float* __restrict__ pq_b,
                                                          const float* __restrict__ w /*[C][CL][3][3]*/, const float* __restrict__ bias, T* __restrict__ out,
                                                          int B, int H, int W, int C, float inv_scale, float shift) {
  constexpr int KK = CL * 9;
  const int pix0 = blockIdx.x * CIN_PIX, b = blockIdx.y;
  __shared__ float z[CIN_PIX][KK];  // [pixel][ci * 9 + tap]: (post_quant_conv'd) latent at the 9 neighbours (0 outside the image)
  for (int id = threadIdx.x; id < CIN_PIX * KK; id += 256) {
    const int pl = id / KK, k = id % KK, co = k / 9, tap = k % 9;
    const int pix = pix0 + pl;
    const int y = pix / W, x = pix % W;
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    float v = 0.f;
    if (pix < H * W && yy >= 0 && yy < H && xx >= 0 && xx < W) {
      if (pq_w) {
        v = pq_b[co];
        for (int ci = 0; ci < CL; ++ci) v += pq_w[co * CL + ci] * (lat[(((size_t)b * CL + ci) * H + yy) * W + xx] * inv_scale + shift);
      } else {
        v = lat[(((size_t)b * CL + co) * H + yy) * W + xx] * inv_scale + shift;
      }
    }
    z[pl][k] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float wr[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) wr[k] = w[c * KK + k];
    const float bc = bias[c];
    for (int pl = 0; pl < CIN_PIX && pix0 + pl < H * W; ++pl) {
      float a = bc;
#pragma unroll
      for (int k = 0; k < KK; ++k) a = __builtin_fmaf(wr[k], z[pl][k], a);
      out[((size_t)b * H * W + pix0 + pl) * C + c] = (T)a;
    }
  }
}

// ---- last layer: conv_norm_out (apply + SiLU) + conv_out 3x3 (128 -> 3) + (v/2 + 0.5) clamp -> uint8 NHWC (PIL-ready).
// A block owns an 8 x 16 tile of output pixels of one image.  The 10 x 18 halo of the RAW pre-norm activations is normalised,
// SiLU'd and staged once in LDS (pixel stride 272 B so the 16 lanes of a fragment read hit 16 distinct 4-bank groups); the conv is
// an MFMA GEMM [pixels x (9 taps * 128)] x [(9 * 128) x 16] with the 3 real output channels in rows 0..2 of a zero-padded weight
// whose 36 B-fragments live in registers for the whole block.  Wave w owns tile rows 2w and 2w+1 (one 16-pixel fragment each).
constexpr int CO_TH = 8, CO_TW = 16, CO_C = 128, CO_PIX_BYTES = CO_C * 2 + 16, CO_HALO_W = CO_TW + 2, CO_HALO_H = CO_TH + 2;
constexpr int CO_LDS_BYTES = CO_HALO_H * CO_HALO_W * CO_PIX_BYTES;  // 48960
template <typename T>
__global__ __launch_bounds__(256) void vae_conv_out_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const T* __restrict__ w /*[16][9][128], rows 3..15 zero*/, const float* __restrict__ bias,
                                                           uint8_t* __restrict__ out, int B, int H, int W) {
  __shared__ __attribute__((aligned(16))) char halo[CO_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int tiles_x = W / CO_TW;
  const int b = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * CO_TH, x0 = tx * CO_TW;

  // weights: B fragment ks (tap = ks / 4, channels (ks % 4) * 32 + g * 8 ..) of output row l15
  Pack8 wf[36];
#pragma unroll
  for (int ks = 0; ks < 36; ++ks) wf[ks] = gload16(w + (size_t)l15 * (9 * CO_C) + ks * 32 + g * 8);

  // stage the halo: thread -> channel chunk tid % 16 (its scale/shift stay in registers), pixel tid / 16 + 16 * it
  {
    const int ch = tid & 15;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = scale[(size_t)b * CO_C + ch * 8 + e];
      sh[e] = shift[(size_t)b * CO_C + ch * 8 + e];
    }
    // all of the thread's halo chunks are requested before the first is normalised (round 6: the loop asked for one at a time -- load, SiLU, LDS write, next --
    // and the kernel ran at a sixth of the HBM rate)
    constexpr int NIT = (CO_HALO_H * CO_HALO_W + 15) / 16;   // 12
    Pack8 raw[NIT];
    bool ok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int hp = (tid >> 4) + 16 * it;
      const int hy = hp / CO_HALO_W, hx = hp - hy * CO_HALO_W;
      const int yy = y0 + hy - 1, xx = x0 + hx - 1;
      ok[it] = hp < CO_HALO_H * CO_HALO_W && yy >= 0 && yy < H && xx >= 0 && xx < W;
      raw[it].u = (u32x4){0u, 0u, 0u, 0u};
      if (ok[it]) raw[it] = gload16(x + (((size_t)b * H + yy) * W + xx) * CO_C + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int hp = (tid >> 4) + 16 * it;
      Pack8 v;
      v.u = (u32x4){0u, 0u, 0u, 0u};
      if (ok[it]) {
        float f[8];
        unpack8<T>(raw[it], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = silu(f[e] * sc[e] + sh[e]);
        v = pack8<T>(f);
      }
      if (hp < CO_HALO_H * CO_HALO_W) *reinterpret_cast<u32x4*>(halo + hp * CO_PIX_BYTES + ch * 16) = v.u;
    }
  }
  __syncthreads();

  f32x4 acc[2];
  acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int hp = (wave * 2 + f + dy) * CO_HALO_W + l15 + dx;
        const Pack8 af = lds_read16(halo + hp * CO_PIX_BYTES + (kc * 4 + g) * 16);
        acc[f] = mfma16<T>(wf[tap * 4 + kc], af, acc[f]);
      }
    }
  }
  // lane (l15, g = 0) holds output channels 0..3 (3 real) of pixel l15 of each fragment
  if (g == 0) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int yy = y0 + wave * 2 + f, xx = x0 + l15;
      uint8_t* o = out + (((size_t)b * H + yy) * W + xx) * 3;
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const float v = acc[f][n] + bias[n];
        o[n] = (uint8_t)__float2int_rn(fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f) * 255.f);
      }
    }
  }
}

// ---- mid-block attention helpers (one head of C = 512 over HW = h*w tokens)
// row softmax of S[rows][n] (fp16/bf16 in place), scale applied first
template <typename T> __global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ s, int n, float scale) {
  T* row = s + (size_t)blockIdx.x * n;
  __shared__ float red[4];
  float m = -1e30f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, (float)row[i] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) sum += __expf((float)row[i] * scale - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int i = threadIdx.x; i < n; i += 256) row[i] = (T)(__expf((float)row[i] * scale - m) * inv);
}

// out[c][r] = in[r][c]  (rows x cols -> cols x rows), per batch slice blockIdx.z
template <typename T> __global__ void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int rows, int cols) {
  __shared__ T tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  int r = blockIdx.y * 32 + threadIdx.y, c = blockIdx.x * 32 + threadIdx.x;
  for (int k = 0; k < 32; k += 8)
    if (r + k < rows && c < cols) tile[threadIdx.y + k][threadIdx.x] = in[base + (size_t)(r + k) * cols + c];
  __syncthreads();
  r = blockIdx.x * 32 + threadIdx.y;
  c = blockIdx.y * 32 + threadIdx.x;
  for (int k = 0; k < 32; k += 8)
    if (r + k < cols && c < rows) out[base + (size_t)(r + k) * rows + c] = tile[threadIdx.x][threadIdx.y + k];
}

}  // namespace ecad
