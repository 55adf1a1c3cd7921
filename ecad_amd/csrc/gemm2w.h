// gemm2w.h -- wide-tile sibling of gemm3s.h for the large-N GEMMs (FF1: N = 4608 = 16 * 288; text K/V).
//
// The K loop of the 256x144 kernels is power/DVFS-limited and the clock tracks the bytes moved per flop (1.28-1.66 GHz
// with the LDS-DMA stream, 1.87-2.0 GHz with it ablated).  A 256x288 tile moves (256+288)/(256*288) = 0.0074 B/flop
// through L2->LDS instead of 0.0109 (-32 %), and 8 waves laid out 4 (M) x 2 (N) own 64x144 each, so the LDS fragment
// reads drop from 0.61 to 0.36 per MFMA.  Price: only a 2-deep stage ring fits (2 x 69.6 KiB), i.e. one K-step of
// prefetch (a step is 144 MFMAs per SIMD, ~1.2 us, enough for L2-resident operands), and N % 288 == 0.
// Measured negative: a persistent walk over the 2 tiles per CU of FF1 with the next tile's first K-step issued under the current tile's
// last step and the epilogue written straight from the accumulators (8-byte stores, no LDS round trip, ring left free for the DMA):
// 95.8 -> 102.6 us per launch -- the 32-byte row segments of the direct stores cost more than the overlap returns.
#pragma once
#include "common.h"
#include "gemm.h"
#include "gemm3s.h"  // ln_row_coeffs

namespace ecad {

constexpr int G2_BM = 256, G2_BN = 288, G2_NW = 8;
constexpr int G2_A_BYTES = G2_BM * 128;
constexpr int G2_STAGE_BYTES = (G2_BM + G2_BN) * 128;  // 69632
constexpr int G2_CS = G2_BN + 8;
constexpr int G2_TILE_BYTES = (2 * G2_STAGE_BYTES) > (G2_BM * G2_CS * 2) ? (2 * G2_STAGE_BYTES) : (G2_BM * G2_CS * 2);  // 151552: operand ring / C tile
constexpr int G2_LDS_BYTES = G2_TILE_BYTES + G2_BM * 8;  // + (rstd, -rstd * mean) of the tile's 256 rows for a folded LayerNorm (gemm.h): 153600

template <typename T, int EPI, int TAG>
__global__ __launch_bounds__(G2_NW * 64) void gemm2w_kernel(const GemmArgs<T> p) {
  constexpr int FM = 4, FN = 9;
  constexpr int A_INST = G2_BM / 8 / G2_NW;  // 4
  constexpr int B_TOTAL = G2_BN / 8;         // 36 -> waves 0-3 issue 5, waves 4-7 issue 4
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;  // waves w and w+4 share a SIMD: one of each column half per SIMD
  const int l15 = lane & 15, g = lane >> 4;
  const int tiles_n = p.N / G2_BN;
  const int tiles_m = ((p.m_end ? p.m_end : p.M) - p.m_begin + G2_BM - 1) / G2_BM;
  const int nwg = tiles_m * tiles_n;
  // tile id -> origin: ids are dealt to the XCDs in contiguous runs (tile_coords, gemm.h)
  auto origin = [&](int v, int& om0, int& on0) {
    const int q = nwg >> 3, r = nwg & 7, xcd = v & 7, idx = v >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    int tm, tn;
    tile_coords(bid, tiles_m, tiles_n, tm, tn);
    om0 = p.m_begin + tm * G2_BM;
    on0 = tn * G2_BN;
  };
  const T* a_src[A_INST];
  const T* b_src[5];
  auto set_sources = [&](int sm0, int sn0) {
    // `ln` is opaque to the optimizer: otherwise the lane-dependent halves of these nine addresses are hoisted out of the tile loop and stay live through
    // the K loop and the epilogue of every tile (26 spilled VGPRs in a kernel that sits at 255)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const int r = (wave * A_INST + i) * 8 + (ln >> 3);
      const int c = (ln & 7) ^ ((r >> 1) & 7);
      int row = sm0 + r;
      row = row < p.M ? row : p.M - 1;
      a_src[i] = p.A + (size_t)row * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int inst = wave + i * G2_NW;
      inst = inst < B_TOTAL ? inst : B_TOTAL - 1;
      const int r = inst * 8 + (ln >> 3);
      const int c = (ln & 7) ^ ((r >> 1) & 7);
      b_src[i] = p.W + (size_t)(sn0 + r) * p.ldw + c * 8;
    }
  };
  // Tile loop (round 3): the grid may be smaller than the tile count (FF1 with the direct epilogue: one workgroup per CU walks its 2 .. 8 tiles).  The direct
  // epilogue leaves the LDS alone, so the first K-step of the NEXT tile is requested before this tile's epilogue starts: no pipeline fill, no workgroup turnover,
  // and the stores drain under the next K loop.
  bool pre_issued = false;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
  int m0, n0;
  origin(vb, m0, n0);
  set_sources(m0, n0);  // (recomputed rather than kept across the previous tile's epilogue: 18 live registers there are 18 spills)
  if (pre_issued) {     // K-step 0 is already on its way
#pragma unroll
    for (int i = 0; i < A_INST; ++i) a_src[i] += GEMM_BK;
#pragma unroll
    for (int i = 0; i < 5; ++i) b_src[i] += GEMM_BK;
  }
  auto issue = [&](int stage) {
    char* abase = smem + stage * G2_STAGE_BYTES;
    char* bbase = abase + G2_A_BYTES;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      glds16(a_src[i], abase + (wave * A_INST + i) * 1024);
      a_src[i] += GEMM_BK;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int inst = wave + i * G2_NW;
      if (inst < B_TOTAL) glds16(b_src[i], bbase + inst * 1024);
      b_src[i] += GEMM_BK;
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wm * 64 + l15) * 128;
  const int b_row_off = (wn * 144 + l15) * 128;
  const int nk = p.K / GEMM_BK;

#ifdef G_STAMP
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_c1 = 0, st_c2 = 0, st_ca = 0, st_cb = 0;
#endif
  if (!pre_issued) {
    // a later tile of the walk whose first K-step was not requested early (odd step count: the last step read stage 0): every wave must have left the previous
    // tile's K loop before its ring is overwritten (the direct epilogue has no barrier of its own; the staged one ends with one)
    if (vb != (int)blockIdx.x) __syncthreads();
    issue(0);
  }
  pre_issued = false;
  // the last PN columns of a step's second fragment set are consumed after the NEXT barrier (4 * PN MFMAs of pipe work while the barrier
  // resolves and the new stage's first fragments arrive); the full set does not fit next to 144 accumulators (PN = 2: 254 VGPRs, no spills).
  // In-run A/B: FF1 -1.6 %, wide K,V -1.5 %, +0.8 % end to end; PN = 4 the same.
  constexpr int PN = 2;
  Pack8 afp[FM], wfp[PN];
  auto step_top = [&](int kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef G_STAMP
    if (kt == 0) st_c1 = __builtin_amdgcn_s_memtime();
#endif
  };
  auto read_set0 = [&](Pack8 (&af0)[FM], Pack8 (&wf0)[FN], int stage) {
    const char* abase = smem + stage * G2_STAGE_BYTES;
    const char* bbase = abase + G2_A_BYTES;
    const int coff = (g ^ swz) * 16;
#pragma unroll
    for (int i = 0; i < FM; ++i) af0[i] = lds_read16(abase + a_row_off + i * 2048 + coff);
#pragma unroll
    for (int j = 0; j < FN; ++j) wf0[j] = lds_read16(bbase + b_row_off + j * 2048 + coff);
  };
  auto read_set1 = [&](Pack8 (&wf)[FN - PN], int stage) {
    const char* abase = smem + stage * G2_STAGE_BYTES;
    const char* bbase = abase + G2_A_BYTES;
    const int coff = ((4 + g) ^ swz) * 16;
#pragma unroll
    for (int i = 0; i < FM; ++i) afp[i] = lds_read16(abase + a_row_off + i * 2048 + coff);
#pragma unroll
    for (int j = 0; j < FN - PN; ++j) wf[j] = lds_read16(bbase + b_row_off + j * 2048 + coff);
#pragma unroll
    for (int j = 0; j < PN; ++j) wfp[j] = lds_read16(bbase + b_row_off + (FN - PN + j) * 2048 + coff);
  };
  // Block schedule (round 1; also the first and the last K-step of the staggered schedule): all 9 pieces of the next stage in one block
  auto step_block = [&](int kt) {
    const int stage = kt & 1;
    step_top(kt);
    Pack8 af0[FM], wf0[FN];
    read_set0(af0, wf0, stage);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) issue(stage ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if (kt > 0) {
#pragma unroll
      for (int j = 0; j < PN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][FN - PN + j] = mfma16<T>(wfp[j], afp[i], acc[i][FN - PN + j]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf0[j], af0[i], acc[i][j]);
    Pack8 wf[FN - PN];
    read_set1(wf, stage);
#pragma unroll
    for (int j = 0; j < FN - PN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf[j], afp[i], acc[i][j]);
  };
  // (measured NEGATIVE on this kernel: the staggered LDS-DMA issue of gemm3s.h -- with a 2-stage ring the pieces the late half issues in the second half of a step
  // have half a step to land before the vmcnt(0) at the next barrier: wide K,V 44.6 -> 49.2 us, FF1 92.6 -> 100.8 us; with every wave spreading its pieces over the
  // first 36 MFMAs of the step instead: 3213 -> 2894 cycles per K-step and the SAME time, the in-kernel clock falling from 1.85 to 1.77 GHz -- the loop is
  // power-limited, cycles saved are not time saved: DESIGN 5h)
  for (int kt = 0; kt < nk; ++kt) step_block(kt);
#pragma unroll
  for (int j = 0; j < PN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][FN - PN + j] = mfma16<T>(wfp[j], afp[i], acc[i][FN - PN + j]);
#ifdef G_STAMP
  st_c2 = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
    // the last K-step read stage (nk - 1) & 1; with an even step count stage 0 is free for the next tile's first step (the source registers are dead here)
    if (vb + (int)gridDim.x < nwg && p.epi_direct > (EPI == EPI_BIAS_GELU ? 0 : 1) && !(nk & 1)) {
      int xm0, xn0;
      origin(vb + (int)gridDim.x, xm0, xn0);
      set_sources(xm0, xn0);
      issue(0);
      pre_issued = true;
    }
  }
#if defined(ECAD_DIAG) && defined(G2_BREAK_TILE)
  // deliberately wrong DIAGNOSTIC build for checking the parity net (make diag only: the product library cannot be built with it): tile column 1 loses one 16-column fragment of its second column half
  if (n0 == G2_BN && wn == 1)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][3] = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
  // all FN bias vectors of this lane in ONE batch before the barrier (see gemm3s.h: hipcc otherwise serializes nine load -> wait -> use round trips)
  f32x4 bvs[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    bvs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bvs[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 144 + 16 * j + 4 * g);
  }
  // LayerNorm folded into this GEMM (gemm.h): acc <- rstd_m * acc - rstd_m * mean_m * c1[n]; the bias vectors above already hold c2
  if (p.ln_stat) {
    // one thread per row sums the row's LN_PARTS partial statistics (16 live registers, one L2 round trip for the workgroup) and leaves (rstd, -rstd * mean)
    // behind the tile in LDS; taking them per lane (FM rows x LN_PARTS loads = 64 registers next to 144 accumulators + 72 of bias / c1) spilled
    f32x2* rowab = reinterpret_cast<f32x2*>(smem + G2_TILE_BYTES);
    if (tid < G2_BM) {
      float ca[1], cb[1];
      ln_row_coeffs<1>(p, m0 + tid, ca, cb);
      rowab[tid] = (f32x2){ca[0], cb[0]};
    }
    __syncthreads();
    // (touching the statistics and the c1 / c2 lines at kernel start, so that these reads hit L2, was measured: no change)
    float ra[FM], rb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const f32x2 ab = rowab[wm * 64 + 16 * i + l15];
      ra[i] = ab[0];
      rb[i] = ab[1];
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(p.ln_c1 + n0 + wn * 144 + 16 * j + 4 * g);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(acc[i][j][r], ra[i], rb[i] * c1[r]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- direct epilogue (round 3; plain / GELU): no C tile in LDS and no barrier.  A lane holds 4 consecutive columns of a row per 16-column fragment
  // (8 bytes in the storage type); v_permlane16_swap_b32 between the packed registers of fragments j and j+1 exchanges the odd 16-lane rows of one with the
  // even rows of the other (lanes l and l ^ 16 = the same row m, column groups 4g and 4(g ^ 1)), after which the lanes with even g hold 8 consecutive
  // columns of fragment j and the lanes with odd g 8 of fragment j+1: one global_store_dwordx4 per fragment pair writes 64 contiguous bytes of each of
  // 16 rows (a full HBM burst), the ninth fragment goes out in 8-byte stores.  Same values as the staged path, bit for bit.
  if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {  // (SiLU: only the small adaLN GEMMs; hipcc contracts acc + bias into its first multiply differently there)
    if (p.epi_direct > (EPI == EPI_BIAS_GELU ? 0 : 1)) {  // 1: the GELU epilogue only (FF1: +4 % in the pipeline; the plain wide K,V launch measured 4 % SLOWER direct), 2: all
      auto cvt = [&](int i, auto jc) {
        constexpr int j = decltype(jc)::value;
        const f32x4 bv = bvs[j];
        Pack4 o;
        if constexpr (EPI == EPI_BIAS_GELU) {
          const f32x2 v01 = gelu_tanh2((f32x2){acc[i][j][0] + bv[0], acc[i][j][1] + bv[1]});
          const f32x2 v23 = gelu_tanh2((f32x2){acc[i][j][2] + bv[2], acc[i][j][3] + bv[3]});
          float_to_elem4<T>(o, 0, v01[0]); float_to_elem4<T>(o, 1, v01[1]); float_to_elem4<T>(o, 2, v23[0]); float_to_elem4<T>(o, 3, v23[1]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) float_to_elem4<T>(o, r, acc[i][j][r] + bv[r]);
        }
        return o;
      };
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * 64 + 16 * i + l15;
        T* crow = p.C + (size_t)(m < p.M ? m : 0) * p.ldc + n0 + wn * 144;
        static_for<FN / 2>([&](auto pc) {
          constexpr int j = 2 * decltype(pc)::value;
          Pack4 a = cvt(i, std::integral_constant<int, j>{}), b = cvt(i, std::integral_constant<int, j + 1>{});
          asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a.u[0]), "+v"(b.u[0]));
          asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a.u[1]), "+v"(b.u[1]));
          Pack8 v;
          v.u = (u32x4){a.u[0], a.u[1], b.u[0], b.u[1]};
          if (m < p.M) gstore16(crow + 16 * (j + (g & 1)) + 8 * (g >> 1), v);
        });
        if constexpr (FN & 1) {
          const Pack4 o = cvt(i, std::integral_constant<int, FN - 1>{});
          if (m < p.M) *reinterpret_cast<u32x2*>(crow + 16 * (FN - 1) + 4 * g) = o.u;
        }
      }
#ifdef G_STAMP
      if (p.gate_mod && wave == 0 && lane == 0) {  // diagnostic build: slot 6 = GELU + store ISSUE (the direct epilogue has no LDS phases), slot 2 = until the stores have drained
        const unsigned long long st_ci = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_c3 = __builtin_amdgcn_s_memtime(), st_r3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<T*>(p.gate_mod)) + 8 * vb;
        dbg[0] = st_c1 - st_c0; dbg[1] = st_c2 - st_c1; dbg[2] = st_c3 - st_c2; dbg[3] = st_r3 - st_r0; dbg[4] = st_r0; dbg[5] = st_c3 - st_c0; dbg[6] = st_ci - st_c2; dbg[7] = 0;
      }
#endif
      continue;
    }
  }
  __syncthreads();
#ifdef G_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st_ca = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue: bias / activation / rounding of the wave's 64 x 144 block into the row-major C tile, one barrier (rows are shared by the two
  // column-half waves), then wave w streams rows 32w .. 32w+31 out in 16-byte chunks.  Stamped (cycles per tile, 256 WGs): wait for the slowest
  // wave + bias 2.9k, accumulators -> LDS 3.3k (FF1 with GELU: 11.0k), LDS -> global 7.1k (the CU's store path moves ~25 B per clock).
  // Measured NEGATIVE and removed: the same in three column slabs so that the VALU works on slab s+1 while the stores of slab s drain --
  // a wave that issues stores into a full queue stalls, so nothing overlaps and two more barriers are paid: wide K,V 44.6 -> 54 us, FF1 92.6 -> 101 us.
  T* ctile = reinterpret_cast<T*>(smem);
  auto convert = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int nl = wn * 144 + 16 * j + 4 * g;
    const f32x4 bv = bvs[j];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ml = wm * 64 + 16 * i + l15;
      Pack4 o;
      if constexpr (EPI == EPI_BIAS_GELU) {
        const f32x2 v01 = gelu_tanh2((f32x2){acc[i][j][0] + bv[0], acc[i][j][1] + bv[1]});
        const f32x2 v23 = gelu_tanh2((f32x2){acc[i][j][2] + bv[2], acc[i][j][3] + bv[3]});
        float_to_elem4<T>(o, 0, v01[0]); float_to_elem4<T>(o, 1, v01[1]); float_to_elem4<T>(o, 2, v23[0]); float_to_elem4<T>(o, 3, v23[1]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][r] + bv[r];
          if (EPI == EPI_BIAS_SILU) v = silu(v);
          float_to_elem4<T>(o, r, v);
        }
      }
      *reinterpret_cast<u32x2*>(ctile + ml * G2_CS + nl) = o.u;
    }
  };
  auto store_chunk = [&](int ml, int cchunk) {
    const int m = m0 + ml;
    if (m >= p.M) return;
    const int n = n0 + cchunk * 8;
    Pack8 v = lds_read16(ctile + ml * G2_CS + cchunk * 8);
    if (EPI == EPI_BIAS_CACHE_RESID) {
      if (p.C) gstore16(p.C + (size_t)m * p.ldc + n, v);
      T* rp = p.resid + (size_t)m * p.ldr + n;
      float o[8], h[8];
      unpack8<T>(v, o);
      unpack8<T>(gload16(rp), h);
      if (p.gate_table) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.gate_table + n);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.gate_table + n + 4);
        float gt[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        if (p.gate_mod) {
          float gm[8];
          const int b = m / p.rows_per_batch;
          unpack8<T>(gload16(p.gate_mod + (size_t)b * p.gate_mod_stride + n), gm);
#pragma unroll
          for (int e = 0; e < 8; ++e) gt[e] += gm[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] += gt[e] * o[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] += o[e];
      }
      gstore16(rp, pack8<T>(h));
    } else {
      gstore16(p.C + (size_t)m * p.ldc + n, v);
    }
  };
  static_for<FN>([&](auto jc) { convert(jc); });
  __syncthreads();  // rows are shared by the two column-half waves
#ifdef G_STAMP
  st_cb = __builtin_amdgcn_s_memtime();
#endif
  // wave w streams rows 32w .. 32w+31, 36 chunks of 16 bytes per row
  constexpr int CPR = G2_BN / 8;
  constexpr int CHUNKS = 32 * CPR;  // 1152 = 18 per lane
  for (int id = lane; id < CHUNKS; id += 64) {
    const int rl = id / CPR, cchunk = id - rl * CPR;
    store_chunk(wave * 32 + rl, cchunk);
  }
#ifdef G_STAMP
  // diagnostic build only: cycles of prologue / K loop / epilogue and the in-kernel clock, per workgroup, into a buffer riding in the unused gate_mod slot
  if (EPI != EPI_BIAS_CACHE_RESID && p.gate_mod && wave == 0 && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_c3 = __builtin_amdgcn_s_memtime(), st_r3 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<T*>(p.gate_mod)) + 8 * vb;  // one record per TILE (a walking workgroup leaves several)
    dbg[0] = st_c1 - st_c0; dbg[1] = st_c2 - st_c1; dbg[2] = st_c3 - st_c2; dbg[3] = st_r3 - st_r0; dbg[4] = st_r0; dbg[5] = st_c3 - st_c0; dbg[6] = st_ca - st_c2; dbg[7] = st_cb - st_ca;
  }
#endif
  if (vb + (int)gridDim.x < nwg) __syncthreads();  // staged epilogue: the C tile has been read before the next tile's DMA overwrites it
  }  // tile loop
}

}  // namespace ecad
