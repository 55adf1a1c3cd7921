// gemm3s.h -- the 256x144-tile GEMM of the PixArt path: one tile per workgroup, 8 mixed waves (each issues its share of the LDS-DMA and
// owns 32 rows x 144 columns), 3-stage ring, one raw barrier per K-step with counted vmcnt.  Serves every N = 1152 shape (Q, both attention
// out-projections, cross-attention Q) and, since its second fragment set is consumed across the K-step barrier, FF2 / caption / the small
// GEMMs as well (FF2: 1.0 PFLOP/s; the persistent producer / consumer kernel of pgemm.h, which it replaced there, is ECAD_GEMM=persistent).
// ECAD_GEMM=classic runs every GEMM on it.
#pragma once
#include "common.h"
#include "gemm.h"
#include "gemm3s_attn.h"

namespace ecad {

constexpr int G3_BM = 256, G3_BN = 144, G3_STAGES = 3, G3_NW = 8;
constexpr int G3_XSRC_MAX = 2;  // cached components replayed in a read-modify-write epilogue (GemmArgs::xsrc)
constexpr int G3_STAGE_BYTES = (G3_BM + G3_BN) * 128;
constexpr int G3_LDS_BYTES = G3_STAGES * G3_STAGE_BYTES;
static_assert(G3A_LDS_BYTES <= G3_LDS_BYTES, "the fused attention tail (Q tile + K/V staging + bias strip) must fit the operand ring: launch_g3_attn requests G3_LDS_BYTES only");

// LayerNorm folded into a consumer GEMM (gemm.h): rstd and -rstd * mean of rows row0, row0 + 16, ... from the per-tile partial sums, summed in tile order
constexpr int LN_PARTS = 8;  // D = 1152 = 8 column tiles of 144 (the engine folds only then)
template <int NR, typename T>
ECAD_DEV void ln_row_coeffs(const GemmArgs<T>& p, int row0, float (&ra)[NR], float (&rb)[NR]) {
  // every kernel that folds a LayerNorm must produce the same two numbers per row: -ffast-math may otherwise re-associate the eight-term sums or contract the
  // products differently from one inlining context to the next (round 5: gemm2w4.h, same source, 1-ulp differences in a few rows)
#pragma clang fp reassociate(off) contract(off)
  f32x2 v[NR][LN_PARTS];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    int m = row0 + 16 * i;
    m = m < p.M ? m : p.M - 1;
#pragma unroll
    for (int t = 0; t < LN_PARTS; ++t) v[i][t] = *reinterpret_cast<const f32x2*>(p.ln_stat + ((size_t)t * p.M + m) * 2);  // all loads in one batch
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    float sx = 0.f, sxx = 0.f;
#pragma unroll
    for (int t = 0; t < LN_PARTS; ++t) { sx += v[i][t][0]; sxx += v[i][t][1]; }  // tile order: the same sum on every launch
    const float mean = sx * p.ln_inv_d;
    const float var = __builtin_fmaf(-mean, mean, sxx * p.ln_inv_d);
    const float rstd = rsqrtf(fmaxf(var, 0.f) + p.ln_eps);
    ra[i] = rstd;
    rb[i] = -rstd * mean;
  }
}

// Latency variant for FEW rows (round 3): BM_ = 64, NW_ = 4, STAGES_ = 3 -- 64 x 144 tiles, four waves of 16 x 144.  A K-step of the 256-row tile is
// ~0.8 us of MFMA work per CU however few tiles the launch has, so at batch 1-4 (M = 512-2048: 16-64 tiles on 256 CUs) every K = 1152 GEMM is an 18-step, ~20 us
// latency chain; here a wave has 18 MFMAs per K-step instead of 36 per wave pair and SIMD, four times as many workgroups share the rows, and two fit a CU.
constexpr int G3Q_BM = 64, G3Q_NW = 4, G3Q_STAGES = 3;
constexpr int G3Q_LDS_BYTES = G3Q_STAGES * (G3Q_BM + G3_BN) * 128;  // 79 872
static_assert(2 * G3Q_LDS_BYTES <= 160 * 1024, "two latency-tile workgroups per CU");

template <typename T, int EPI, int TAG, int BM_ = G3_BM, int NW_ = G3_NW, int STAGES_ = G3_STAGES>
__global__ __launch_bounds__(NW_ * 64, 2) void gemm3s_kernel(const GemmArgs<T> p) {  // two waves per SIMD either way: 8 of one workgroup, or 4 + 4 of two
  constexpr int BM = BM_, BN = G3_BN, NWAVES = NW_, NSTG = STAGES_;
  constexpr int WTM = BM / NWAVES;
  constexpr int FM = WTM / 16, FN = BN / 16;
  static_assert((FM == 1 || FM == 2) && FN == 9 && (NSTG == 2 || NSTG == 3), "wave tile 32 x 144 or 16 x 144");
  static_assert((EPI != EPI_BIAS_ATTN && EPI != EPI_BIAS_ATTN_TAIL) || (BM == WTM * NWAVES && g3a_lds_bytes<BM>() <= NSTG * (BM + BN) * 128 && NSTG == 3),
                "the attention tail (Q tile + K/V staging + bias strip) lives in the operand ring: 256-row tiles, or the 64-row latency tile");
  constexpr int A_BYTES = BM * 128;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int A_INST = BM / 8 / NWAVES;
  constexpr int B_INST_TOTAL = BN / 8;
  constexpr int B_INST_MAX = (B_INST_TOTAL + NWAVES - 1) / NWAVES;
  constexpr int VMW = A_INST + B_INST_TOTAL / NWAVES;  // fewest LDS-DMA pieces any wave issues per stage: what may stay in flight across a 3-stage step top (6)
  constexpr int CS = BN + 8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int tiles_n = p.N / BN;
  const int tiles_m = ((p.m_end ? p.m_end : p.M) - p.m_begin + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  tile_coords(bid, tiles_m, tiles_n, tm, tn);
  const int m0 = p.m_begin + tm * BM, n0 = tn * BN;

  // EPI_SPLITK: this workgroup multiplies K-slice blockIdx.y of its tile (gemm.h)
  const int k_len = EPI == EPI_SPLITK ? p.K / (int)gridDim.y : p.K;
  const int k_first = EPI == EPI_SPLITK ? (int)blockIdx.y * k_len : 0;
  const T* a_src[A_INST];
  const T* b_src[B_INST_MAX];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int r = (wave * A_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int row = m0 + r;
    row = row < p.M ? row : p.M - 1;
    a_src[i] = p.A + (size_t)row * p.lda + c * 8 + k_first;
  }
#pragma unroll
  for (int i = 0; i < B_INST_MAX; ++i) {
    int inst = wave + i * NWAVES;
    inst = inst < B_INST_TOTAL ? inst : B_INST_TOTAL - 1;
    const int r = inst * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[i] = p.W + (size_t)(n0 + r) * p.ldw + c * 8 + k_first;
  }
  auto issue = [&](int stage) {
    char* abase = smem + stage * STAGE_BYTES;
    char* bbase = abase + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      glds16(a_src[i], abase + (wave * A_INST + i) * 1024);
      a_src[i] += GEMM_BK;
    }
#pragma unroll
    for (int i = 0; i < B_INST_MAX; ++i) {
      const int inst = wave + i * NWAVES;
      if (inst < B_INST_TOTAL) glds16(b_src[i], bbase + inst * 1024);
      b_src[i] += GEMM_BK;
    }
  };

  // one piece (1 KiB: 8 rows x 128 B) of this wave's share of a stage; q = 0 .. A_INST + B_INST_MAX - 1
  auto issue_one = [&](int stage, auto qc) {
    constexpr int q = decltype(qc)::value;
    char* abase = smem + stage * STAGE_BYTES;
    if constexpr (q < A_INST) {
      glds16(a_src[q], abase + (wave * A_INST + q) * 1024);
    } else {
      constexpr int i = q - A_INST;
      if constexpr (i < B_INST_MAX) {
        const int inst = wave + i * NWAVES;
        if (inst < B_INST_TOTAL) glds16(b_src[i], abase + A_BYTES + inst * 1024);
      }
    }
  };
  auto advance_src = [&]() {
#pragma unroll
    for (int i = 0; i < A_INST; ++i) a_src[i] += GEMM_BK;
#pragma unroll
    for (int i = 0; i < B_INST_MAX; ++i) b_src[i] += GEMM_BK;
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wave * WTM + l15) * 128;
  const int b_row_off = l15 * 128;
  const int nk = k_len / GEMM_BK;
#ifdef G_STAMP
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_c1 = 0, st_c2 = 0, st_ca = 0, st_cb = 0;
#endif
  issue(0);
  if (NSTG == 3 && nk > 1) issue(1);
  // EPI_BIAS_CACHE_RESID: the hidden-state chunks this lane updates in the epilogue are fetched during the K loop, so their HBM latency hides
  // behind it instead of sitting at the tail of a 1-tile-per-CU kernel -- but not in front of it: vmcnt retires in order, so loads issued
  // before the first K-steps make the pipeline fill wait for them (stamped: prologue 5.7k cycles against 2.6k for the plain epilogue).
  // They go out in K-step 1, right behind that step's DMA issue.  (Split-K layout: 144 accumulators leave no room for them inside the loop -- they go out
  // right behind it and land under the exchange of the accumulator halves.)
  constexpr int CPR = BN / 8;
  constexpr int CHUNKS = WTM * CPR;
  constexpr int NCH = (CHUNKS + 63) / 64;
  Pack8 hpre[EPI == EPI_BIAS_CACHE_RESID ? NCH : 1];
  // The gate (scale_shift_table row + adaLN modulation) depends on the column and the sample only.  When the tile lies inside one sample
  // (always for N = 256 / 4096 tokens and 256-row tiles) lanes 0 .. CPR-1 fetch the 8 gate values of their column chunk with the residual --
  // three small loads per wave instead of three per output chunk at the tail -- and the epilogue reads them back from LDS.
  float gpre[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  Pack8 gmod;
  gmod.u = (u32x4){0, 0, 0, 0};
  bool gate_lds = false;
  if (EPI == EPI_BIAS_CACHE_RESID && p.gate_table) {
    const int m_last = (m0 + BM <= p.M ? m0 + BM : p.M) - 1;
    gate_lds = m0 / p.rows_per_batch == m_last / p.rows_per_batch;
  }
  auto prefetch_resid = [&]() {
    if (EPI != EPI_BIAS_CACHE_RESID) return;
    // the (few, L2-resident) gate loads first, then EXACTLY NCH residual loads, every one issued by every wave (clamped addresses instead of lane conditions): the
    // last K-step's wait counts on it -- step_top lets the newest NCH loads stay in flight across the final barrier
    if (gate_lds && lane < CPR) {
      const int n = n0 + lane * 8;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.gate_table + n);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.gate_table + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gpre[e] = t0[e]; gpre[4 + e] = t1[e]; }
      if (p.gate_mod) gmod = gload16(p.gate_mod + (size_t)(m0 / p.rows_per_batch) * p.gate_mod_stride + n);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      int id = lane + it * 64;
      id = id < CHUNKS ? id : CHUNKS - 1;
      const int rl = id / CPR, cchunk = id - rl * CPR;
      int m = m0 + wave * WTM + rl;
      m = m < p.M ? m : p.M - 1;
      hpre[it] = gload16(p.resid + (size_t)m * p.ldr + n0 + cchunk * 8);  // (rows past M / chunks past the wave's share are never stored: the row pass skips them)
    }
  };
  const bool resid_in_tail = nk >= 3;
  if (!resid_in_tail) prefetch_resid();
  // All FN bias vectors of this lane in ONE batch of loads: hipcc otherwise emits load -> s_waitcnt vmcnt(0) -> use per column group, nine
  // serialized L2 / HBM latencies at the tail of the tile (stamped: 7.4k cycles of epilogue with the global stores ablated, 5.0k batched).
  // Plain epilogues (registers to spare) fetch them two K-steps before the end, so not even one latency is left; the residual epilogue
  // (256 VGPRs in the loop) and the split-K layout fetch them right after it.
  f32x4 bvs[FN];
  auto load_bias = [&]() {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      bvs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias) bvs[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * j + 4 * g);
    }
  };
  // Invariant of the last step's `s_waitcnt vmcnt(NCH)` (step_top): the newest NCH vector-memory operations of every wave at that point are exactly the residual
  // prefetches of step nk - 2 -- no LDS-DMA piece, bias load, store or stamp may be issued between prefetch_resid() and step_top(nk - 1).  Hence: the read-modify-write
  // epilogue never loads its bias inside the loop (below), step_stagger never runs the last two steps, and the stamps sit outside that window.
  const bool bias_in_loop = EPI != EPI_BIAS_CACHE_RESID && EPI != EPI_SPLITK && nk >= 2;
  int stage = 0;
  // top of K-step kt: stage kt has landed when at most the newest stage's DMA (A_INST + 2 = 6 per wave) is outstanding; in the step after the
  // residual prefetch went out behind that DMA its NCH loads are newer still, so they may be outstanding as well
  auto step_top = [&](int kt) {
    // 2-stage ring: the stage of this step is the newest thing in flight (it was issued during the previous step): everything must have landed
    // (last step of a read-modify-write tile: the NCH residual loads went out in the step before, behind every LDS-DMA piece; vmcnt retires in order, so "at most
    // NCH outstanding" means the operands have landed while the residual -- an HBM round trip -- keeps travelling into the epilogue.  Waiting for it here cost
    // the loop ~230 cycles per K-step against the plain epilogue's, 2001 vs 1774: the whole latency, exposed once per tile.)
    if (EPI == EPI_BIAS_CACHE_RESID && NSTG == 3 && resid_in_tail && kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");
    else if (NSTG == 2 || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef G_STAMP
    if (kt == 0) st_c1 = __builtin_amdgcn_s_memtime();
#endif
  };
  constexpr int NPIECE = A_INST + B_INST_MAX;  // LDS-DMA slots of a wave per stage: 7 (the last one only on the waves that own an 18th W piece)
  // second fragment set of a step consumed after the NEXT step's barrier: the matrix pipe has a full group to chew on while the barrier
  // resolves and the new stage's first fragments travel from LDS (11 more live fragment registers: fits this kernel's 72 accumulators;
  // 244 VGPRs, no spills).  In-run A/B: Q -2.7 %, attn1-out -3.7 %, attn2-out -3.2 %, cross-Q -3.0 %.
  Pack8 afp[FM], wfp[FN];
  auto read_set = [&](Pack8 (&af)[FM], Pack8 (&wf)[FN], int stg, int half) {
    const char* abase = smem + stg * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;
    const int coff = ((half * 4 + g) ^ swz) * 16;
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = lds_read16(abase + a_row_off + i * 16 * 128 + coff);
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[j] = lds_read16(bbase + b_row_off + j * 16 * 128 + coff);
  };
  auto mfma_all = [&](const Pack8 (&wf)[FN], const Pack8 (&af)[FM]) {
    static_for<FN>([&](auto jc) {
      static_for<FM>([&](auto ic) {
        constexpr int j = decltype(jc)::value, i = decltype(ic)::value;
        acc[i][j] = mfma16<T>(wf[j], af[i], acc[i][j]);
      });
    });
  };
  // Block schedule (round 1; also the first and the last two K-steps of the staggered schedule): every wave issues its pieces in one block
  auto step_block = [&](int kt) {
    step_top(kt);
    Pack8 af0[FM], wf0[FN];
    read_set(af0, wf0, stage, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NSTG - 1 < nk) {
      int ns = stage + NSTG - 1;
      ns = ns >= NSTG ? ns - NSTG : ns;
      issue(ns);
    }
    if (resid_in_tail && kt == nk - 2) prefetch_resid();
    if (bias_in_loop && kt == nk - 2) load_bias();
    __builtin_amdgcn_sched_barrier(0);
    if (kt > 0) mfma_all(wfp, afp);
    __builtin_amdgcn_sched_barrier(0);
    read_set(afp, wfp, stage, 1);
    mfma_all(wf0, af0);
    stage = stage + 1 >= NSTG ? 0 : stage + 1;
  };
  // Staggered schedule (K-steps 1 .. nk-3: a carried fragment set exists and a stage is still to be fetched, so the body has no conditions).
  // Issuing one LDS-DMA piece stalls the issuing wave for ~60-180 cycles (MI355X_MICROARCH.md, cycle constants) but hardly its SIMD partner; with
  // every wave issuing its 6-7 pieces in one block right after the barrier both waves of a SIMD stall together and the matrix pipe idles
  // (stamped: 1810 cycles per K-step against 1152 of MFMA work, with 8 or with 256 workgroups on the chip).  Here the pieces go out one per
  // group of 3 MFMAs, waves 0-3 (the older half) beside the MFMAs carried over from the last step, waves 4-7 (their SIMD partners) beside the
  // MFMAs of this step's first fragment set: whenever one wave of a SIMD sits in a DMA issue, the other has MFMAs to feed.
  // 18 MFMAs in 6 groups of 3 (always 9 groups of 2 for the 4-wave variant, whose waves issue 9 pieces each)
  constexpr int NMF = FM * FN, GROUPS = (NPIECE > 6 + 1 || NMF % 6) ? 9 : 6, PER = NMF / GROUPS;
  static_assert(NMF % GROUPS == 0 && NPIECE <= GROUPS + 1, "piece / MFMA interleave");
  // both halves run the same MFMA stream (one code path: a wave-group branch around the MFMAs makes hipcc shuffle and spill the accumulators);
  // only the small DMA-issue blocks in between are conditional on the half the wave belongs to
  auto mfma_dma = [&](const Pack8 (&wf)[FN], const Pack8 (&af)[FM], int ns, bool mine) {
    static_for<GROUPS>([&](auto gc) {
      constexpr int gi = decltype(gc)::value;
      if constexpr (gi < NPIECE) {
        if (mine) {
          issue_one(ns, gc);
          if constexpr (gi == GROUPS - 1 && NPIECE > GROUPS) issue_one(ns, std::integral_constant<int, GROUPS>{});
        }
      }
      static_for<PER>([&](auto tc) {
        constexpr int idx = gi * PER + decltype(tc)::value;
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
    int ns = stage + NSTG - 1;
    ns = ns >= NSTG ? ns - NSTG : ns;
    const bool early = wave < NWAVES / 2;
    mfma_dma(wfp, afp, ns, early);
    read_set(afp, wfp, stage, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_dma(wf0, af0, ns, !early);
    advance_src();
    stage = stage + 1 >= NSTG ? 0 : stage + 1;
  };
  // Measured NEGATIVE and removed (round 2): a ping-pong schedule -- every K-step as phase A (both fragment sets -> registers, DMA issue, wait) and
  // phase B (the 36 MFMAs) with a barrier after each, waves 4-7 one barrier behind waves 0-3 so that every SIMD has one wave in A and one in B.
  // Bit-identical results, 2177 cycles per K-step against 1740 staggered / 1810 block: a wave's phase A alone takes ~1100 cycles (22 fragment reads,
  // 6-7 DMA pieces, their waits) -- twice its 576 cycles of MFMA -- so the phases cannot balance.
  {
    int kt = 0;
    step_block(kt++);
    if (p.variant != 0 && NSTG == 3)  // 2-stage ring: a piece issued late in a step has less than a step to land before the next vmcnt(0) (gemm2w.h measured the same)
      for (; kt + 2 < nk; ++kt) step_stagger(kt);  // (a stage is still to be fetched in each of these steps for either ring depth)
    for (; kt < nk; ++kt) step_block(kt);
  }
  mfma_all(wfp, afp);
#ifdef G_STAMP
  st_c2 = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (EPI == EPI_SPLITK) {  // fp32 partial sums of this K-slice, straight from the accumulators: lane (l15, g) holds row l15, columns 4g .. 4g + 3 of every fragment
    float* part = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wave * WTM + 16 * i + l15;
      if (m < p.M) {
#pragma unroll
        for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(part + (size_t)m * p.N + n0 + 16 * j + 4 * g) = acc[i][j];
      }
    }
    return;
  }
  if (!bias_in_loop) load_bias();
  // cached components replayed in this epilogue (GemmArgs::xsrc): their chunks of this lane's rows, requested now so that they travel during the C-tile phase
  Pack8 xs[EPI == EPI_BIAS_CACHE_RESID ? NCH : 1][EPI == EPI_BIAS_CACHE_RESID ? G3_XSRC_MAX : 1];
  if constexpr (EPI == EPI_BIAS_CACHE_RESID) {
    if (p.n_xsrc > 0) {
#pragma unroll
      for (int sx = 0; sx < G3_XSRC_MAX; ++sx) {
        if (sx >= p.n_xsrc) break;
        const T* cbase = reinterpret_cast<const T*>(p.xsrc[sx].cache);
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
          const int id = lane + it * 64;
          const int rl = id / CPR, cchunk = id - rl * CPR;
          const int m = m0 + wave * WTM + rl;
          xs[it][sx].u = (u32x4){0, 0, 0, 0};
          if (id < CHUNKS && m < p.M) xs[it][sx].u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(cbase + (size_t)m * p.ldr + n0 + cchunk * 8));
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (EPI == EPI_BIAS_ATTN || EPI == EPI_BIAS_ATTN_TAIL) {  // the tile is a query projection: its attention runs here (gemm3s_attn.h)
    {
      if (p.ln_stat) {  // LayerNorm folded into this projection: acc <- rstd * acc - rstd * mean * c1; the bias vectors already hold c2 (gemm.h)
        float ra[FM], rb[FM];
        ln_row_coeffs<FM>(p, m0 + wave * WTM + l15, ra, rb);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const f32x4 c1 = *reinterpret_cast<const f32x4*>(p.ln_c1 + n0 + 16 * j + 4 * g);
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(acc[i][j][r], ra[i], rb[i] * c1[r]);
        }
      }
      g3_attention_epilogue<T, EPI == EPI_BIAS_ATTN_TAIL, FM, NWAVES>(p, smem, m0, tn, acc, bvs);
    }
    return;
  }
  // The C tile (and the per-wave gate copies behind it) is staged in LDS rows that only their own wave writes and reads, so the only hazard
  // is the operand ring underneath.  When the last K-step reads stage 2 (nk % 3 == 0: every K of the model) the tile lies in stages 0 and 1,
  // which nobody touches after the last step's barrier: no workgroup barrier at all, and a wave that leaves the loop early streams its rows out
  // while the others still multiply (stamped: 1.3-2k cycles of waiting for the slowest wave gone).  Otherwise: one barrier, as before.
  if constexpr (EPI != EPI_BIAS_CACHE_RESID) {
    if (p.ln_stat) {  // LayerNorm folded into this GEMM (gemm.h): plain / GELU / SiLU epilogues of the unfused and the latency-tile launches
      float ra[FM], rb[FM];
      ln_row_coeffs<FM>(p, m0 + wave * WTM + l15, ra, rb);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(p.ln_c1 + n0 + 16 * j + 4 * g);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(acc[i][j][r], ra[i], rb[i] * c1[r]);
      }
    }
  }
  const bool private_epi = NSTG == 3 && nk >= 2 && (nk - 1) % NSTG == 2 && (BM * CS * (int)sizeof(T) + NWAVES * BN * (int)sizeof(float)) <= 2 * STAGE_BYTES;
  static_assert(BM * CS * (int)sizeof(T) + NWAVES * BN * (int)sizeof(float) <= NSTG * STAGE_BYTES, "C tile + gate copies inside the ring");
  if (!private_epi) __syncthreads();
#ifdef G_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st_ca = __builtin_amdgcn_s_memtime();
#endif
  T* ctile = reinterpret_cast<T*>(smem);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = 16 * j + 4 * g;
    const f32x4 bv = bvs[j];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ml = wave * WTM + 16 * i + l15;
      Pack4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[r];
        if (EPI == EPI_BIAS_GELU) v = gelu_tanh(v);
        if (EPI == EPI_BIAS_SILU) v = silu(v);
        float_to_elem4<T>(o, r, v);
      }
      *reinterpret_cast<u32x2*>(ctile + ml * CS + nl) = o.u;
    }
  }
  float* gwave = reinterpret_cast<float*>(smem + BM * CS * sizeof(T)) + wave * BN;  // behind the C tile; one private copy per wave: no block barrier
  if (gate_lds && lane < CPR) {
    if (p.gate_mod) {
      float gm[8];
      unpack8<T>(gmod, gm);
#pragma unroll
      for (int e = 0; e < 8; ++e) gpre[e] += gm[e];
    }
    *reinterpret_cast<f32x4*>(gwave + lane * 8) = (f32x4){gpre[0], gpre[1], gpre[2], gpre[3]};
    *reinterpret_cast<f32x4*>(gwave + lane * 8 + 4) = (f32x4){gpre[4], gpre[5], gpre[6], gpre[7]};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef G_STAMP
  st_cb = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int id = lane + it * 64;
    const int rl = id / CPR, cchunk = id - rl * CPR;
    const int ml = wave * WTM + rl;
    const int m = m0 + ml;
    if (id >= CHUNKS || m >= p.M) continue;
    const int n = n0 + cchunk * 8;
    Pack8 v = lds_read16(ctile + ml * CS + cchunk * 8);
    if (EPI == EPI_BIAS_CACHE_RESID) {
      if (p.C) gstore16_nt(p.C + (size_t)m * p.ldc + n, v);
      T* rp = p.resid + (size_t)m * p.ldr + n;
      Pack8 hv = hpre[it];
      float o[8], h[8];
      unpack8<T>(v, o);
      unpack8<T>(hv, h);
      if (gate_lds) {
#pragma clang fp reassociate(off) contract(off)  // one fused multiply-add per element, said explicitly: gemm2w4.h's read-modify-write epilogue reproduces this update bit for bit
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gwave + cchunk * 8), g1 = *reinterpret_cast<const f32x4*>(gwave + cchunk * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = __builtin_fmaf(g0[e], o[e], h[e]); h[4 + e] = __builtin_fmaf(g1[e], o[4 + e], h[4 + e]); }
      } else if (p.gate_table) {
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
      Pack8 hn = pack8<T>(h);
      if (p.n_xsrc > 0) {  // the replay kernel's arithmetic on top of the rounded update: h <- round(h + sum gate_i . cache_i), sources in plan order
        float h2[8];
        unpack8<T>(hn, h2);
#pragma unroll
        for (int sx = 0; sx < G3_XSRC_MAX; ++sx) {
          if (sx >= p.n_xsrc) break;
          const SrcDesc d = p.xsrc[sx];
          float cf[8], gate[8];
          unpack8<T>(xs[it][sx], cf);
          if (d.gate_table) {
            unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate_mod) + (size_t)(m / p.rows_per_batch) * p.xsrc_mod_stride + n), gate);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(d.gate_table + n);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(d.gate_table + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gate[e] += t0[e]; gate[4 + e] += t1[e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) h2[e] += gate[e] * cf[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) h2[e] += cf[e];
          }
        }
        hn = pack8<T>(h2);
      }
      gstore16(rp, hn);
      if (p.stat_out) *reinterpret_cast<u32x4*>(ctile + ml * CS + cchunk * 8) = hn.u;  // the wave's own rows of the C tile now hold the updated hidden states
    } else {
      gstore16(p.C + (size_t)m * p.ldc + n, v);
    }
  }
  if constexpr (EPI == EPI_BIAS_CACHE_RESID) {
    if (p.stat_out) {
      // row statistics for a LayerNorm folded into the NEXT GEMM (gemm.h): lane l sums row (l & 31), chunks [9 (l >> 5), 9 (l >> 5) + 9) of the wave's 32 rows in
      // fixed order, the two halves meet through one shuffle: deterministic, and no atomics anywhere.  Values are the ROUNDED hidden states, i.e. exactly what
      // the LayerNorm kernel would read back from memory.
#pragma clang fp reassociate(off) contract(off)  // the order below is the definition of these numbers: gemm2w4.h's read-modify-write epilogue reproduces it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int rl = lane & (WTM - 1), hh = (lane / WTM) & 1;  // 16-row wave tiles: lanes 32-63 repeat lanes 0-31 and do not store
      float sx = 0.f, sxx = 0.f;
#pragma unroll
      for (int c = 0; c < CPR / 2; ++c) {
        float x[8];
        unpack8<T>(lds_read16(ctile + (wave * WTM + rl) * CS + (hh * (CPR / 2) + c) * 8), x);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sx += x[e]; sxx = __builtin_fmaf(x[e], x[e], sxx); }
      }
      const float ox = __shfl_xor(sx, WTM), oxx = __shfl_xor(sxx, WTM);
      const int m = m0 + wave * WTM + rl;
      if (lane < WTM && m < p.M) {
        float* dst = p.stat_out + ((size_t)tn * p.M + m) * 2;
        dst[0] = sx + ox;    // lower half + upper half: the same order on every launch
        dst[1] = sxx + oxx;
      }
    }
  }
#ifdef G_STAMP
  // diagnostic build only (see gemm2w.h): the buffer rides in gate_mod, which the epilogue does not read while gate_table is null
  if (!p.gate_table && p.gate_mod && wave == 0 && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_c3 = __builtin_amdgcn_s_memtime(), st_r3 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<T*>(p.gate_mod)) + 8 * blockIdx.x;
    dbg[0] = st_c1 - st_c0; dbg[1] = st_c2 - st_c1; dbg[2] = st_c3 - st_c2; dbg[3] = st_r3 - st_r0; dbg[4] = st_r0; dbg[5] = st_c3 - st_c0; dbg[6] = st_ca - st_c2; dbg[7] = st_cb - st_ca;
  }
#endif
}

// Second half of a split-K read-modify-write GEMM (EPI_SPLITK above): out = round(sum over the K-slices, in slice order, + bias), then the epilogue of gemm3s_kernel
// for EPI_BIAS_CACHE_RESID on that value -- cache slot, h += gate . out, the replayed cached components (xsrc), and the per-tile row statistics of the updated h.
// One wave per (row, 144-column tile): lanes 0 .. 17 own one 16-byte chunk each, the statistics meet through a fixed butterfly (deterministic: graph == eager,
// run == re-run).  The K-slices are summed in fp32 before the one rounding, so the result differs from the sequential-K kernels in the last bits (tests hold it to
// the oracle and to the unsplit kernel within a tolerance, not bit for bit).
template <typename T>
__global__ __launch_bounds__(256) void splitk_rmw_kernel(const GemmArgs<T> p, int slices) {
  constexpr int CPR = G3_BN / 8;
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tiles_n = p.N / G3_BN;
  const int m = w / tiles_n, tn = w - m * tiles_n;
  if (m >= p.M) return;  // wave-uniform
  const bool act = lane < CPR;
  const int n = tn * G3_BN + (act ? lane : 0) * 8;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < slices; ++sl) {
    const float* q = p.part + ((size_t)sl * p.M + m) * p.N + n;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(q), q1 = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] += q0[e]; a[4 + e] += q1[e]; }
  }
  if (p.bias) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] += b0[e]; a[4 + e] += b1[e]; }
  }
  const Pack8 v = pack8<T>(a);
  float sx = 0.f, sxx = 0.f;
  if (act) {
    if (p.C) gstore16_nt(p.C + (size_t)m * p.ldc + n, v);
    T* rp = p.resid + (size_t)m * p.ldr + n;
    float o[8], h[8];
    unpack8<T>(v, o);
    unpack8<T>(gload16(rp), h);
    const int b = m / p.rows_per_batch;
    if (p.gate_table) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.gate_table + n), t1 = *reinterpret_cast<const f32x4*>(p.gate_table + n + 4);
      float gt[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      if (p.gate_mod) {
        float gm[8];
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
    Pack8 hn = pack8<T>(h);
    if (p.n_xsrc > 0) {
      float h2[8];
      unpack8<T>(hn, h2);
      for (int sxi = 0; sxi < p.n_xsrc; ++sxi) {
        const SrcDesc d = p.xsrc[sxi];
        float cf[8], gate[8];
        Pack8 cv;
        cv.u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(d.cache) + (size_t)m * p.ldr + n));
        unpack8<T>(cv, cf);
        if (d.gate_table) {
          unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate_mod) + (size_t)b * p.xsrc_mod_stride + n), gate);
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(d.gate_table + n), t1 = *reinterpret_cast<const f32x4*>(d.gate_table + n + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { gate[e] += t0[e]; gate[4 + e] += t1[e]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) h2[e] += gate[e] * cf[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) h2[e] += cf[e];
        }
      }
      hn = pack8<T>(h2);
    }
    gstore16(rp, hn);
    float x[8];
    unpack8<T>(hn, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) { sx += x[e]; sxx = __builtin_fmaf(x[e], x[e], sxx); }
  }
  if (p.stat_out) {  // (sum x, sum x^2) of this tile's 144 columns of the updated row: lanes 18 .. 63 contribute zeros, fixed butterfly over the first 32 lanes
#pragma unroll
    for (int o2 = 16; o2 > 0; o2 >>= 1) { sx += __shfl_xor(sx, o2); sxx += __shfl_xor(sxx, o2); }
    if (lane == 0) {
      float* dst = p.stat_out + ((size_t)tn * p.M + m) * 2;
      dst[0] = sx;
      dst[1] = sxx;
    }
  }
}

}  // namespace ecad
