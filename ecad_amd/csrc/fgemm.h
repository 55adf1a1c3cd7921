// fgemm.h -- MFMA GEMM for the FLUX blocks (D = 3072 = 12 x 256: the 144-column tiles of the PixArt kernels do not divide it).
//
//   C[M,N] = epilogue( [A | A2][M,K] @ W[N,K]^T + bias[N] )
//
// Replaces the cuBLAS calls behind diffusers' FluxAttnProcessor2_0 projections, FeedForward, proj_mlp / proj_out and the
// adaLN linears (reference call sites: ecad/transformer_blocks/cached_flux_transformer_block.py:63-66,92-93,178-182,215-216)
// and the gated residual tails `hidden = hidden + gate * out` (:128-129,256-257,267-269,273-274,285-287).
//
// Same LDS image / swizzle / LDS-DMA / MFMA operand order as gemm.h.  256 x BN tile, 8 waves laid out 4 (M) x 2 (N), each
// owning 64 x BN/2:
//   BN = 256: 2-stage ring (128 KiB), 0.0078 B of L2->LDS traffic per flop  -- the wide tile for N = 9216 / 12288
//   BN = 128: 3-stage ring (144 KiB), counted vmcnt                       -- N = 3072 on 8192 image rows = 768 tiles = 3.0 rounds
// Split A: K-steps >= k_split read A2 (lda2) instead of A -- the single-stream block's proj_out consumes
// cat([attn, gelu(mlp)]) straight from the two cache slots (cached_flux_transformer_block.py:120-127) without a concat copy.
// Measured negative: stream-K (persistent workgroups over contiguous (tile, K-step) ranges, fp32 partials + per-tile arrival counter,
// last arriver sums the parts in workgroup order and runs the epilogue; built to recover the partially filled last round -- 432 tiles
// on 256 CUs, the joint blocks' text-row GEMMs at 37-75 % of one round).  Bit-exact and parity-green, but the agent-scope
// release / acquire around the counter is a whole-L2 write-back + invalidate on this multi-XCD part: single proj_out 842 -> 1255 us,
// joint attn_out 125 -> 242 us.  A fence-free variant (partials as agent-scope relaxed atomics = sc1 write-through stores / loads,
// ordered against the relaxed counter add by s_waitcnt vmcnt(0)) is also exact but still loses: proj_out 850 -> 1107 us, joint QKV
// +53 %; only the text-row FF2 gained (-5 %).  Workgroups at unrelated K offsets of neighbouring tiles no longer share A / W panels in
// L2 at the same time, and 56 spilled VGPRs sit in the segment loop.  Not in the tree.
// Row -> sample index is segment aware: rows [0, seg0) are text rows (rows_per_batch0 each), the rest image rows
// (rows_per_batch1 each), the layout of the joint [text | image] hidden-state matrix (flux.hip).
#pragma once
#include "common.h"
#include "gemm.h"

namespace ecad {

// FEPI_QKV_ROPE: bias, then per-head RMSNorm * weight + RoPE on the columns < rope_cols (the q and k thirds of a fused QKV output;
//   FluxAttnProcessor2_0 [ext]: norm_q / norm_k, apply_rope) -- a 128-column head never straddles a tile, so the epilogue's row pass does
//   it on 16-lane groups (one (row, head) each) and the separate in-place pass over the QKV buffer disappears.
// FEPI_QKV_ROPE_MLP: the single-stream block's to_q/k/v and proj_mlp share their input (cached_flux_transformer_block.py:108-118):
//   one GEMM over the stacked [3D + 4D, D] weight; columns < n_split as FEPI_QKV_ROPE into C, columns >= n_split bias + GELU into C2.
enum FGemmEpi { FEPI_BIAS = 0, FEPI_BIAS_GELU = 1, FEPI_BIAS_GATE_RESID = 2, FEPI_BIAS_SILU = 3, FEPI_QKV_ROPE = 4, FEPI_QKV_ROPE_MLP = 5 };

template <typename T> struct FGemmArgs {
  const T* A;
  const T* A2;        // K-steps >= k_split come from here (null: single source)
  const T* W;
  const float* bias;  // [N] fp32 or null
  T* C;               // output / cache slot (may be null for FEPI_BIAS_GATE_RESID)
  T* resid;           // FEPI_BIAS_GATE_RESID: resid[m][n] += gate[b(m)][n] * out
  const T* gate;      // gate[b * gate_stride + n]; null => gate 1
  int64_t gate_stride;
  int seg0, rows_per_batch0, rows_per_batch1;
  int M, N, K;
  int k_split;        // in elements (multiple of 64); >= K when A2 is null
  int lda, lda2, ldw, ldc, ldr;
  // FEPI_QKV_ROPE / FEPI_QKV_ROPE_MLP
  const float* rope;    // [S][64][2] (cos, sin) by sequence position
  const float* nq;      // [128] RMSNorm weights of q / k; *_txt (may be null) for the text rows of a joint block
  const float* nk;
  const float* nq_txt;
  const float* nk_txt;
  int rope_cols;        // 2 * D
  int pos_T;            // sequence position of image token j is pos_T + j; of text token j it is j
  float rope_eps;
  float q_scale;        // 0 / 1: none; else the q third leaves the epilogue multiplied by it (softmax scale * log2 e folded into Q before its ONE rounding: attention_w4.h)
  T* C2;                // FEPI_QKV_ROPE_MLP: gelu(mlp) output
  int ldc2, n_split;
  int clip16;           // FEPI_BIAS_GATE_RESID: the updated rows are a FLUX block's output: fp16 saturates at +-65504 (cached_flux_transformer_block.py:127-128,288-289)
  int epi_direct;       // GELU tiles leave straight from the registers (permlane-transposed 16-byte stores, gemm2w.h) instead of through the C tile in LDS
};

ECAD_DEV int seg_batch(int m, int seg0, int rpb0, int rpb1) { return m < seg0 ? m / rpb0 : (m - seg0) / rpb1; }

template <int BN> constexpr int fgemm_stages() { return BN == 256 ? 2 : 3; }
template <int BN> constexpr int fgemm_lds_bytes() {
  constexpr int ring = fgemm_stages<BN>() * (256 + BN) * 128;
  constexpr int ctile = 256 * (BN + 8) * 2;
  return ring > ctile ? ring : ctile;
}

// the whole K loop of the tile at (m0, n0) into acc (zeroed here); ends with a barrier: the LDS ring may be reused right away
template <typename T, int BN>
ECAD_DEV void fgemm_mainloop(const FGemmArgs<T>& p, char* smem, int m0, int n0, f32x4 (&acc)[4][BN / 32]) {
  constexpr int BM = 256, NW = 8, STAGES = fgemm_stages<BN>();
  constexpr int FM = 4, FN = BN / 32;  // wave tile 64 x BN/2
  constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
  constexpr int A_INST = BM / 8 / NW;  // 4
  constexpr int B_INST = BN / 8 / NW;  // 4 (BN 256) or 2 (BN 128)
  [[maybe_unused]] constexpr int CS = BN + 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  const T* a_src[A_INST];
  const T* a2_src[A_INST];
  const T* b_src[B_INST];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int r = (wave * A_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int row = m0 + r;
    row = row < p.M ? row : p.M - 1;
    a_src[i] = p.A + (size_t)row * p.lda + c * 8;
    a2_src[i] = p.A2 ? p.A2 + (size_t)row * p.lda2 + c * 8 : nullptr;
  }
#pragma unroll
  for (int i = 0; i < B_INST; ++i) {
    const int r = (wave * B_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[i] = p.W + (size_t)(n0 + r) * p.ldw + c * 8;
  }
  const int nk = p.K / GEMM_BK;
  const int ksplit = p.A2 ? p.k_split / GEMM_BK : nk;
  int kissue = 0;
  auto issue = [&](int stage) {
    char* abase = smem + stage * STAGE_BYTES;
    char* bbase = abase + A_BYTES;
    if (kissue < ksplit) {
#pragma unroll
      for (int i = 0; i < A_INST; ++i) {
        glds16(a_src[i], abase + (wave * A_INST + i) * 1024);
        a_src[i] += GEMM_BK;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_INST; ++i) {
        glds16(a2_src[i], abase + (wave * A_INST + i) * 1024);
        a2_src[i] += GEMM_BK;
      }
    }
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
      glds16(b_src[i], bbase + (wave * B_INST + i) * 1024);
      b_src[i] += GEMM_BK;
    }
    ++kissue;
  };

#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wm * 64 + l15) * 128;
  const int b_row_off = (wn * (BN / 2) + l15) * 128;

  issue(0);
  if (STAGES == 3 && nk > 1) issue(1);
  int stage = 0;
  // gemm2w.h's scheme: the last PN = 2 columns of a step's second fragment set are consumed after the NEXT barrier (8 MFMAs of pipe work
  // while the barrier resolves).  FLUX-1024 in-run: single-stream QKV+MLP -1 %, proj_out -2.9 %, +0.6 % end to end.  The full set spills
  // at BN = 256 (-25 %).
  constexpr int PN = 2;
  Pack8 afp[FM], wfp[PN];
  for (int kt = 0; kt < nk; ++kt) {
    if (STAGES == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INST + B_INST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* abase = smem + stage * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;
    Pack8 af0[FM], wf0[FN];
    {
      const int coff = (g ^ swz) * 16;
#pragma unroll
      for (int i = 0; i < FM; ++i) af0[i] = lds_read16(abase + a_row_off + i * 2048 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf0[j] = lds_read16(bbase + b_row_off + j * 2048 + coff);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + STAGES - 1 < nk) {
      int ns = stage + STAGES - 1;
      ns = ns >= STAGES ? ns - STAGES : ns;
      issue(ns);
    }
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
    {
      const int coff = ((4 + g) ^ swz) * 16;
      Pack8 wf[FN - PN];
#pragma unroll
      for (int i = 0; i < FM; ++i) afp[i] = lds_read16(abase + a_row_off + i * 2048 + coff);
#pragma unroll
      for (int j = 0; j < FN - PN; ++j) wf[j] = lds_read16(bbase + b_row_off + j * 2048 + coff);
#pragma unroll
      for (int j = 0; j < PN; ++j) wfp[j] = lds_read16(bbase + b_row_off + (FN - PN + j) * 2048 + coff);
#pragma unroll
      for (int j = 0; j < FN - PN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf[j], afp[i], acc[i][j]);
    }
    stage = stage + 1 >= STAGES ? 0 : stage + 1;
  }
#pragma unroll
  for (int j = 0; j < PN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][FN - PN + j] = mfma16<T>(wfp[j], afp[i], acc[i][FN - PN + j]);
  __syncthreads();
}

// bias / activation / rounding through the LDS C tile, then the row pass (plain store, gated residual, RoPE ...); shared with fgemm8.h
// Generic in the wave geometry: NW = 8 waves as 4 (M) x 2 (N) of 64 x BN/2 (fgemm_kernel, fgemm8_kernel), or NW = 4 waves as 2 x 2 of 128 x BN/2 (gemm4.h: the
// accumulators live in a[0:255] and come back through `acc(i, j)`, i and j integral constants).  acc(i, j)[r] = row 16 i + (lane & 15), column 16 j + 4 (lane >> 4) + r
// of the wave's block.
template <typename T, int BN, int EPI, int NW, typename AccFn>
ECAD_DEV void fgemm_epilogue_g(const FGemmArgs<T>& p, char* smem, int m0, int n0, AccFn&& acc) {
  [[maybe_unused]] constexpr int BM = 256;
  constexpr int WM = NW / 2;                    // waves along M (two along N)
  constexpr int FM = BM / WM / 16, FN = BN / 32;  // wave tile (256 / WM) x BN/2
  constexpr int WROWS = BM / WM;                // rows of a wave's block
  constexpr int RPW = BM / NW;                  // rows a wave streams in the row pass
  constexpr int CS = BN + 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int l15 = lane & 15, g = lane >> 4;
  // ---- direct epilogue of the GELU tiles (round 3, gemm2w.h has the description): FF1 of the joint blocks and the proj_mlp columns of the single blocks' stacked GEMM
  if constexpr (EPI == FEPI_BIAS_GELU || EPI == FEPI_QKV_ROPE_MLP) {
    if (p.epi_direct && (EPI == FEPI_BIAS_GELU || n0 >= p.n_split)) {
      T* cbase = EPI == FEPI_QKV_ROPE_MLP ? p.C2 : p.C;
      const int ld = EPI == FEPI_QKV_ROPE_MLP ? p.ldc2 : p.ldc;
      const int c0 = (EPI == FEPI_QKV_ROPE_MLP ? n0 - p.n_split : n0) + wn * (BN / 2);
      f32x4 bvs[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        bvs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias) bvs[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * (BN / 2) + 16 * j + 4 * g);
      }
      auto cvt = [&](auto ic, auto jc) {
        constexpr int j = decltype(jc)::value;
        const f32x4 av = acc(ic, jc);
        const f32x2 v01 = gelu_tanh2((f32x2){av[0] + bvs[j][0], av[1] + bvs[j][1]});
        const f32x2 v23 = gelu_tanh2((f32x2){av[2] + bvs[j][2], av[3] + bvs[j][3]});
        Pack4 o;
        float_to_elem4<T>(o, 0, v01[0]); float_to_elem4<T>(o, 1, v01[1]); float_to_elem4<T>(o, 2, v23[0]); float_to_elem4<T>(o, 3, v23[1]);
        return o;
      };
      static_assert(FN % 2 == 0, "fragment pairs");
      static_for<FM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int m = m0 + wm * WROWS + 16 * i + l15;
        T* crow = cbase + (size_t)(m < p.M ? m : 0) * ld + c0;
        static_for<FN / 2>([&](auto jh) {
          constexpr int j = 2 * decltype(jh)::value;
          Pack4 a = cvt(ic, std::integral_constant<int, j>{}), b = cvt(ic, std::integral_constant<int, j + 1>{});
          asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a.u[0]), "+v"(b.u[0]));
          asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a.u[1]), "+v"(b.u[1]));
          Pack8 v;
          v.u = (u32x4){a.u[0], a.u[1], b.u[0], b.u[1]};
          if (m < p.M) gstore16(crow + 16 * (j + (g & 1)) + 8 * (g >> 1), v);
        });
      });
      return;
    }
  }
  // ---- epilogue 1: wave's 64 x BN/2 block -> row-major C tile in LDS (bias, activation, rounding)
  T* ctile = reinterpret_cast<T*>(smem);
  static_for<FN>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int nl = wn * (BN / 2) + 16 * j + 4 * g;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + nl);
    static_for<FM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int ml = wm * WROWS + 16 * i + l15;
      const f32x4 av = acc(ic, jc);
      Pack4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = av[r] + bv[r];
        if (EPI == FEPI_BIAS_GELU || (EPI == FEPI_QKV_ROPE_MLP && n0 >= p.n_split)) v = gelu_tanh(v);
        if (EPI == FEPI_BIAS_SILU) v = silu(v);
        float_to_elem4<T>(o, r, v);
      }
      *reinterpret_cast<u32x2*>(ctile + ml * CS + nl) = o.u;
    });
  });
  __syncthreads();

  // ---- epilogue 2: wave w streams rows RPW w .. RPW w + RPW - 1 in 16-byte chunks
  constexpr int CPR = BN / 8;
  constexpr int CHUNKS = RPW * CPR;
  if (EPI == FEPI_BIAS_GATE_RESID) {
    // 64 % CPR == 0: a lane keeps its column chunk for all its rows.  All hidden-state chunks of the wave's 32 rows are requested in
    // one batch (inside a loop each load would wait behind the previous row's store to the same buffer), and the gate of a strip that
    // lies in one sample and segment is fetched once per lane instead of once per chunk.  (The strip = the RPW rows this wave streams.)
    constexpr int NIT = CHUNKS / 64, RSTEP = 64 / CPR;
    const int cchunk = lane % CPR, rl0 = lane / CPR;
    const int n = n0 + cchunk * 8;
    const int mw = m0 + wave * RPW;
    Pack8 hpre[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int m = mw + rl0 + it * RSTEP;
      m = m < p.M ? m : p.M - 1;
      hpre[it] = gload16(p.resid + (size_t)m * p.ldr + n);
    }
    const int m_last = (mw + RPW - 1 < p.M ? mw + RPW - 1 : p.M - 1);
    const int b_first = seg_batch(mw < p.M ? mw : p.M - 1, p.seg0, p.rows_per_batch0, p.rows_per_batch1);
    const bool uni = p.gate && (mw < p.seg0) == (m_last < p.seg0) && b_first == seg_batch(m_last, p.seg0, p.rows_per_batch0, p.rows_per_batch1);
    float gu[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (uni) unpack8<T>(gload16(p.gate + (size_t)b_first * p.gate_stride + n), gu);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int rl = rl0 + it * RSTEP;
      const int ml = wave * RPW + rl;
      const int m = m0 + ml;
      if (m >= p.M) continue;
      Pack8 v = lds_read16(ctile + ml * CS + cchunk * 8);
      if (p.C) gstore16(p.C + (size_t)m * p.ldc + n, v);
      float o[8], h[8];
      unpack8<T>(v, o);
      unpack8<T>(hpre[it], h);
      if (p.gate) {
        float gm[8];
        if (uni) {
#pragma unroll
          for (int e = 0; e < 8; ++e) gm[e] = gu[e];
        } else {
          unpack8<T>(gload16(p.gate + (size_t)seg_batch(m, p.seg0, p.rows_per_batch0, p.rows_per_batch1) * p.gate_stride + n), gm);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] += round_elem<T>(gm[e] * o[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] += o[e];
      }
      if (p.clip16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = clip_half<T>(h[e]);
      }
      gstore16(p.resid + (size_t)m * p.ldr + n, pack8<T>(h));
    }
    return;
  }
  for (int id = lane; id < CHUNKS; id += 64) {
    const int rl = id / CPR, cchunk = id - rl * CPR;
    const int ml = wave * RPW + rl;
    const int m = m0 + ml;
    if (m >= p.M) continue;
    const int n = n0 + cchunk * 8;
    Pack8 v = lds_read16(ctile + ml * CS + cchunk * 8);
    if (EPI == FEPI_BIAS_GATE_RESID) {
      if (p.C) gstore16(p.C + (size_t)m * p.ldc + n, v);
      T* rp = p.resid + (size_t)m * p.ldr + n;
      float o[8], h[8];
      unpack8<T>(v, o);
      unpack8<T>(gload16(rp), h);
      if (p.gate) {
        float gm[8];
        const int b = seg_batch(m, p.seg0, p.rows_per_batch0, p.rows_per_batch1);
        unpack8<T>(gload16(p.gate + (size_t)b * p.gate_stride + n), gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] += round_elem<T>(gm[e] * o[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] += o[e];
      }
      if (p.clip16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = clip_half<T>(h[e]);
      }
      gstore16(rp, pack8<T>(h));
    } else if (EPI == FEPI_QKV_ROPE || EPI == FEPI_QKV_ROPE_MLP) {
      if (EPI == FEPI_QKV_ROPE_MLP && n >= p.n_split) {
        gstore16(p.C2 + (size_t)m * p.ldc2 + (n - p.n_split), v);
      } else {
        if (n < p.rope_cols) {  // 16 consecutive lanes hold the 128 columns of one (row, head)
          const int hc = cchunk & 15;
          float x[8];
          unpack8<T>(v, x);
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
          const float rr = rsqrtf(ss * (1.0f / 128.0f) + p.rope_eps);
          const bool isk = n >= (p.rope_cols >> 1), txt = m < p.seg0;
          const float* w = isk ? (txt && p.nk_txt ? p.nk_txt : p.nk) : (txt && p.nq_txt ? p.nq_txt : p.nq);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + hc * 8), w1 = *reinterpret_cast<const f32x4*>(w + hc * 8 + 4);
          const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = round_elem<T>(round_elem<T>(x[e] * rr) * wv[e]);
          const int pos = txt ? m % p.rows_per_batch0 : p.pos_T + (m - p.seg0) % p.rows_per_batch1;
          const float* cs = p.rope + ((size_t)pos * 64 + hc * 4) * 2;
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs), c1 = *reinterpret_cast<const f32x4*>(cs + 4);
          const float cosv[4] = {c0[0], c0[2], c1[0], c1[2]}, sinv[4] = {c0[1], c0[3], c1[1], c1[3]};
          float o[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[2 * j] = cosv[j] * x[2 * j] - sinv[j] * x[2 * j + 1];
            o[2 * j + 1] = sinv[j] * x[2 * j] + cosv[j] * x[2 * j + 1];
          }
          if (!isk && p.q_scale != 0.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= p.q_scale;
          }
          v = pack8<T>(o);
        }
        gstore16(p.C + (size_t)m * p.ldc + n, v);
      }
    } else {
      gstore16(p.C + (size_t)m * p.ldc + n, v);
    }
  }
}


template <typename T, int BN, int EPI>
ECAD_DEV void fgemm_epilogue(const FGemmArgs<T>& p, char* smem, int m0, int n0, f32x4 (&acc)[4][BN / 32]) {
  fgemm_epilogue_g<T, BN, EPI, 8>(p, smem, m0, n0, [&](auto ic, auto jc) -> f32x4 { return acc[decltype(ic)::value][decltype(jc)::value]; });
}

// tile id -> (tm, tn): blocks b, b + 8 share an XCD, so each XCD gets a contiguous run of ids; 4 x 4 patches inside the run
ECAD_DEV void fgemm_tile_of(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  if ((tiles_m & 3) == 0 && (tiles_n & 3) == 0) {  // the 16 resident tiles of a patch share 4 A panels and 4 W panels
    const int patch = bid >> 4, within = bid & 15;
    const int pn = tiles_n >> 2;
    const int patch_m = patch / pn, patch_n = patch - patch_m * pn;
    tm = patch_m * 4 + (within >> 2);
    tn = patch_n * 4 + (within & 3);
  } else {
    tm = bid / tiles_n;
    tn = bid - tm * tiles_n;
  }
}

template <typename T, int BN, int EPI>
__global__ __launch_bounds__(512) void fgemm_kernel(const FGemmArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  fgemm_tile_of(blockIdx.x, (p.M + 255) / 256, p.N / BN, tm, tn);
  f32x4 acc[4][BN / 32];
  fgemm_mainloop<T, BN>(p, smem, tm * 256, tn * BN, acc);
  fgemm_epilogue<T, BN, EPI>(p, smem, tm * 256, tn * BN, acc);
}

}  // namespace ecad
