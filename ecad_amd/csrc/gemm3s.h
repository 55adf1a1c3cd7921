// gemm3s.h -- the 256x144-tile GEMM of the PixArt path: one tile per workgroup, 8 mixed waves (each issues its share of the LDS-DMA and
// owns 32 rows x 144 columns), 3-stage ring, one raw barrier per K-step with counted vmcnt.  Serves every N = 1152 shape (Q, both attention
// out-projections, cross-attention Q) and, since its second fragment set is consumed across the K-step barrier, FF2 / caption / the small
// GEMMs as well (FF2: 1.0 PFLOP/s; the persistent producer / consumer kernel of pgemm.h, which it replaced there, is ECAD_GEMM=persistent).
// ECAD_GEMM=classic runs every GEMM on it.
#pragma once
#include "common.h"
#include "gemm.h"

namespace ecad {

constexpr int G3_BM = 256, G3_BN = 144, G3_STAGES = 3, G3_NW = 8;
constexpr int G3_STAGE_BYTES = (G3_BM + G3_BN) * 128;
constexpr int G3_LDS_BYTES = G3_STAGES * G3_STAGE_BYTES;

template <typename T, int EPI, int TAG>
__global__ __launch_bounds__(G3_NW * 64) void gemm3s_kernel(const GemmArgs<T> p) {
  constexpr int BM = G3_BM, BN = G3_BN, NWAVES = G3_NW;
  constexpr int WTM = BM / NWAVES;
  constexpr int FM = WTM / 16, FN = BN / 16;
  constexpr int A_BYTES = BM * 128;
  constexpr int STAGE_BYTES = G3_STAGE_BYTES;
  constexpr int A_INST = BM / 8 / NWAVES;
  constexpr int B_INST_TOTAL = BN / 8;
  constexpr int B_INST_MAX = (B_INST_TOTAL + NWAVES - 1) / NWAVES;
  constexpr int CS = BN + 8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int tiles_n = p.N / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  tile_coords(bid, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const T* a_src[A_INST];
  const T* b_src[B_INST_MAX];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int r = (wave * A_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int row = m0 + r;
    row = row < p.M ? row : p.M - 1;
    a_src[i] = p.A + (size_t)row * p.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < B_INST_MAX; ++i) {
    int inst = wave + i * NWAVES;
    inst = inst < B_INST_TOTAL ? inst : B_INST_TOTAL - 1;
    const int r = inst * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[i] = p.W + (size_t)(n0 + r) * p.ldw + c * 8;
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

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wave * WTM + l15) * 128;
  const int b_row_off = l15 * 128;
  const int nk = p.K / GEMM_BK;
  issue(0);
  if (nk > 1) issue(1);
  // EPI_BIAS_CACHE_RESID: the hidden-state chunks this lane updates in the epilogue are fetched NOW, so their HBM
  // latency hides behind the whole K loop instead of sitting at the tail of a 1-tile-per-CU kernel
  constexpr int CPR = BN / 8;
  constexpr int CHUNKS = WTM * CPR;
  constexpr int NCH = (CHUNKS + 63) / 64;
  Pack8 hpre[EPI == EPI_BIAS_CACHE_RESID ? NCH : 1];
#ifndef G3_NO_RESID_PREFETCH
  if (EPI == EPI_BIAS_CACHE_RESID) {
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int id = lane + it * 64;
      const int rl = id / CPR, cchunk = id - rl * CPR;
      const int m = m0 + wave * WTM + rl;
      hpre[it].u = (u32x4){0, 0, 0, 0};
      if (id < CHUNKS && m < p.M) hpre[it] = gload16(p.resid + (size_t)m * p.ldr + n0 + cchunk * 8);
    }
  }
#endif
  // The gate (scale_shift_table row + adaLN modulation) depends on the column and the sample only.  When the tile lies inside one sample
  // (always for N = 256 / 4096 tokens and 256-row tiles) lanes 0 .. CPR-1 fetch the 8 gate values of their column chunk NOW -- three small
  // loads per wave instead of three per output chunk at the tail -- and the epilogue reads them back from LDS.
  float gpre[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool gate_lds = false;
#ifndef G3_GATE_PER_CHUNK
  if (EPI == EPI_BIAS_CACHE_RESID && p.gate_table) {
    const int m_last = (m0 + BM <= p.M ? m0 + BM : p.M) - 1;
    gate_lds = m0 / p.rows_per_batch == m_last / p.rows_per_batch;
    if (gate_lds && lane < CPR) {
      const int n = n0 + lane * 8;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.gate_table + n);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.gate_table + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gpre[e] = t0[e]; gpre[4 + e] = t1[e]; }
      if (p.gate_mod) {
        float gm[8];
        unpack8<T>(gload16(p.gate_mod + (size_t)(m0 / p.rows_per_batch) * p.gate_mod_stride + n), gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) gpre[e] += gm[e];
      }
    }
  }
#endif
  int stage = 0;
#ifndef G3_NO_XBAR
  // second fragment set of a step consumed after the NEXT step's barrier: the matrix pipe has a full group to chew on while the barrier
  // resolves and the new stage's first fragments travel from LDS (11 more live fragment registers: fits this kernel's 72 accumulators;
  // 244 VGPRs, no spills).  In-run A/B: Q -2.7 %, attn1-out -3.7 %, attn2-out -3.2 %, cross-Q -3.0 %.  -DG3_NO_XBAR: the plain loop below.
  Pack8 afp[FM], wfp[FN];
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* abase = smem + stage * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;
    Pack8 af0[FM], wf0[FN];
    {
      const int coff = (g ^ swz) * 16;
#pragma unroll
      for (int i = 0; i < FM; ++i) af0[i] = lds_read16(abase + a_row_off + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf0[j] = lds_read16(bbase + b_row_off + j * 16 * 128 + coff);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nk) {
      int ns = stage + 2;
      ns = ns >= G3_STAGES ? ns - G3_STAGES : ns;
      issue(ns);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt > 0) {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wfp[j], afp[i], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const int coff = ((4 + g) ^ swz) * 16;
#pragma unroll
      for (int i = 0; i < FM; ++i) afp[i] = lds_read16(abase + a_row_off + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wfp[j] = lds_read16(bbase + b_row_off + j * 16 * 128 + coff);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf0[j], af0[i], acc[i][j]);
    stage = stage + 1 >= G3_STAGES ? 0 : stage + 1;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wfp[j], afp[i], acc[i][j]);
#else
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // A_INST(4) + 2: oldest stage retired for every wave
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* abase = smem + stage * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;
#ifdef G_LATE_READS
    if (kt + 2 < nk) {
      int ns = stage + 2;
      ns = ns >= G3_STAGES ? ns - G3_STAGES : ns;
      issue(ns);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + g) ^ swz) * 16;
      Pack8 af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = lds_read16(abase + a_row_off + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = lds_read16(bbase + b_row_off + j * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf[j], af[i], acc[i][j]);
    }
#else
    // first fragment set requested before the next stage's DMA is issued: its LDS latency hides behind the DMA issues (fgemm.h has the numbers)
    Pack8 af0[FM], wf0[FN];
    {
      const int coff = (g ^ swz) * 16;
#pragma unroll
      for (int i = 0; i < FM; ++i) af0[i] = lds_read16(abase + a_row_off + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf0[j] = lds_read16(bbase + b_row_off + j * 16 * 128 + coff);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nk) {
      int ns = stage + 2;
      ns = ns >= G3_STAGES ? ns - G3_STAGES : ns;
      issue(ns);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf0[j], af0[i], acc[i][j]);
    {
      const int coff = ((4 + g) ^ swz) * 16;
      Pack8 af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = lds_read16(abase + a_row_off + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = lds_read16(bbase + b_row_off + j * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf[j], af[i], acc[i][j]);
    }
#endif
    stage = stage + 1 >= G3_STAGES ? 0 : stage + 1;
  }
#endif
  __syncthreads();
  T* ctile = reinterpret_cast<T*>(smem);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = 16 * j + 4 * g;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + nl);
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
    *reinterpret_cast<f32x4*>(gwave + lane * 8) = (f32x4){gpre[0], gpre[1], gpre[2], gpre[3]};
    *reinterpret_cast<f32x4*>(gwave + lane * 8 + 4) = (f32x4){gpre[4], gpre[5], gpre[6], gpre[7]};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
#ifdef G3_NO_RESID_PREFETCH
      Pack8 hv = gload16(rp);
#else
      Pack8 hv = hpre[it];
#endif
      float o[8], h[8];
      unpack8<T>(v, o);
      unpack8<T>(hv, h);
      if (gate_lds) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gwave + cchunk * 8), g1 = *reinterpret_cast<const f32x4*>(gwave + cchunk * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] += g0[e] * o[e]; h[4 + e] += g1[e] * o[4 + e]; }
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
      gstore16(rp, pack8<T>(h));
    } else {
      gstore16(p.C + (size_t)m * p.ldc + n, v);
    }
  }
}

}  // namespace ecad
