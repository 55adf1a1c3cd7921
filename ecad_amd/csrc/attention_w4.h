// attention_w4.h -- stand-alone attention for head dim 128 (FLUX joint attention over [text | image] rows; replaces torch SDPA behind
// ecad/transformer_blocks/cached_flux_transformer_block.py:62-65,188-192) on the one-wave-per-SIMD structure of cdna_hip_programming.md, "Fused attention prefill":
//
//   * a workgroup = 4 waves = one 256-row query block of one (sample, head); a wave owns 64 query rows (RQ = 2 blocks of 32) and the whole 512-entry register
//     file of its SIMD: O^T accumulators (8 tiles of 32 x 32: 128 registers), the Q fragments (64), two score states S(t), S(t + 1) (2 x 64), packed P, K / V
//     fragment windows.
//   * v_mfma_f32_32x32x16: S^T = K Q^T (lane = one query column, its 32 scores of a 64-key tile in registers: row maximum and row sum are in-lane f32 ops + ONE
//     permlane32 exchange), then O^T += V^T P^T with the score registers converted pairwise as the B operand (k order permuted; V^T fragments come out of
//     ds_read_b64_tr_b16 in the same permuted order).  No ones column: the denominator is f32 adds.
//   * K / V tiles of 64 keys arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) three tiles ahead of their use into rings of four 16-KiB
//     slots each (256-byte rows, 16-byte pieces XOR-swizzled on the DMA *source* side: K by (row & 15) for conflict-free ds_read_b128, V by ((row & 3) << 2) for
//     conflict-free transposed reads), behind a COUNTED s_waitcnt vmcnt and ONE s_barrier per tile; the K / V stream runs on across work items.
//   * software pipeline over tiles, written out in the source and pinned gap by gap with sched_barrier: per tile two phases of 32 MFMAs,
//       phase 1: S(t+1) = K(t+1) Q^T      || finish softmax(t): the remaining exponentials, row sums, P(t) packed      || K fragment reads
//       phase 2: O^T += V(t)^T P(t)^T      || start softmax(t+1): row maximum, rescale decision, the first exponentials || V transposed reads || DMA of tile t+3
//     so the matrix pipe always has work while the VALU runs the softmax of the neighbouring tile.
//   * the accumulator rescale is deferred (ATT_W4_THR, log2 units; cdna_hip_programming.md T13): a row's running maximum is only raised when the tile maximum
//     exceeds it by more than the threshold, so P <= 2^THR instead of <= 1 -- exact in f32 / bf16 / f16 arithmetic up to the usual rounding of P (the exponent
//     moves, the mantissa does not); THR = 0 is the textbook recurrence.  When a rescale is due it happens at the top of the next iteration, after every PV MFMA
//     of the tile before it has been issued and before anything at the new scale is accumulated (T13's hazard).
// Numerics: fp32 scores, exp2 with the scale folded into one fma, P rounded to the storage type for the PV product, fp32 row sums of the UNROUNDED P (as torch's
// flash kernels do), fp32 accumulation, one division at the end.  Not bit-identical to attention128.h (other MFMA shape, other summation order): both are held to
// the fp32 reference in tests/test_gpu_ops.py / test_gpu_flux.py.
#pragma once
#include "attention.h"
#include "attention_w4_asm.h"
#include "common.h"

namespace ecad {

#ifndef ATT_W4_THR
#define ATT_W4_THR 6
#endif

constexpr int W4_DH = 128, W4_KT = 64;                 // head dim, keys per tile
constexpr int W4_SLOT = W4_KT * W4_DH * 2;             // 16 KiB per K or V tile
constexpr int W4_NSLOT = 4;
constexpr int W4_LDS_BYTES = 2 * W4_NSLOT * W4_SLOT;   // 128 KiB
constexpr int W4_AHEAD = 3;                            // tiles in flight / landed ahead of the one being multiplied

template <typename T> ECAD_DEV f32x16 mfma32(const Pack8& a, const Pack8& b, f32x16 c);
template <> ECAD_DEV f32x16 mfma32<f16_t>(const Pack8& a, const Pack8& b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, c, 0, 0, 0); }
template <> ECAD_DEV f32x16 mfma32<bf16_t>(const Pack8& a, const Pack8& b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, c, 0, 0, 0); }

// max over the lane pair {l, l ^ 32} (the two halves of a 32x32 accumulator column)
ECAD_DEV float pair_max(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
ECAD_DEV float pair_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// RQ = 32-row query blocks per wave: 2 (256-row work items) or 1 (128-row items for a partial last round of the chip).
template <typename T, int RQ = 2>
__global__ __launch_bounds__(256) void attn_w4_kernel(const AttnArgs p) {
  constexpr int DH = W4_DH, KT = W4_KT, NKS = DH / 16;   // 8 k-steps of 16 over the head dim
  constexpr int QB = 4 * 32 * RQ;                        // query rows per work item
  constexpr float THR = (float)ATT_W4_THR;
  extern __shared__ __attribute__((aligned(16))) char att_smem[];
  char* const Kring = att_smem;
  char* const Vring = att_smem + W4_NSLOT * W4_SLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;

  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  T* O = reinterpret_cast<T*>(p.out);
  const int q_cnt = p.q_cnt > 0 ? p.q_cnt : p.Nq;
  const int nqb = (q_cnt + QB - 1) / QB;
  const int NT = p.Nk / KT;
  const int items = p.B * p.H * nqb;
  const float c = p.scale_log2e;

  // work items of this workgroup: XCD-contiguous runs (the query blocks of one (sample, head) stream the same K / V: let them meet in one L2), as attention_dma.h
  const int first = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int n_my = first < items ? (items - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int total_tiles = n_my * NT;

  // ---- LDS-DMA: instruction j (0..3) of this wave fills rows 16 wave + 4 j + (lane >> 4) of a tile, 16-byte position lane & 15
  unsigned koff[4], voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 16 * wave + 4 * j + (lane >> 4), pos = lane & 15;
    koff[j] = (unsigned)(((size_t)row * p.ldk + (size_t)((pos ^ (row & 15)) * 8)) * sizeof(T));
    voff[j] = (unsigned)(((size_t)row * p.ldv + (size_t)((pos ^ ((row & 3) << 2)) * 8)) * sizeof(T));
  }
  int is_item = first, is_t = 0, is_g = 0;  // next tile to issue: work item, tile inside it, flat tile counter
  const char* is_kb = nullptr;
  const char* is_vb = nullptr;
  auto issue_bases = [&]() {
    const int bh = is_item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    (void)b;
    is_kb = reinterpret_cast<const char*>(K + (size_t)h * DH);
    is_vb = reinterpret_cast<const char*>(V + (size_t)h * DH);
  };
  auto issue_tile = [&]() {  // wave-uniform control flow; the caller checks is_g < total_tiles
    const int bh = is_item / nqb;
    const int b = bh / p.H;
    const size_t row0 = att_row(b, is_t * KT, p.Nk, p.seg_t, p.seg_bt);
    const char* kb = is_kb + row0 * p.ldk * sizeof(T);
    const char* vb = is_vb + row0 * p.ldv * sizeof(T);
    char* kd = Kring + (is_g & (W4_NSLOT - 1)) * W4_SLOT + wave * 4096;
    char* vd = Vring + (is_g & (W4_NSLOT - 1)) * W4_SLOT + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_raw(kb + koff[j], __builtin_amdgcn_readfirstlane(lds_offset(kd + j * 1024)));
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_raw(vb + voff[j], __builtin_amdgcn_readfirstlane(lds_offset(vd + j * 1024)));
    ++is_g;
    if (++is_t == NT) {
      is_t = 0;
      is_item += gridDim.x;
      if (is_item < items) issue_bases();
    }
  };

  // ---- fragment read addresses (bytes inside a tile)
  // K, A operand of S^T = K Q^T: lane (r, hh) reads key row 32 kb + r, pieces 2 ks + hh, stored at piece ^ (row & 15)
  unsigned kaddr[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) kaddr[ks] = (unsigned)(r * 256 + (((2 * ks + hh) ^ (r & 15)) * 16));
  // V, A operand of O^T = V^T P^T through ds_read_b64_tr_b16: 16-lane group G = lane >> 4 covers keys 4 (G >> 1) + {0..3} (+ 8 for the second read) and head-dim
  // columns 32 db + 16 (G & 1) + {0..15}; lane 4 q + pp of the group supplies row q, columns 4 pp .. 4 pp + 3
  unsigned vaddr[4];
  {
    const int q4 = (lane & 15) >> 2, pp = lane & 3, dhalf = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < 4; ++db) vaddr[db] = (unsigned)((4 * hh + q4) * 256 + (((4 * db + 2 * dhalf + (pp >> 1)) ^ (q4 << 2)) * 16) + 8 * (pp & 1));
  }

  if (n_my > 0) issue_bases();
#pragma unroll 1
  for (int i = 0; i < W4_AHEAD; ++i)
    if (is_g < total_tiles) issue_tile();

  int g = 0;  // flat counter of the tile being multiplied
#pragma unroll 1
  for (int it = 0; it < n_my; ++it) {
    const int item = first + it * (int)gridDim.x;
    const int qb = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int qloc0 = qb * QB + wave * (32 * RQ);

    // ---- Q fragments (B operand): lane (r, hh) holds Q[q0 + 32 rq + r][16 ks + 8 hh ..]
    Pack8 qf[RQ][NKS];
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
      const int qloc = qloc0 + 32 * rq + r, qrow = p.q_off + qloc;
      const bool ok = qloc < q_cnt && qrow < p.Nq;
      const T* qp = Q + att_row(b, ok ? qrow : p.q_off, p.Nq, p.seg_t, p.seg_bt) * p.ldq + (size_t)h * DH + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        qf[rq][ks] = gload16(qp + 16 * ks);
        if (!ok) qf[rq][ks].u = (u32x4){0, 0, 0, 0};
      }
    }
    f32x16 o[RQ][4];
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq)
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[rq][db][e] = 0.f;
    float m_run[RQ], l_run[RQ], mneg[RQ], alpha[RQ];
    bool resc = false;
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) { m_run[rq] = ATT_NEG; l_run[rq] = 0.f; mneg[rq] = 0.f; alpha[rq] = 1.f; }

    f32x16 sA[RQ][2], sB[RQ][2];
    Pack8 pA[RQ][4], pB[RQ][4];

    // Closes an iteration: tiles <= g + 1 have landed (every wave waits for its own pieces, then the barrier) and nobody still reads the slots that the next
    // iteration's DMA will write.  The youngest tile issued (g + 2) may stay in flight: 8 pieces per wave.
    auto wait_landed = [&]() {
      if (is_g - g >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    // DMA of the next tile to issue, one 1-KiB piece at a time (pieces 0..3 K, 4..7 V); wave-uniform control flow
    const char* is_src_k = nullptr;
    const char* is_src_v = nullptr;
    char* is_dst_k = nullptr;
    char* is_dst_v = nullptr;
    bool is_on = false;
    auto issue_prepare = [&]() {
      is_on = is_g < total_tiles;
      if (is_on) {
        const int ib = (is_item / nqb) / p.H;
        const size_t row0 = att_row(ib, is_t * KT, p.Nk, p.seg_t, p.seg_bt);
        is_src_k = is_kb + row0 * p.ldk * sizeof(T);
        is_src_v = is_vb + row0 * p.ldv * sizeof(T);
        is_dst_k = Kring + (is_g & (W4_NSLOT - 1)) * W4_SLOT + wave * 4096;
        is_dst_v = Vring + (is_g & (W4_NSLOT - 1)) * W4_SLOT + wave * 4096;
      }
    };
    auto issue_piece = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (is_on) {
        if constexpr (j < 4) glds16_raw(is_src_k + koff[j], __builtin_amdgcn_readfirstlane(lds_offset(is_dst_k + j * 1024)));
        else glds16_raw(is_src_v + voff[j - 4], __builtin_amdgcn_readfirstlane(lds_offset(is_dst_v + (j - 4) * 1024)));
      }
    };
    auto issue_advance = [&]() {
      if (is_on) {
        ++is_g;
        if (++is_t == NT) {
          is_t = 0;
          is_item += gridDim.x;
          if (is_item < items) issue_bases();
        }
      }
    };

    auto qk_all = [&](f32x16 (&s)[RQ][2], int gt) {  // S(gt) with nothing beside it (pipeline fill at the start of a work item)
      const char* Kt = Kring + (gt & (W4_NSLOT - 1)) * W4_SLOT;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[rq][kb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const Pack8 kf = lds_read16(Kt + kb * 8192 + kaddr[ks]);
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) s[rq][kb] = mfma32<T>(kf, qf[rq][ks], s[rq][kb]);
        }
      }
    };
    float mx_part[RQ];
    // maximum of the 16 scores a lane holds of one 32 x 32 score tile
    auto tile_max = [&](const f32x16& s) {
      float m0 = fmaxf(s[0], s[1]);
#pragma unroll
      for (int e = 2; e < 16; e += 2) m0 = fmaxf(fmaxf(m0, s[e]), s[e + 1]);
      return m0;
    };
    // decision for the tile whose lane maxima are in mx_part: raise a row's running maximum only past the threshold; the accumulators are rescaled later (resc)
    auto decide = [&]() {
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) {
        const float mx = pair_max(mx_part[rq]) * c;
        const bool grow = mx > m_run[rq] + THR;
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {  // wave-uniform: rare once the maxima have settled
          const float m_new = fmaxf(m_run[rq], mx);
          const float a = fast_exp2(m_run[rq] - m_new);  // 1.0 on the rows that did not grow
          alpha[rq] *= a;
          l_run[rq] *= a;                                // everything summed so far is at the old scale
          m_run[rq] = m_new;
          resc = true;
        }
        mneg[rq] = -m_run[rq];
      }
    };
    auto rescale = [&]() {
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) {
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[rq][db][e] *= alpha[rq];
        alpha[rq] = 1.f;
      }
      resc = false;
    };
    // two exponentials: scores 2 IDX, 2 IDX + 1 of tile (RQ_, KB_) -> register IDX & 3 of P fragment 2 KB_ + (IDX >> 2); fp32 row sum of the unrounded values
#define W4_EXP2(S_, KB_, IDX_, PF_, RQ_)                                                                         \
  {                                                                                                              \
    const float e0_ = fast_exp2(__builtin_fmaf(S_[RQ_][KB_][2 * (IDX_)], c, mneg[RQ_]));                         \
    const float e1_ = fast_exp2(__builtin_fmaf(S_[RQ_][KB_][2 * (IDX_) + 1], c, mneg[RQ_]));                     \
    l_run[RQ_] += e0_;                                                                                           \
    l_run[RQ_] += e1_;                                                                                           \
    float_to_elem<T>(PF_[RQ_][2 * (KB_) + ((IDX_) >> 2)], 2 * ((IDX_)&3), e0_);                                  \
    float_to_elem<T>(PF_[RQ_][2 * (KB_) + ((IDX_) >> 2)], 2 * ((IDX_)&3) + 1, e1_);                              \
  }

    // One pipelined iteration: tile gt (scores sc, already through "start softmax": maximum decided, P fragments of key block 0 in pc) is finished and multiplied
    // into O; S(gt + 1) is produced into sn and started (when HAS_NEXT).  A "step" = one K or V fragment = RQ MFMAs + its share of the softmax, pinned by sched_barrier.
    auto iteration = [&](f32x16 (&sc)[RQ][2], Pack8 (&pc)[RQ][4], f32x16 (&sn)[RQ][2], Pack8 (&pn)[RQ][4], int gt, auto has_next_c) {
      constexpr bool HAS_NEXT = decltype(has_next_c)::value;
      const char* Kt = Kring + ((gt + 1) & (W4_NSLOT - 1)) * W4_SLOT;
      const char* Vt = Vring + (gt & (W4_NSLOT - 1)) * W4_SLOT;
      if (resc) rescale();
      issue_prepare();
      // ---------------- phase 1: S(gt+1) = K(gt+1) Q^T || finish softmax(gt): key block 1 -> P fragments 2, 3 || partial maximum of S(gt+1) key block 0
      Pack8 kf[3];
      if constexpr (HAS_NEXT) {
        kf[0] = lds_read16(Kt + kaddr[0]);
        kf[1] = lds_read16(Kt + kaddr[1]);
      }
      static_for<2 * NKS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;  // K fragment: key block i / 8, k-step i % 8
        constexpr int kb = i / NKS, ks = i % NKS;
        if constexpr (HAS_NEXT) {
          if constexpr (i + 2 < 2 * NKS) kf[(i + 2) % 3] = lds_read16(Kt + ((i + 2) / NKS) * 8192 + kaddr[(i + 2) % NKS]);
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) {
            if constexpr (ks == 0) {
              f32x16 z;
#pragma unroll
              for (int e = 0; e < 16; ++e) z[e] = 0.f;
              sn[rq][kb] = mfma32<T>(kf[i % 3], qf[rq][ks], z);
            } else {
              sn[rq][kb] = mfma32<T>(kf[i % 3], qf[rq][ks], sn[rq][kb]);
            }
          }
        }
        if constexpr (i / 8 < RQ) W4_EXP2(sc, 1, i % 8, pc, i / 8)
        if constexpr (HAS_NEXT && i >= 10 && i - 10 < RQ) mx_part[i - 10] = tile_max(sn[i - 10][0]);
        __builtin_amdgcn_sched_barrier(0);
      });
      // ---------------- phase 2: O^T += V(gt)^T P(gt)^T || start softmax(gt+1): maximum, decision, key block 0 -> P fragments 0, 1 || DMA of tile gt + 3
      Pack4 vlo[3], vhi[3];
      vlo[0] = lds_read_tr16(Vt + vaddr[0]);
      vhi[0] = lds_read_tr16(Vt + 2048 + vaddr[0]);
      vlo[1] = lds_read_tr16(Vt + 4096 + vaddr[0]);
      vhi[1] = lds_read_tr16(Vt + 4096 + 2048 + vaddr[0]);
      static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;  // V fragment: head-dim block i / 4, k-step i % 4
        constexpr int db = i / 4, kst = i % 4;
        if constexpr (i + 2 < 16) {
          vlo[(i + 2) % 3] = lds_read_tr16(Vt + ((i + 2) % 4) * 4096 + vaddr[(i + 2) / 4]);
          vhi[(i + 2) % 3] = lds_read_tr16(Vt + ((i + 2) % 4) * 4096 + 2048 + vaddr[(i + 2) / 4]);
        }
        Pack8 vf;
        vf.u = (u32x4){vlo[i % 3].u[0], vlo[i % 3].u[1], vhi[i % 3].u[0], vhi[i % 3].u[1]};
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) o[rq][db] = mfma32<T>(vf, pc[rq][kst], o[rq][db]);
        if constexpr (HAS_NEXT) {
          if constexpr (i < RQ) mx_part[i] = fmaxf(mx_part[i], tile_max(sn[i][1]));
          if constexpr (i >= 2) {
            constexpr int pr = i - 2;  // pairs 0..13 of the 8 RQ of key block 0; the last two ride in steps 9 and 15
            if constexpr (pr / 8 < RQ) W4_EXP2(sn, 0, pr % 8, pn, pr / 8)
            if constexpr (i == 9 && RQ == 2) W4_EXP2(sn, 0, 6, pn, 1)
            if constexpr (i == 15 && RQ == 2) W4_EXP2(sn, 0, 7, pn, 1)
          }
        }
        if constexpr (i >= 4 && i < 12) issue_piece(std::integral_constant<int, i - 4>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HAS_NEXT && i == 1) {
          decide();
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      issue_advance();
    };

    // ---- pipeline fill: S(0), its maximum, the exponentials of key block 0
    if (it == 0) wait_landed();
    qk_all(sA, g);
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) mx_part[rq] = fmaxf(tile_max(sA[rq][0]), tile_max(sA[rq][1]));
    decide();
    static_for<8 * RQ>([&](auto ic) {
      constexpr int pr = decltype(ic)::value;
      W4_EXP2(sA, 0, pr % 8, pA, pr / 8)
    });
    // ---- tiles in pairs (A -> B, B -> A); NT is even (Nk % 128 == 0 is the dispatch condition)
#pragma unroll 1
    for (int t = 0; t + 2 < NT; t += 2) {
      iteration(sA, pA, sB, pB, g, std::true_type{});
      ++g;
      wait_landed();
      iteration(sB, pB, sA, pA, g, std::true_type{});
      ++g;
      wait_landed();
    }
    iteration(sA, pA, sB, pB, g, std::true_type{});
    ++g;
    wait_landed();
    iteration(sB, pB, sA, pA, g, std::false_type{});
    ++g;
    wait_landed();
    if (resc) rescale();
#undef W4_EXP2

    // ---- finish: denominators of both lane halves, normalise, store 4 consecutive head-dim columns per lane and register group
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
      const float inv = 1.0f / pair_sum(l_run[rq]);
      const int qloc = qloc0 + 32 * rq + r, qrow = p.q_off + qloc;
      if (qloc < q_cnt && qrow < p.Nq) {
        T* orow = O + att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldo + (size_t)h * DH + 4 * hh;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            Pack4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) float_to_elem4<T>(ov, e, o[rq][db][4 * i4 + e] * inv);
            *reinterpret_cast<u32x2*>(orow + 32 * db + 8 * i4) = ov.u;
          }
      }
    }
  }
}

// ---- the product kernel: the same algorithm with the per-work-item body as ONE hand-placed asm statement (attention_w4_asm.h, generated by
// scripts/gen_attn_w4.py, which holds the register map and the schedule).  The C++ here walks the work items and computes addresses; attn_w4_kernel above is the
// readable statement of the algorithm (and the compiler-scheduled baseline the asm is measured against in scripts/probe_attn_w4.hip).
// Differences of the asm body from attn_w4_kernel: Q is prescaled by scale * log2 e and rounded to the storage type once per work item; the running maximum enters
// the scores as the C operand of a tile's first MFMA (C-init registers), so an exponential is ONE instruction per score; the maximum of tile t + 1 is checked
// before any of its exponentials and, if it exceeds the running maximum by more than 2^8, every live score register and C-init are recentred (rare path), the
// accumulators follow at the end of the iteration.
// Shapes: d_h 128, Nk % 256 == 0, whole 256-row query blocks, no bias; the joint [text | image] layout needs seg_t % 256 == 0 (flux.hip dispatches).
template <typename T> ECAD_DEV unsigned w4_scale_operand(float c);
template <> ECAD_DEV unsigned w4_scale_operand<bf16_t>(float c) { return __builtin_bit_cast(unsigned, c); }
template <> ECAD_DEV unsigned w4_scale_operand<f16_t>(float c) {
  const unsigned short hb = __builtin_bit_cast(unsigned short, (f16_t)c);
  return (unsigned)hb | ((unsigned)hb << 16);
}

// RQ = 32-row query blocks per wave: 2 (256-row work items) or 1 (128-row items, for a partial last round of the chip: half the MFMAs per item on the same K / V
// stream).  A row's arithmetic does not depend on RQ, on its neighbours in the wave or on how a launch is cut into items: every output column of an MFMA is its own
// sum, and a recentre that another row of the wave triggers subtracts 0 and multiplies by 1 on this one.
template <typename T, int RQ = 2>
__global__ __launch_bounds__(256) void attn_w4a_kernel(const AttnArgs p) {
  constexpr int DH = W4_DH, KT = W4_KT, QB = 128 * RQ;
  extern __shared__ __attribute__((aligned(16))) char att_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const char* Q = reinterpret_cast<const char*>(p.q);
  const char* K = reinterpret_cast<const char*>(p.k);
  const char* V = reinterpret_cast<const char*>(p.v);
  char* O = reinterpret_cast<char*>(p.out);
  const int q_cnt = p.q_cnt > 0 ? p.q_cnt : p.Nq;
  const int nqb = q_cnt / QB;
  const int NT = p.Nk / KT;
  const int items = p.B * p.H * nqb;
  const int first = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int n_my = first < items ? (items - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const unsigned rsk = (unsigned)(p.ldk * sizeof(T)), rsv = (unsigned)(p.ldv * sizeof(T));
  const unsigned tsk = rsk * KT, tsv = rsv * KT;
  const unsigned nt_text = p.seg_t > 0 ? (unsigned)(p.seg_t / KT) : 0u;
  const unsigned lds = __builtin_amdgcn_readfirstlane(lds_offset(att_smem));
  const unsigned cq = w4_scale_operand<T>(p.scale_log2e);
  auto tile_k = [&](int item, int t) { const int bh = item / nqb; return K + ((size_t)att_row(bh / p.H, t * KT, p.Nk, p.seg_t, p.seg_bt) * p.ldk + (size_t)(bh % p.H) * DH) * sizeof(T); };
  auto tile_v = [&](int item, int t) { const int bh = item / nqb; return V + ((size_t)att_row(bh / p.H, t * KT, p.Nk, p.seg_t, p.seg_bt) * p.ldv + (size_t)(bh % p.H) * DH) * sizeof(T); };

  if (n_my > 0) {  // tiles 0..2 of the first work item; from then on the asm body keeps the stream three tiles ahead, across work items
#pragma unroll 1
    for (int t = 0; t < W4_AHEAD; ++t) {
      const char* kb = tile_k(first, t);
      const char* vb = tile_v(first, t);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 16 * wave + 4 * j + (lane >> 4), pos = lane & 15;
        glds16_raw(kb + (size_t)row * rsk + (size_t)((pos ^ (row & 15)) * 16), __builtin_amdgcn_readfirstlane(lds + t * W4_SLOT + wave * 4096 + j * 1024));
        glds16_raw(vb + (size_t)row * rsv + (size_t)((pos ^ ((row & 3) << 2)) * 16), __builtin_amdgcn_readfirstlane(lds + W4_NSLOT * W4_SLOT + t * W4_SLOT + wave * 4096 + j * 1024));
      }
    }
  }
#pragma unroll 1
  for (int it = 0; it < n_my; ++it) {
    const int item = first + it * (int)gridDim.x;
    const int qb = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int nxt = it + 1 < n_my ? item + (int)gridDim.x : item;  // past the last work item the stream re-reads this one's first tiles into dead ring slots
    const char* kp = tile_k(item, W4_AHEAD);
    const char* vp = tile_v(item, W4_AHEAD);
    const char* kimg = tile_k(item, (int)nt_text);
    const char* vimg = tile_v(item, (int)nt_text);
    const char* knext = tile_k(nxt, 0);
    const char* vnext = tile_v(nxt, 0);
    const int qrow = p.q_off + qb * QB + wave * (32 * RQ) + r;
    const int qrow1 = RQ == 2 ? qrow + 32 : qrow;
    const char* q0 = Q + ((size_t)att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldq + (size_t)h * DH + 8 * hh) * sizeof(T);
    const char* q1 = Q + ((size_t)att_row(b, qrow1, p.Nq, p.seg_t, p.seg_bt) * p.ldq + (size_t)h * DH + 8 * hh) * sizeof(T);
    char* o0 = O + ((size_t)att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldo + (size_t)h * DH + 4 * hh) * sizeof(T);
    char* o1 = O + ((size_t)att_row(b, qrow1, p.Nq, p.seg_t, p.seg_bt) * p.ldo + (size_t)h * DH + 4 * hh) * sizeof(T);
    const unsigned nt = (unsigned)NT, trips = (unsigned)(NT / 4);
    if constexpr (std::is_same<T, bf16_t>::value && RQ == 2) {
      W4_ASM_BF16_2(kp, vp, kimg, vimg, knext, vnext, tsk, tsv, rsk, rsv, nt_text, nt, trips, lds, wave, cq, q0, q1, o0, o1);
    } else if constexpr (std::is_same<T, bf16_t>::value) {
      W4_ASM_BF16_1(kp, vp, kimg, vimg, knext, vnext, tsk, tsv, rsk, rsv, nt_text, nt, trips, lds, wave, cq, q0, q1, o0, o1);
    } else if constexpr (RQ == 2) {
      W4_ASM_F16_2(kp, vp, kimg, vimg, knext, vnext, tsk, tsv, rsk, rsv, nt_text, nt, trips, lds, wave, cq, q0, q1, o0, o1);
    } else {
      W4_ASM_F16_1(kp, vp, kimg, vimg, knext, vnext, tsk, tsv, rsk, rsv, nt_text, nt, trips, lds, wave, cq, q0, q1, o0, o1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

}  // namespace ecad
