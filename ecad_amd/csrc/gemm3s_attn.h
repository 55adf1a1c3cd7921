// gemm3s_attn.h -- attention fused into the tail of the query projection (epilogues EPI_BIAS_ATTN / EPI_BIAS_ATTN_TAIL of gemm3s.h).
//
// A 256 x 144 tile of the Q projection of a PixArt block is exactly one attention work item pair: 256 query rows of ONE sample (N % 256 == 0)
// times the 144 columns of TWO heads (d_h 72).  The stand-alone attention kernels of the alpha-256 shapes are latency chains, not throughput
// kernels (stamped: 26k cycles per (sample, head) item for ~4k cycles of MFMA + softmax work: global loads of Q, K, V -> LDS -> barriers -> ...,
// twice per CU, 25-31 us per launch for 4.5-9.7 GFLOP), and the Q they read was written to HBM by the GEMM a few microseconds earlier.  Here the
// workgroup that has just produced the Q tile keeps it in LDS, stages the K / V rows of its sample and heads, and runs the attention of
// attention128.h (same fragment layouts, same operation order: S^T = K Q^T strips in registers, exp2'd P straight into O^T = V^T P^T, ones column
// for the denominator, running-max rescale across key chunks) for its 32 query rows per wave -- one launch, no Q round trip through HBM.
//   self-attention  (cached_transformer_block.py:231-239 -> Attention.forward [ext]): K, V = columns [D, 3D) of the fused QKV buffer, written by the
//                   wide K,V GEMM that now runs BEFORE this launch; Nk = tokens per sample
//   cross-attention (:282-288): K, V = the hoisted text K/V of the block, Nk = text tokens, additive mask bias (TAIL)
// Output rows go to the attention-output buffer the out-projection reads, as the stand-alone kernel's do.
#pragma once
#include "attention.h"
#include "common.h"
#include "gemm.h"

namespace ecad {

constexpr int G3A_DH = 72, G3A_KC = 128;
constexpr int G3A_Q_BYTES = 256 * 152 * 2;                                   // the C tile of gemm3s.h (CS = 152), i.e. the Q tile
constexpr int G3A_KV_OFF = (G3A_Q_BYTES + 1023) & ~1023;                     // K / V staging behind it
constexpr int G3A_LDS_BYTES = G3A_KV_OFF + attn_lds_bytes<G3A_KC, G3A_DH>();  // Q tile 77824 (on a 1 KiB boundary) + K / V staging + bias strip; gemm3s.h
                                                                              // asserts that this fits the operand ring the launch requests (3 x 51200)

// Called by every thread of the workgroup after the K loop.  `ring_free`: false => the caller has NOT yet synchronised the workgroup after the last
// K-step (other waves may still read operand stage 2).  acc / bvs: the wave's 32 x 144 accumulators and bias vectors (gemm3s.h layout).
// FM = 16-row query fragments per wave (2: the 256-row tile of 8 waves; 1: the 64-row latency tile of 4 waves, gemm3s.h), NW = waves of the workgroup.  Every
// query fragment runs the same operations in the same order whatever FM is, so the two tile sizes give bit-identical attention outputs.
template <int ROWS> constexpr int g3a_kv_off() { return (ROWS * 152 * 2 + 1023) & ~1023; }  // K / V staging behind the [ROWS][152] Q tile
template <int ROWS> constexpr int g3a_lds_bytes() { return g3a_kv_off<ROWS>() + attn_lds_bytes<G3A_KC, G3A_DH>(); }
static_assert(g3a_kv_off<256>() == G3A_KV_OFF && g3a_lds_bytes<256>() == G3A_LDS_BYTES && g3a_lds_bytes<64>() == 19456 + 47616, "Q tile sizes");
template <typename T, bool TAIL, int FM_ = 2, int NW_ = 8>
ECAD_DEV void g3_attention_epilogue(const GemmArgs<T>& p, char* smem, int m0, int tn, const f32x4 (&acc)[FM_][9], const f32x4 (&bvs)[9]) {
  constexpr int DH = G3A_DH, KC = G3A_KC, KS = AttCfg<DH>::KS, VS = AttCfg<DH>::VS, ND = AttCfg<DH>::ND, KSTEPS = AttCfg<DH>::KSTEPS, NCH = AttCfg<DH>::NCH;
  constexpr int NDO = DH / 16 + (DH % 16 ? 1 : 0), CS = 152, FN = 9, FM = FM_, NT = NW_ * 64, WR = 16 * FM_;
  constexpr int MAXL = (KC * NCH * 2 + NT - 1) / NT;  // staging loads per thread and chunk: 5 (512 threads) / 9 (256)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  T* qtile = reinterpret_cast<T*>(smem);
  T* Ks = reinterpret_cast<T*>(smem + g3a_kv_off<NW_ * 16 * FM_>());
  T* Vs = Ks + KC * KS;
  float* bias_s = reinterpret_cast<float*>(Vs + KC * VS);
  const int b = m0 / p.rows_per_batch;  // the tile lies inside one sample (checked on the host)
  const int nkc = (p.att_nk + KC - 1) / KC;
  const T* Kg = p.att_k + (size_t)b * p.att_nk * p.att_ldk;
  const T* Vg = p.att_v + (size_t)b * p.att_nk * p.att_ldk;

  Pack8 stg[MAXL];
  auto load_stage = [&](int hd, int kc) {
    const int h = 2 * tn + hd;
    const T* kbase = Kg + (size_t)kc * KC * p.att_ldk + (size_t)h * DH;
    const T* vbase = Vg + (size_t)kc * KC * p.att_ldk + (size_t)h * DH;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      int id = tid + i * NT;
      // branch-free: lanes without a piece and rows past the end read the chunk's first row (always there) and are dropped / zeroed in store_stage.
      // A load under a lane-dependent branch makes the compiler give up counting: it then waits vmcnt(0) at the next use of ANY earlier load --
      // here the bias vectors the Q tile needs at once, i.e. the K / V latency this prefetch is meant to hide.
      const bool has = id < 2 * KC * NCH;
      const bool isv = id >= KC * NCH;
      id = isv ? id - KC * NCH : id;
      const int key = id / NCH, c = id - key * NCH;
      const bool ok = has && kc * KC + key < p.att_nk;
      // wave-uniform chunk base + a 32-bit per-lane offset (the 64-bit row arithmetic per load was VALU time taken from the softmax: attention128.h)
      stg[i] = gload16((isv ? vbase : kbase) + (ok ? (unsigned)(key * p.att_ldk + c * 8) : 0u));
    }
  };
  // LDS address and key of each of this thread's pieces: computed once, not per chunk (attention128.h: -2.3 % on the 32-chunk Sigma-1024 shape)
  unsigned l_arr[MAXL];
  int k_arr[MAXL];
#pragma unroll
  for (int i = 0; i < MAXL; ++i) {
    int id = tid + i * NT;
    const bool isv = id >= KC * NCH;
    id = isv ? id - KC * NCH : id;
    const int key = id / NCH, c = id - key * NCH;
    k_arr[i] = key;
    l_arr[i] = (unsigned)(g3a_kv_off<NW_ * 16 * FM_>() + (isv ? KC * KS + key * VS + c * 8 : key * KS + c * 8) * (int)sizeof(T));
  }
  auto store_stage = [&](int kc) {
    const int left = p.att_nk - kc * KC;
#pragma unroll
    for (int i = 0; i < MAXL; ++i)
      if (tid + i * NT < 2 * KC * NCH) *reinterpret_cast<u32x4*>(smem + l_arr[i]) = k_arr[i] < left ? stg[i].u : (u32x4){0, 0, 0, 0};
    if (TAIL) {
      for (int i = tid; i < KC; i += NT) {
        const int gk = kc * KC + i;
        float bv = ATT_NEG;
        if (gk < p.att_nk) bv = p.att_bias ? p.att_bias[(size_t)b * p.att_nk + gk] * 1.4426950408889634f : 0.f;
        bias_s[i] = bv;
      }
    }
  };

  load_stage(0, 0);  // in flight while the Q tile is converted
  __syncthreads();   // every wave is out of the K loop: the operand ring is free
  // Q tile: bias, rounding to the storage type (what the stand-alone path writes to and reads back from HBM), row-major [256][152]
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = 16 * j + 4 * g;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ml = wave * WR + 16 * i + l15;
      Pack4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) float_to_elem4<T>(o, r, acc[i][j][r] + bvs[j][r]);
      *reinterpret_cast<u32x2*>(qtile + ml * CS + nl) = o.u;
    }
  }
  // K / V staging area: zero (the padding columns of K and V must be zero), ones column of V (softmax denominator by the PV MFMAs)
  for (int i = tid; i < KC * (KS + VS) / 8; i += NT) reinterpret_cast<u32x4*>(Ks)[i] = (u32x4){0, 0, 0, 0};
  __syncthreads();
  for (int i = tid; i < KC; i += NT) Vs[i * VS + DH] = (T)1.0f;

  for (int hd = 0; hd < 2; ++hd) {
    const int h = 2 * tn + hd;
    f32x4 oacc[FM][ND];
    float m_run[FM];
#pragma unroll
    for (int r = 0; r < FM; ++r) {
      m_run[r] = ATT_NEG;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) oacc[r][nd] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // query fragments of this head from the wave's own rows of the Q tile (B operand: lane (l15, g) holds 8 consecutive d of query row l15)
    Pack8 qfr[FM][KSTEPS];
#pragma unroll
    for (int r = 0; r < FM; ++r)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        qfr[r][ks].u = (u32x4){0, 0, 0, 0};
        if (32 * ks + 8 * g + 8 <= DH) qfr[r][ks] = lds_read16(qtile + (wave * WR + r * 16 + l15) * CS + hd * DH + 32 * ks + 8 * g);
      }
    for (int kc = 0; kc < nkc; ++kc) {
      __syncthreads();  // every wave finished reading the previous chunk (first chunk: the zero fill and the ones column are in place)
      store_stage(kc);
      __syncthreads();
      if (kc + 1 < nkc) load_stage(hd, kc + 1);
      else if (hd == 0) load_stage(1, 0);

#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        const T* Kh = Ks + half * 64 * KS;
        const T* Vh = Vs + half * 64 * VS;
        f32x4 s[FM][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
#pragma unroll
          for (int r = 0; r < FM; ++r) s[r][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            const Pack8 kf = lds_read16(Kh + (16 * f + l15) * KS + 32 * ks + 8 * g);
#pragma unroll
            for (int r = 0; r < FM; ++r) s[r][f] = mfma16<T>(kf, qfr[r][ks], s[r][f]);
          }
        }
        float m_new[FM];
#pragma unroll
        for (int r = 0; r < FM; ++r) {
          float mt = ATT_NEG;
          if (TAIL) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + half * 64 + 16 * f + 4 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                s[r][f][e] = s[r][f][e] * p.att_scale_log2e + bv[e];
                mt = fmaxf(mt, s[r][f][e]);
              }
            }
          } else {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
              for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[r][f][e]);
            mt *= p.att_scale_log2e;
          }
          mt = rows_max(mt);
          m_new[r] = fmaxf(m_run[r], mt);
          const float alpha = fast_exp2(m_run[r] - m_new[r]);
          m_run[r] = m_new[r];
#pragma unroll
          for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int e = 0; e < 4; ++e) oacc[r][nd][e] *= alpha;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          Pack8 pf[FM];
#pragma unroll
          for (int r = 0; r < FM; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float p0, p1;
              if (TAIL) {
                p0 = fast_exp2(s[r][2 * s2][e] - m_new[r]);
                p1 = fast_exp2(s[r][2 * s2 + 1][e] - m_new[r]);
              } else {
                p0 = fast_exp2(__builtin_fmaf(s[r][2 * s2][e], p.att_scale_log2e, -m_new[r]));
                p1 = fast_exp2(__builtin_fmaf(s[r][2 * s2 + 1][e], p.att_scale_log2e, -m_new[r]));
              }
              float_to_elem<T>(pf[r], e, p0);
              float_to_elem<T>(pf[r], 4 + e, p1);
            }
          const int krow = 32 * s2 + 4 * g + (l15 >> 2);
          const int dcol = 4 * (l15 & 3);
#pragma unroll
          for (int nd = 0; nd < ND; ++nd) {
            const Pack4 lo = lds_read_tr16(Vh + krow * VS + 16 * nd + dcol);
            const Pack4 hi = lds_read_tr16(Vh + (krow + 16) * VS + 16 * nd + dcol);
            Pack8 vf;
            vf.u = (u32x4){lo.u[0], lo.u[1], hi.u[0], hi.u[1]};
#pragma unroll
            for (int r = 0; r < FM; ++r) oacc[r][nd] = mfma16<T>(vf, pf[r], oacc[r][nd]);
          }
        }
      }
    }
    // denominator = O^T row DH (the ones column), normalise, round, store the wave's 32 rows of this head
#pragma unroll
    for (int r = 0; r < FM; ++r) {
      constexpr int DL = DH & 15;
      const float den = __shfl(oacc[r][DH >> 4][DL & 3], l15 + 16 * (DL >> 2));
      const float inv = 1.0f / den;
      const int m = m0 + wave * WR + r * 16 + l15;
      if (m < p.M) {
        T* orow = p.att_out + (size_t)m * p.att_ldo + (size_t)h * DH;
#pragma unroll
        for (int nd = 0; nd < NDO; ++nd) {
          if (16 * nd + 4 * g + 4 <= DH) {
            Pack4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) float_to_elem4<T>(o, e, oacc[r][nd][e] * inv);
            *reinterpret_cast<u32x2*>(orow + 16 * nd + 4 * g) = o.u;
          }
        }
      }
    }
  }
}

}  // namespace ecad
