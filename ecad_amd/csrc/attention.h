// attention.h -- multi-head attention core for the PixArt block: self-attention over the image tokens and
// cross-attention over the (masked) text tokens.
//
//   out[b, q, h, :] = softmax_k( Q[b,q,h,:] . K[b,k,h,:] / sqrt(dh) + bias[b,k] ) @ V[b,k,h,:]
//
// Replaces torch SDPA inside diffusers' AttnProcessor2_0 (reference call site:
// ecad/transformer_blocks/cached_transformer_block.py:348-353; the additive -10000 bias comes from
// ecad/transformer_2d_models/pixart_transformer_2d_edited.py:255-291).
//
// Structure (gfx950): persistent workgroups (8 waves) walk work items (batch, head, 256-query block).  The K and V
// rows of one item's key chunk (KC = 256 keys for self-attention, 128 for the <= 120/300-token text) are staged
// ONCE in LDS (head dim 72 zero-padded to 96 for K / 80 for V, in LDS only) and shared by all 8 waves; the next
// chunk is prefetched into registers while the current one is consumed.  Each wave owns 16 query rows at a time:
//   S^T = K . Q^T (MFMA 16x16x32) for ALL KC keys stays in registers, so the softmax is ONE max / exp2 pass per
//   chunk -- the lane owns one query column (q = lane & 15), so row max = in-lane ops + 2 cross-lane steps;
//   exp2'd P packs straight into the B operand of O^T = V^T . P^T (k order permuted; V^T is read with
//   ds_read_b64_tr_b16 in the same permuted order); column 72 of the padded V tile is all ones, so the softmax
//   denominator is accumulated by the same MFMAs (O^T row 72) -- no VALU row sums;
//   chunks are combined with the usual running-max rescale (only matters for N > KC, i.e. 1024^2 images).
//
// DHC selects the head-dim class: 72 (PixArt: K padded to 96, V to 80) or 128 (FLUX joint attention over
// [text | image] rows, cached_flux_transformer_block.py:178-182 / 63-66: K 128 (+8 pad), V 128 + ones column -> 144).
// seg_t > 0 switches the row addressing to the two-segment joint layout of flux.hip: sequence position r of sample b
// lives at row b*seg_t + r (text, r < seg_t) or seg_bt + b*(N - seg_t) + (r - seg_t) (image).
#pragma once
#include "common.h"

namespace ecad {

struct AttnArgs {
  const void* q;
  const void* k;
  const void* v;
  void* out;
  const float* bias;  // [B, Nk] additive, or null
  int64_t ldq, ldk, ldv, ldo;  // row strides in elements
  int B, H, Nq, Nk, dh;
  float scale_log2e;  // (1/sqrt(dh)) * log2(e)
  int seg_t, seg_bt;  // 0, 0: plain [B, N] rows; else the joint [text | image] layout (same for q, k, v, out)
  int q_off, q_cnt;   // attention_dma.h: this launch covers query rows [q_off, q_off + q_cnt) of every (sample, head); q_cnt 0 = all Nq rows
  // The two conventions of the Q columns of a qkv buffer (flux.hip): 0 = plain Q, the kernel multiplies the scores by scale_log2e; 1 = whoever produced Q (RoPE
  // epilogue / kernel, FGemmArgs::q_scale) already multiplied it by scale * log2(e) and scale_log2e must be 1.  Launchers check the pairing: a debug dump, a parity
  // hook on the post-RoPE q or a cached-Q path that reads the buffer has to know which one it holds.
  int q_prescaled;
};

template <int DHC> struct AttCfg;
template <> struct AttCfg<72> { static constexpr int KSTEPS = 3, ND = 5, KS = 104, VS = 80, NCH = 9; };    // K: 96 padded + 8; V: 80 (col 72 = ones)
template <> struct AttCfg<128> { static constexpr int KSTEPS = 4, ND = 9, KS = 136, VS = 144, NCH = 16; };  // K: 128 + 8; V: 144 (col 128 = ones): 288 B rows tile the banks
constexpr int ATT_QB = 256;       // query rows per work item (8 waves x 16 rows x 2 rounds)
constexpr float ATT_NEG = -1.0e30f;

template <int KC, int DHC = 72> constexpr int attn_lds_bytes() { return KC * (AttCfg<DHC>::KS + AttCfg<DHC>::VS) * 2 + KC * 4; }

ECAD_DEV size_t att_row(int b, int r, int n, int seg_t, int seg_bt) {
  if (seg_t == 0) return (size_t)b * n + r;
  return r < seg_t ? (size_t)b * seg_t + r : (size_t)seg_bt + (size_t)b * (n - seg_t) + (r - seg_t);
}

template <typename T, int KC, bool MASKED, int DHC = 72>
__global__ __launch_bounds__(512) void attn_kernel(const AttnArgs p) {
  constexpr int ATT_KS = AttCfg<DHC>::KS, ATT_VS = AttCfg<DHC>::VS, KSTEPS = AttCfg<DHC>::KSTEPS, ND = AttCfg<DHC>::ND;
  extern __shared__ __attribute__((aligned(16))) char att_smem[];
  T* Ks = reinterpret_cast<T*>(att_smem);
  T* Vs = Ks + KC * ATT_KS;
  float* bias_s = reinterpret_cast<float*>(Vs + KC * ATT_VS);

  constexpr int NF = KC / 16;   // key fragments per chunk
  constexpr int NS2 = KC / 32;  // 32-key steps for P.V
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int dh = p.dh;
  const int nch = dh >> 3;            // 16-byte chunks per head row (9 for dh = 72)
  const int ch_per_tile = KC * nch;   // chunks in the K (or V) tile
  constexpr int MAXL = (KC * AttCfg<DHC>::NCH * 2 + 511) / 512;  // staging loads per thread

  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  T* O = reinterpret_cast<T*>(p.out);

  const int nqb = (p.Nq + ATT_QB - 1) / ATT_QB;
  const int nkc = (p.Nk + KC - 1) / KC;
  const int items = p.B * p.H * nqb;

  // zero the tiles once (pad columns stay zero), then the ones column of V (softmax denominator rides the P.V MFMAs)
  for (int i = tid; i < KC * (ATT_KS + ATT_VS) / 8; i += 512) reinterpret_cast<u32x4*>(att_smem)[i] = (u32x4){0, 0, 0, 0};
  __syncthreads();
  for (int i = tid; i < KC; i += 512) Vs[i * ATT_VS + dh] = (T)1.0f;

  // ---- register staging of one (item, chunk): ids over [K chunks | V chunks]
  Pack8 stg[MAXL];
  auto load_stage = [&](int item, int kc) {
    const int qb = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    (void)qb;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      int id = tid + i * 512;
      stg[i].u = (u32x4){0, 0, 0, 0};
      if (id < 2 * ch_per_tile) {
        const bool isv = id >= ch_per_tile;
        id = isv ? id - ch_per_tile : id;
        const int key = id / nch, c = id - key * nch;
        const int gk = kc * KC + key;
        if (gk < p.Nk) {
          const size_t grow = att_row(b, gk, p.Nk, p.seg_t, p.seg_bt);
          const T* base = isv ? V + grow * p.ldv : K + grow * p.ldk;
          stg[i] = gload16(base + (size_t)h * dh + c * 8);
        }
      }
    }
  };
  auto store_stage = [&](int item, int kc) {
    const int bh = item / nqb;
    const int b = bh / p.H;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      int id = tid + i * 512;
      if (id < 2 * ch_per_tile) {
        const bool isv = id >= ch_per_tile;
        id = isv ? id - ch_per_tile : id;
        const int key = id / nch, c = id - key * nch;
        T* dst = isv ? Vs + key * ATT_VS + c * 8 : Ks + key * ATT_KS + c * 8;
        *reinterpret_cast<u32x4*>(dst) = stg[i].u;
      }
    }
    if (MASKED) {
      for (int i = tid; i < KC; i += 512) {
        const int gk = kc * KC + i;
        float bv = ATT_NEG;
        if (gk < p.Nk) bv = p.bias ? p.bias[(size_t)b * p.Nk + gk] * 1.4426950408889634f : 0.f;
        bias_s[i] = bv;
      }
    }
  };

  int item = blockIdx.x;
  if (item < items) load_stage(item, 0);

  for (; item < items; item += gridDim.x) {
    const int qb = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    f32x4 oacc[2][ND];
    float m_run[2] = {ATT_NEG, ATT_NEG};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) oacc[r][nd] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Q fragments of both rounds (B operand of S^T = K Q^T): lane holds Q[qrow][32*ks + 8g + j]; issued before the
    // staging barriers so their latency hides behind the LDS fill
    Pack8 qfr[2][KSTEPS];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int qrow = qb * ATT_QB + r * 128 + wave * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int d = 32 * ks + 8 * g;
        qfr[r][ks].u = (u32x4){0, 0, 0, 0};
        if (qrow < p.Nq && d + 8 <= dh) qfr[r][ks] = gload16(Q + att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldq + (size_t)h * dh + d);
      }
    }
    for (int kc = 0; kc < nkc; ++kc) {
      __syncthreads();  // every wave finished reading the previous chunk
      store_stage(item, kc);
      __syncthreads();
      // prefetch the next chunk (or the next item's first chunk) while this one is consumed
      if (kc + 1 < nkc) load_stage(item, kc + 1);
      else if (item + (int)gridDim.x < items) load_stage(item + gridDim.x, 0);

      auto round = [&](const Pack8 (&qf)[KSTEPS], f32x4 (&oa)[ND], float& m_r) {
        // ---- S^T strip: NF key fragments x 16 queries, all in registers
        f32x4 s[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            Pack8 kf = lds_read16(Ks + (16 * f + l15) * ATT_KS + 32 * ks + 8 * g);
            s[f] = mfma16<T>(kf, qf[ks], s[f]);
          }
        }
        float mt = ATT_NEG;
        if (MASKED) {
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + 16 * f + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s[f][e] = s[f][e] * p.scale_log2e + bv[e];
              mt = fmaxf(mt, s[f][e]);
            }
          }
        } else {
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[f][e]);
          mt *= p.scale_log2e;  // scale > 0: max commutes with it
        }
        mt = rows_max(mt);
        const float m_new = fmaxf(m_r, mt);
        const float alpha = fast_exp2(m_r - m_new);
        m_r = m_new;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
          for (int e = 0; e < 4; ++e) oa[nd][e] *= alpha;

        // ---- O^T += V^T P^T.  k slot (g, j) of 32-key step s2 <-> key 32*s2 + (j < 4 ? 4g + j : 16 + 4g + j - 4)
#pragma unroll
        for (int s2 = 0; s2 < NS2; ++s2) {
          Pack8 pf;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float p0, p1;
            if (MASKED) {
              p0 = fast_exp2(s[2 * s2][e] - m_new);
              p1 = fast_exp2(s[2 * s2 + 1][e] - m_new);
            } else {
              p0 = fast_exp2(s[2 * s2][e] * p.scale_log2e - m_new);
              p1 = fast_exp2(s[2 * s2 + 1][e] * p.scale_log2e - m_new);
            }
            float_to_elem<T>(pf, e, p0);
            float_to_elem<T>(pf, 4 + e, p1);
          }
          const int krow = 32 * s2 + 4 * g + (l15 >> 2);
          const int dcol = 4 * (l15 & 3);
#pragma unroll
          for (int nd = 0; nd < ND; ++nd) {
            Pack4 lo = lds_read_tr16(Vs + krow * ATT_VS + 16 * nd + dcol);
            Pack4 hi = lds_read_tr16(Vs + (krow + 16) * ATT_VS + 16 * nd + dcol);
            Pack8 vf;
            vf.u = (u32x4){lo.u[0], lo.u[1], hi.u[0], hi.u[1]};
            oa[nd] = mfma16<T>(vf, pf, oa[nd]);
          }
        }
      };
      round(qfr[0], oacc[0], m_run[0]);
      __builtin_amdgcn_sched_barrier(0);  // rounds strictly sequential: interleaving them spills (2 S strips + staging)
      round(qfr[1], oacc[1], m_run[1]);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- finish the item: denominator = O^T row dh (ones column), normalise, store 4 consecutive d per lane
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int dl = dh & 15;  // column dh sits in d-fragment dh/16 at (g = dl/4, e = dl%4)
      float den = 1.f;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (nd == (dh >> 4) && e == (dl & 3)) den = oacc[r][nd][e];  // static indices only: no scratch
      den = __shfl(den, l15 + 16 * (dl >> 2));  // valid on the lanes with g == dl / 4
      const float inv = 1.0f / den;
      const int qrow = qb * ATT_QB + r * 128 + wave * 16 + l15;
      if (qrow < p.Nq) {
        T* orow = O + att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldo + (size_t)h * dh;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) {
          const int d = 16 * nd + 4 * g;
          if (d + 4 <= dh) {
            Pack4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) float_to_elem4<T>(o, e, oacc[r][nd][e] * inv);
            *reinterpret_cast<u32x2*>(orow + d) = o.u;
          }
        }
      }
    }
  }
}

}  // namespace ecad
