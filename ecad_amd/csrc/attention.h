// attention.h -- multi-head attention core for the PixArt block: self-attention over the image tokens and
// cross-attention over the (masked) text tokens.
//
//   out[b, q, h, :] = softmax_k( Q[b,q,h,:] . K[b,k,h,:] / sqrt(dh) + bias[b,k] ) @ V[b,k,h,:]
//
// Replaces torch SDPA inside diffusers' AttnProcessor2_0 (reference call site:
// ecad/transformer_blocks/cached_transformer_block.py:348-353; the additive -10000 bias comes from
// ecad/transformer_2d_models/pixart_transformer_2d_edited.py:255-291).
//
// Structure (gfx950, flash-style so N = 4096 (1024^2) works as well as N = 256):
//   workgroup = 4 waves = 64 query rows of one (batch, head); loop over 64-key tiles staged in LDS.
//   S^T = K . Q^T on MFMA 16x16x32 with the head dim (72) zero-padded to 96 in LDS only, so each lane owns
//   ONE query column (q = lane & 15) and 16 keys per tile: row max / sum are 15 in-lane ops + 2 cross-lane
//   steps, and exp2'd P packs straight into the B operand of O^T = V^T . P^T (k order permuted; V^T is read
//   with ds_read_b64_tr_b16 in the same permuted order).  O^T accumulators have q on the lane as well, so the
//   online-softmax rescale needs no shuffles.  head dim padded to 80 for PV (5 d-fragments).
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
};

constexpr int ATT_KT = 64;        // keys per tile
constexpr int ATT_KS = 104;       // K tile row stride (elements): 96 padded head dim + 8
constexpr int ATT_VS = 80;        // V tile row stride (elements): head dim padded to 80
constexpr int ATT_MAXCH = 12;     // max 16-byte chunks per row supported (dh <= 96)
constexpr float ATT_NEG = -1.0e30f;

template <typename T, bool MASKED>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) T Ks[ATT_KT * ATT_KS];
  __shared__ __attribute__((aligned(16))) T Vs[ATT_KT * ATT_VS];
  __shared__ __attribute__((aligned(16))) float bias_s[ATT_KT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int dh = p.dh;
  const int nch = dh >> 3;  // 16-byte chunks per head row (9 for dh = 72)

  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  T* O = reinterpret_cast<T*>(p.out);

  // zero both tiles once: pad columns are never overwritten afterwards
  for (int i = tid; i < ATT_KT * ATT_KS / 8; i += 256) reinterpret_cast<u32x4*>(Ks)[i] = (u32x4){0, 0, 0, 0};
  for (int i = tid; i < ATT_KT * ATT_VS / 8; i += 256) reinterpret_cast<u32x4*>(Vs)[i] = (u32x4){0, 0, 0, 0};

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + l15][32*ks + 8g + j]
  Pack8 qf[3];
  {
    const int qrow = q0 + l15;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int d = 32 * ks + 8 * g;
      qf[ks].u = (u32x4){0, 0, 0, 0};
      if (qrow < p.Nq && d + 8 <= dh) qf[ks] = gload16(Q + ((size_t)b * p.Nq + qrow) * p.ldq + (size_t)h * dh + d);
    }
  }

  // ---- register staging of the K/V tiles: chunk ids tid, tid+256, tid+512 over 64 keys x nch chunks
  const int total_ch = ATT_KT * nch;
  Pack8 kreg[3], vreg[3];
  auto load_tile = [&](int key0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = tid + i * 256;
      kreg[i].u = (u32x4){0, 0, 0, 0};
      vreg[i].u = (u32x4){0, 0, 0, 0};
      if (id < total_ch) {
        const int key = id / nch, c = id - key * nch;
        if (key0 + key < p.Nk) {
          const size_t row = (size_t)b * p.Nk + key0 + key;
          kreg[i] = gload16(K + row * p.ldk + (size_t)h * dh + c * 8);
          vreg[i] = gload16(V + row * p.ldv + (size_t)h * dh + c * 8);
        }
      }
    }
  };
  auto store_tile = [&](int key0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = tid + i * 256;
      if (id < total_ch) {
        const int key = id / nch, c = id - key * nch;
        *reinterpret_cast<u32x4*>(Ks + key * ATT_KS + c * 8) = kreg[i].u;
        *reinterpret_cast<u32x4*>(Vs + key * ATT_VS + c * 8) = vreg[i].u;
      }
    }
    if (MASKED && tid < ATT_KT) {
      const int key = key0 + tid;
      float bv = ATT_NEG;
      if (key < p.Nk) bv = p.bias ? p.bias[(size_t)b * p.Nk + key] * 1.4426950408889634f : 0.f;
      bias_s[tid] = bv;
    }
  };

  f32x4 oacc[5];
#pragma unroll
  for (int nd = 0; nd < 5; ++nd) oacc[nd] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = ATT_NEG, l_run = 0.f;  // running max (log2 units) and this lane's partial row sum

  const int ntiles = (p.Nk + ATT_KT - 1) / ATT_KT;
  load_tile(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();  // everyone finished reading tile t-1 (and the zero fill on t = 0)
    store_tile(t * ATT_KT);
    __syncthreads();
    if (t + 1 < ntiles) load_tile((t + 1) * ATT_KT);

    // ---- S^T tile: 4 key fragments x 16 queries
    f32x4 s[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        Pack8 kf = lds_read16(Ks + (16 * f + l15) * ATT_KS + 32 * ks + 8 * g);
        s[f] = mfma16<T>(kf, qf[ks], s[f]);
      }
    }
    float mt = ATT_NEG;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (MASKED) bv = *reinterpret_cast<const f32x4*>(bias_s + 16 * f + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[f][r] = s[f][r] * p.scale_log2e + bv[r];
        mt = fmaxf(mt, s[f][r]);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16));
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[f][r] = exp2f(s[f][r] - m_new);
        psum += s[f][r];
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int nd = 0; nd < 5; ++nd)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[nd][r] *= alpha;

    // ---- O^T += V^T P^T.  k slot (g, j) of 32-key step s2 <-> key 32*s2 + (j < 4 ? 4g + j : 16 + 4g + j - 4)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      Pack8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float_to_elem<T>(pf, r, s[2 * s2][r]);
        float_to_elem<T>(pf, 4 + r, s[2 * s2 + 1][r]);
      }
      const int krow = 32 * s2 + 4 * g + (l15 >> 2);
      const int dcol = 4 * (l15 & 3);
#pragma unroll
      for (int nd = 0; nd < 5; ++nd) {
        Pack4 lo = lds_read_tr16(Vs + krow * ATT_VS + 16 * nd + dcol);
        Pack4 hi = lds_read_tr16(Vs + (krow + 16) * ATT_VS + 16 * nd + dcol);
        Pack8 vf;
        vf.u = (u32x4){lo.u[0], lo.u[1], hi.u[0], hi.u[1]};
        oacc[nd] = mfma16<T>(vf, pf, oacc[nd]);
      }
    }
  }

  // ---- finish: total row sum across the 4 lane groups, normalise, store 4 consecutive d per lane
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_run;
  const int qrow = q0 + l15;
  if (qrow < p.Nq) {
    T* orow = O + ((size_t)b * p.Nq + qrow) * p.ldo + (size_t)h * dh;
#pragma unroll
    for (int nd = 0; nd < 5; ++nd) {
      const int d = 16 * nd + 4 * g;
      if (d + 4 <= dh) {
        Pack4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) float_to_elem4<T>(o, r, oacc[nd][r] * inv);
        *reinterpret_cast<u32x2*>(orow + d) = o.u;
      }
    }
  }
}

}  // namespace ecad
