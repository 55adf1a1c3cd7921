// attention128.h -- joint [text | image] attention of the FLUX blocks, head dim 128 (FluxAttnProcessor2_0 /
// FluxSingleAttnProcessor2_0 [ext]; reference call sites ecad/transformer_blocks/cached_flux_transformer_block.py:63-66,178-182).
//
// Same data flow as attention.h (persistent workgroups over (sample, head, 256-query block); K/V chunk of 128 keys staged once
// in LDS and shared by 8 waves; S^T = K Q^T strip in registers; exp2'd P packs straight into the B operand of O^T = V^T P^T;
// the ones column of the V tile accumulates the softmax denominator), restructured for d_h = 128 where the LDS fragment reads,
// not the MFMAs, bound the kernel: a wave owns 32 query rows and BOTH 16-row fragments consume every K / V fragment it reads
// (half the LDS traffic of two sequential rounds), the score strip covers 64 keys at a time (32 VGPRs for both fragments), and
// the V tile row stride is 144 elements (288 B = 8 banks mod 64: the 8 rows x 32 B of a 32-lane transposed read tile the 64
// banks exactly).  Also instantiated for the PixArt head dim 72 (engine.hip), where it replaced the two-sequential-rounds kernel
// of attention.h: self-attention 9.1 -> 7.1 ms per alpha-256 batch, 256 -> 185 ms per Sigma-1024 batch.  Ablation on the FLUX shape (scripts/bench_flux_attn.py, 795 us): without the staging + barriers 508 us, without the K fragment reads 575,
// without the V reads 704, without the PV MFMAs 737, without the exp2s 787 -- the waves run the S / softmax / PV phases in lockstep
// behind the per-chunk barriers, so LDS, MFMA and VALU time add up instead of overlapping.  Measured null or negative on top of this
// kernel: K/V chunks by LDS-DMA into a double-buffered,
// source-swizzled tile with one barrier per chunk and an all-ones MFMA for the denominator (-1 %: kept out of the tree); handing
// each XCD a contiguous run of work items so the 18 query blocks of a (sample, head) share one L2 (+9 % time on FLUX, +13 % on
// Sigma-1024: the CUs of an XCD then request the same K/V lines at the same moment); 4-wave workgroups on 128-query items, two per
// CU (33 -> 52 us per alpha-256 self-attention launch: twice the K/V staging per query row outweighs the decoupled barriers).
// Round 2: the accumulator rescale is skipped when no lane's running maximum grew (wave-uniform branch, exact: alpha == 1.0): +-0 on the PixArt shapes as
// measured before, -2.7 % on FLUX-1024's 4608-key attention (273 -> 265 ms per batch), kept.  Three more rewrites were measured slower there (output stores
// deferred behind the next item's Q prefetch; LDS-transposed 16-byte output stores; all K fragments of a 64-key unit requested before the first MFMA) and a
// fixed s_setprio split between the two waves of a SIMD changed nothing (gemm3s_attn.h).
// Late round 2, by ablation on the FLUX shape (771 us per launch): staging compiled out 491, barriers compiled out 712, ONLY the next chunk's global loads
// compiled out 615 -- asynchronous loads that cost 22 %: their address arithmetic (division by the pieces per row, att_row, 64-bit multiplies, 8 times per
// thread and chunk) was VALU time taken from the softmax.  Now: d_h 128 derives every piece's offsets from the thread id with one multiply-add on a
// wave-uniform chunk base (771 -> 700 us; FLUX-1024 end to end 2.03-2.09 -> 2.17 images/s on the A/B box), d_h 72 uses a uniform base + 32-bit offsets
// (-1.3 %; per-thread offset ARRAYS were slower there: 979 -> 1126 us on the Sigma-1024 shape).  A double-buffered image with the stores between the two
// 64-key units and one barrier per chunk was slower again (771 -> 796, 979 -> 1046).
#pragma once
#include "attention.h"
#include "common.h"

namespace ecad {

constexpr int A128_KC = 128;
constexpr int A128_LDS_BYTES = attn_lds_bytes<A128_KC, 128>();  // 74240
constexpr int A72_LDS_BYTES = attn_lds_bytes<A128_KC, 72>();    // 47616

// TAIL: keys beyond Nk are masked and (PixArt cross-attention) the additive per-key bias p.bias is applied.
// DHC = 128 (FLUX) or 72 (PixArt: K zero-padded to 96, V to 80 in LDS only; the generic kernel's strides).
template <typename T, bool TAIL, int DHC = 128>
__global__ __launch_bounds__(512) void attn128_kernel(const AttnArgs p) {
  constexpr int KC = A128_KC, KS = AttCfg<DHC>::KS, VS = AttCfg<DHC>::VS, DH = DHC, ND = AttCfg<DHC>::ND, KSTEPS = AttCfg<DHC>::KSTEPS, NCH = AttCfg<DHC>::NCH;
  constexpr int NDO = DH / 16 + (DH % 16 ? 1 : 0);  // d-fragments that hold real output columns
  extern __shared__ __attribute__((aligned(16))) char att_smem[];
  T* Ks = reinterpret_cast<T*>(att_smem);
  T* Vs = Ks + KC * KS;
  float* bias_s = reinterpret_cast<float*>(Vs + KC * VS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  constexpr int MAXL = (KC * NCH * 2 + 511) / 512;  // staging loads per thread: 8 (d_h 128) / 5 (d_h 72)

  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  T* O = reinterpret_cast<T*>(p.out);
  const int nqb = (p.Nq + ATT_QB - 1) / ATT_QB;
  const int nkc = (p.Nk + KC - 1) / KC;
  const int items = p.B * p.H * nqb;

  for (int i = tid; i < KC * (KS + VS) / 8; i += 512) reinterpret_cast<u32x4*>(att_smem)[i] = (u32x4){0, 0, 0, 0};
  __syncthreads();
  for (int i = tid; i < KC; i += 512) Vs[i * VS + DH] = (T)1.0f;

  // staging ids run over [K chunks | V chunks]; for d_h = 128 a thread owns one chunk column of 4 K rows and 4 V rows.  Everything that depends on the
  // thread only -- which key and 16-byte piece, its byte offset from the chunk's first row in global memory, its address in the LDS image -- is computed
  // once: stamped by ablation, the per-chunk address arithmetic of the 8 loads (a division by the pieces per row, att_row, 64-bit multiplies) cost 22 % of the
  // kernel on the FLUX shape and the stores' another part of their 16 %, all of it VALU time taken from the softmax.
  Pack8 stg[MAXL];
  // d_h 128: 16 pieces per row, so piece i of a thread is row (tid >> 4) + 32 (i & 3), column piece tid & 15, K for i < 4 and V for i >= 4 -- affine in i,
  // nothing to keep.  d_h 72 (9 pieces per row, no such pattern) keeps the per-chunk arithmetic: per-thread offset arrays measured SLOWER there (Sigma-1024
  // shape 979 -> 1126 us per launch).
  constexpr bool AFFINE = NCH == 16 && MAXL == 8;
  const unsigned a_g0k = (unsigned)(((size_t)(tid >> 4) * p.ldk + (tid & 15) * 8) * sizeof(T)), a_g0v = (unsigned)(((size_t)(tid >> 4) * p.ldv + (tid & 15) * 8) * sizeof(T));
  const unsigned a_l0k = (unsigned)(((tid >> 4) * KS + (tid & 15) * 8) * sizeof(T)), a_l0v = (unsigned)((KC * KS + (tid >> 4) * VS + (tid & 15) * 8) * sizeof(T));
  unsigned l_arr[AFFINE ? 1 : MAXL];  // d_h 72: the LDS address of each piece, computed once (the global side keeps its per-chunk arithmetic, see above)
  if constexpr (!AFFINE) {
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      int id = tid + i * 512;
      const bool v = id >= KC * NCH;
      id = v ? id - KC * NCH : id;
      const int key = id / NCH, c = id - key * NCH;
      l_arr[i] = (unsigned)((v ? KC * KS + key * VS + c * 8 : key * KS + c * 8) * sizeof(T));
    }
  }
  auto isv = [&](int i) -> bool { return i >= 4; };
  auto keyi = [&](int i) -> int { return (tid >> 4) + 32 * (i & 3); };
  auto goff = [&](int i) -> unsigned {
    return i >= 4 ? a_g0v + (unsigned)(i & 3) * (unsigned)(32 * p.ldv * (int)sizeof(T)) : a_g0k + (unsigned)(i & 3) * (unsigned)(32 * p.ldk * (int)sizeof(T));
  };
  auto loff = [&](int i) -> unsigned {
    return i >= 4 ? a_l0v + (unsigned)(i & 3) * (unsigned)(32 * VS * (int)sizeof(T)) : a_l0k + (unsigned)(i & 3) * (unsigned)(32 * KS * (int)sizeof(T));
  };
  auto load_stage = [&](int item, int kc) {
    const int bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    if constexpr (AFFINE) {
      const int k0 = kc * KC;
      // rows of a chunk are consecutive in memory, except that a chunk which straddles the [text | image] boundary of the FLUX layout (seg_t; never for
      // the shapes of the models) continues `jump` rows further on: one wave-uniform correction
      const size_t row0 = att_row(b, k0, p.Nk, p.seg_t, p.seg_bt);
      const bool straddle = p.seg_t != 0 && k0 < p.seg_t && k0 + KC > p.seg_t;
      const long jump = straddle ? (long)att_row(b, p.seg_t, p.Nk, p.seg_t, p.seg_bt) - (long)(row0 + (size_t)(p.seg_t - k0)) : 0;
      const char* kb = reinterpret_cast<const char*>(K + row0 * p.ldk + (size_t)h * DH);  // wave-uniform
      const char* vb = reinterpret_cast<const char*>(V + row0 * p.ldv + (size_t)h * DH);
      const int left = p.Nk - k0, first_after = p.seg_t - k0;
#pragma unroll
      for (int i = 0; i < MAXL; ++i) {
        const bool ok = !TAIL || keyi(i) < left;
        // branch-free: rows past the end read the chunk's first bytes and drop them (a load under a lane-dependent branch makes hipcc wait vmcnt(0))
        const char* src = (isv(i) ? vb : kb) + (ok ? goff(i) : 0u);
        if (straddle && ok && keyi(i) >= first_after) src += jump * (long)((isv(i) ? p.ldv : p.ldk) * (int)sizeof(T));
        const Pack8 v = gload16(src);
        stg[i].u = ok ? v.u : (u32x4){0, 0, 0, 0};
      }
    } else if (p.seg_t == 0) {
      // PixArt layout (rows of a sample consecutive): wave-uniform chunk base + a 32-bit per-lane offset instead of a 64-bit row computation per load
      const size_t row0 = (size_t)b * p.Nk + (size_t)kc * KC;
      const T* kb = K + row0 * p.ldk + (size_t)h * DH;
      const T* vb = V + row0 * p.ldv + (size_t)h * DH;
      const int left = p.Nk - kc * KC;
#pragma unroll
      for (int i = 0; i < MAXL; ++i) {
        int id = tid + i * 512;
        stg[i].u = (u32x4){0, 0, 0, 0};
        if (id < 2 * KC * NCH) {
          const bool v = id >= KC * NCH;
          id = v ? id - KC * NCH : id;
          const int key = id / NCH, c = id - key * NCH;
          if (!TAIL || key < left) stg[i] = gload16((v ? vb : kb) + (unsigned)(key * (v ? p.ldv : p.ldk) + c * 8));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MAXL; ++i) {
        int id = tid + i * 512;
        stg[i].u = (u32x4){0, 0, 0, 0};
        if (id < 2 * KC * NCH) {
          const bool v = id >= KC * NCH;
          id = v ? id - KC * NCH : id;
          const int key = id / NCH, c = id - key * NCH;
          const int gk = kc * KC + key;
          if (!TAIL || gk < p.Nk) {
            const size_t grow = att_row(b, gk, p.Nk, p.seg_t, p.seg_bt);
            const T* base = v ? V + grow * p.ldv : K + grow * p.ldk;
            stg[i] = gload16(base + (size_t)h * DH + c * 8);
          }
        }
      }
    }
  };
  auto store_stage = [&](int item, int kc) {
    if constexpr (AFFINE) {
#pragma unroll
      for (int i = 0; i < MAXL; ++i) *reinterpret_cast<u32x4*>(att_smem + loff(i)) = stg[i].u;
    } else {
#pragma unroll
      for (int i = 0; i < MAXL; ++i)
        if (tid + i * 512 < 2 * KC * NCH) *reinterpret_cast<u32x4*>(att_smem + l_arr[i]) = stg[i].u;
    }
    if (TAIL) {
      const int b = (item / nqb) / p.H;
      for (int i = tid; i < KC; i += 512) {
        const int gk = kc * KC + i;
        float bv = ATT_NEG;
        if (gk < p.Nk) bv = p.bias ? p.bias[(size_t)b * p.Nk + gk] * 1.4426950408889634f : 0.f;
        bias_s[i] = bv;
      }
    }
  };

  // Workgroups b and b + 8 share an XCD (and its L2): give each XCD a contiguous run of work items, so that the query blocks of one (sample, head) -- which all
  // stream the same K / V -- meet in ONE L2 instead of pulling that K / V into all eight (PMC, Sigma-1024 self-attention: 1361 MB fetched per launch against 302 MB
  // algorithmic before this mapping).  Placement is a speed matter only.
  int item = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
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
    Pack8 qfr[2][KSTEPS];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int qrow = qb * ATT_QB + wave * 32 + r * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        qfr[r][ks].u = (u32x4){0, 0, 0, 0};
        if (qrow < p.Nq && 32 * ks + 8 * g + 8 <= DH) qfr[r][ks] = gload16(Q + att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldq + (size_t)h * DH + 32 * ks + 8 * g);
      }
    }
    for (int kc = 0; kc < nkc; ++kc) {
      __syncthreads();  // every wave finished reading the previous chunk
      store_stage(item, kc);
      __syncthreads();
      if (kc + 1 < nkc) load_stage(item, kc + 1);
      else if (item + (int)gridDim.x < items) load_stage(item + gridDim.x, 0);

#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        const T* Kh = Ks + half * 64 * KS;
        const T* Vh = Vs + half * 64 * VS;
        // ---- S^T strips of both query fragments for 64 keys: every K fragment read feeds two MFMAs
        f32x4 s[2][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          s[0][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
          s[1][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            const Pack8 kf = lds_read16(Kh + (16 * f + l15) * KS + 32 * ks + 8 * g);
            s[0][f] = mfma16<T>(kf, qfr[0][ks], s[0][f]);
            s[1][f] = mfma16<T>(kf, qfr[1][ks], s[1][f]);
          }
        }
        float m_new[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float mt = ATT_NEG;
          if (TAIL) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + half * 64 + 16 * f + 4 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                s[r][f][e] = s[r][f][e] * p.scale_log2e + bv[e];
                mt = fmaxf(mt, s[r][f][e]);
              }
            }
          } else {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
              for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[r][f][e]);
            mt *= p.scale_log2e;
          }
          mt = rows_max(mt);
          m_new[r] = fmaxf(m_run[r], mt);
          // the running maximum stops growing after the first chunks of a long key sequence: alpha == exp2(0) == 1.0 exactly for every lane then, and
          // skipping the rescale of the 2 x ND accumulator fragments is bit-identical (wave-uniform branch)
          if (__builtin_amdgcn_ballot_w64(m_new[r] > m_run[r]) != 0) {
            const float alpha = fast_exp2(m_run[r] - m_new[r]);
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
#pragma unroll
              for (int e = 0; e < 4; ++e) oacc[r][nd][e] *= alpha;
          }
          m_run[r] = m_new[r];
        }
        // ---- O^T += V^T P^T: k slot (g, j) of 32-key step s2 <-> key 32 s2 + (j < 4 ? 4g + j : 16 + 4g + j - 4)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          Pack8 pf[2];
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float p0, p1;
              if (TAIL) {
                p0 = fast_exp2(s[r][2 * s2][e] - m_new[r]);
                p1 = fast_exp2(s[r][2 * s2 + 1][e] - m_new[r]);
              } else {
                p0 = fast_exp2(__builtin_fmaf(s[r][2 * s2][e], p.scale_log2e, -m_new[r]));
                p1 = fast_exp2(__builtin_fmaf(s[r][2 * s2 + 1][e], p.scale_log2e, -m_new[r]));
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
            oacc[0][nd] = mfma16<T>(vf, pf[0], oacc[0][nd]);
            oacc[1][nd] = mfma16<T>(vf, pf[1], oacc[1][nd]);
          }
        }
      }
    }
    // ---- denominator = O^T row DH (ones column: d-fragment DH / 16, g = (DH % 16) / 4, e = DH % 4), normalise, store
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      constexpr int DL = DH & 15;
      const float den = __shfl(oacc[r][DH >> 4][DL & 3], l15 + 16 * (DL >> 2));
      const float inv = 1.0f / den;
      const int qrow = qb * ATT_QB + wave * 32 + r * 16 + l15;
      if (qrow < p.Nq) {
        T* orow = O + att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldo + (size_t)h * DH;
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
