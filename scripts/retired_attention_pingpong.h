// retired_attention_pingpong.h (was csrc/attention_pipe.h; built only by scripts/probe_attention.hip) -- attention_dma.h with the two waves of every SIMD in OPPOSITE phases (round 3).
//
// What the stamps of attention_dma.h showed (scripts/probe_attention.hip, FLUX joint shape, cycles per 64-key unit of one wave, 2.2 GHz): S = Q K^T 1330, row
// maximum 680, exp + PV 1410, DMA issue 550, barrier 910 -- 4900 cycles for 1088 cycles of MFMA work.  The eight waves of the workgroup run the same program
// behind the same barrier, so they all want the LDS at once (16 K-fragment reads each), then the matrix pipe, then the VALU (exponentials), then the LDS
// again (V fragments): each unit is used in bursts and idles in between, and the second wave of a SIMD cannot fill the first one's stalls because it stalls
// on the same thing.  Ablations: exponentials removed 0 %, K reads / 4 -23 %, no DMA -14 %, no MFMAs -19 % -- no single unit is the bound, their serialisation is.
// Batching the K reads (one exposed latency instead of eight) only moved the wait into the next phase; an XOR-swizzled conflict-free K image bought 2 %.
//
// Here waves 0-3 (group A) and 4-7 (group B; wave w + 4 shares wave w's SIMD) alternate between two kinds of phases, one barrier apart:
//     M(u):  S' = Q K(u+1)^T  and  O += P(u) V(u)        -- 68 MFMAs (d_h 128) and all the LDS fragment reads
//     V(u):  m' = max(m, rowmax(S')), (rare) O *= exp2(m - m'), P(u) = exp2(S' c - m'), this wave's DMA pieces of image u+2   -- VALU + the DMA issue
// and group B runs one phase behind group A: while one wave of a SIMD multiplies, the other exponentiates and issues DMA.  Every wave executes the same number
// of barriers (B one more at the start, A one more at the end).  Per item: M(-1) [S(0) only], V(0), M(0), ..., V(nu-1), M(nu-1), E [normalise, store, next Q]:
// an even number of phases, so the groups stay one apart across work items.  The arithmetic per query row is attention128.h's operation for operation (same
// fragments, same accumulation order, rescale between PV(u-1) and PV(u)), so results are bit-identical.
//   * K / V arrive by LDS-DMA in 64-key images {K(j) | V(j)} with attention128.h's padded row strides, in a ring of four.  M(u) reads V of image u and K of image
//     u+1; image u+2 is issued in V(u) -- into the slot of image u-2, last read in M(u-2), which group B finished a phase before A's V(u) -- and each wave waits
//     for its own pieces at the end of its next M phase, a phase before the other group first reads them.
//   * Pads (zeros; 1.0 in V column d_h) are written once and never touched by the DMA (EXEC-masked lanes do not write LDS).
// MEASURED (bit-identical to attention128.h on every shape) and RETIRED: Sigma-1024 self-attention (d_h 72, two fragments) 845-853 us against 801-813 for attention_dma.h
// with two fragments and 709-719 with four; FLUX (d_h 128) 609-615 us against 575-616 with the plain M phase, 1080-1165 with the hand-pipelined one (256 VGPRs, spills
// at the item boundaries).  Stamps (Sigma shape, cycles per unit): M phase 1450, V phase 1740, barriers 560 -- the V phase is the long one, because a wave's VALU
// work runs at ~40 % of its rate while the other wave of the SIMD issues MFMAs back to back (scripts/probe_mfma_valu_overlap.hip: 64 v_fma 180 -> 446 cycles, 16 v_exp
// + 16 v_fma 232 -> 343, beside 16 MFMAs per 272 cycles that do not slow down at all).  Opposite phases therefore stretch the softmax instead of hiding it; DESIGN 5g.
#pragma once
#include "../ecad_amd/csrc/attention128.h"
#include "../ecad_amd/csrc/common.h"
#ifndef ATT_STAMP
#define ATT_STAMP 0
#endif

namespace ecad {

constexpr int AP_UK = 64, AP_SLOTS = 4;
#ifndef AP_SB
#define AP_SB 0x7  // sched_barrier mask between a read batch and an MFMA batch: ALU instructions may cross, LDS reads and MFMAs may not
#endif
template <int DHC> constexpr int attn_pipe_image_bytes() { return AP_UK * (AttCfg<DHC>::KS + AttCfg<DHC>::VS) * 2; }
template <int DHC> constexpr int attn_pipe_lds_bytes() { return AP_SLOTS * attn_pipe_image_bytes<DHC>(); }
static_assert(attn_pipe_lds_bytes<128>() <= 160 * 1024, "four 64-key images of the d_h 128 unit");

template <typename T, int DHC, int R = 2>
__global__ __launch_bounds__(512) void attn_pipe_kernel(const AttnArgs p) {
  constexpr int UK = AP_UK, KS = AttCfg<DHC>::KS, VS = AttCfg<DHC>::VS, DH = DHC, ND = AttCfg<DHC>::ND, KSTEPS = AttCfg<DHC>::KSTEPS;
  constexpr int NDO = DH / 16 + (DH % 16 ? 1 : 0);
  constexpr int KPR = KS / 8, VPR = VS / 8;      // 16-byte pieces per image row
  constexpr int KDATA = DH / 8, VDATA = DH / 8;  // of which carry data
  constexpr int K_INST = UK * KS * 2 / 1024, V_INST = UK * VS * 2 / 1024;  // 1-KiB DMA instructions per image: 17 + 18 (d_h 128), 13 + 10 (d_h 72)
  static_assert(UK * KS * 2 % 1024 == 0 && UK * VS * 2 % 1024 == 0, "image regions in whole DMA instructions");
  constexpr int N_INST = K_INST + V_INST;
  constexpr int NI = (N_INST + 7) / 8;   // per wave: 5 (d_h 128) / 3 (d_h 72)
  constexpr int IMG = attn_pipe_image_bytes<DHC>();
  extern __shared__ __attribute__((aligned(16))) char att_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;

  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  T* O = reinterpret_cast<T*>(p.out);
  constexpr int QB = 8 * 16 * R;  // query rows per work item
  const int q_cnt = p.q_cnt > 0 ? p.q_cnt : p.Nq;
  const int nqb = (q_cnt + QB - 1) / QB;
  const int nu = p.Nk / UK;
  const int items = p.B * p.H * nqb;

  for (int i = tid; i < AP_SLOTS * IMG / 16; i += 512) reinterpret_cast<u32x4*>(att_smem)[i] = (u32x4){0, 0, 0, 0};
  __syncthreads();
  for (int i = tid; i < AP_SLOTS * UK; i += 512) {
    T* Vs = reinterpret_cast<T*>(att_smem + (i / UK) * IMG) + UK * KS;
    Vs[(i % UK) * VS + DH] = (T)1.0f;
  }

  unsigned off[NI];
  bool live[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int inst = wave + 8 * j;
    const bool isv = inst >= K_INST;
    const int piece = (isv ? inst - K_INST : inst) * 64 + lane;
    const int row = piece / (isv ? VPR : KPR), col = piece - row * (isv ? VPR : KPR);
    live[j] = inst < N_INST && col < (isv ? VDATA : KDATA);
    off[j] = (unsigned)(((size_t)row * (isv ? p.ldv : p.ldk) + col * 8) * sizeof(T));
  }
  // the image stream: (item, unit) pairs in the order this workgroup consumes them; `nx_*` is the next image to issue, two ahead of the unit whose softmax is running
  int nx_item = blockIdx.x, nx_u = 0, nx_slot = 0;
  auto issue_next = [&]() {
    if (nx_item < items) {
      const int bh = nx_item / nqb;
      const int h = bh % p.H, b = bh / p.H;
      const size_t row0 = att_row(b, nx_u * UK, p.Nk, p.seg_t, p.seg_bt);
      const char* kb = reinterpret_cast<const char*>(K + row0 * p.ldk + (size_t)h * DH);  // wave-uniform
      const char* vb = reinterpret_cast<const char*>(V + row0 * p.ldv + (size_t)h * DH);
      char* img = att_smem + nx_slot * IMG;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int inst = wave + 8 * j;
        if (inst < N_INST) {
          const char* src = (inst >= K_INST ? vb : kb) + off[j];
          if (live[j]) glds16(src, img + inst * 1024);
        }
      }
    }
    nx_slot = (nx_slot + 1) & (AP_SLOTS - 1);
    if (++nx_u == nu) { nx_u = 0; nx_item += gridDim.x; }
  };

  __syncthreads();  // pads and ones columns in place before any image is filled or read
  issue_next();
  issue_next();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // images 0 and 1 complete
  const int group = wave >> 2;
  int slot = 0;  // ring slot of the current item's unit u
  f32x4 oacc[R][ND];
  float m_run[R];
  Pack8 qfr[R][KSTEPS];
  f32x4 s[R][4];
  Pack8 pf0[R], pf1[R];
  auto load_q = [&](int item) {
    const int qb = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int qloc = qb * QB + wave * (16 * R) + r * 16 + l15, qrow = p.q_off + qloc;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        qfr[r][ks].u = (u32x4){0, 0, 0, 0};
        if (item < items && qloc < q_cnt && qrow < p.Nq && 32 * ks + 8 * g + 8 <= DH)
          qfr[r][ks] = gload16(Q + att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldq + (size_t)h * DH + 32 * ks + 8 * g);
      }
    }
  };
  // M phase, software-pipelined by hand (a single wave per SIMD multiplies in this phase: nobody fills a stall on an LDS read).  The phase is a list of groups --
  // four S groups (the 16-key fragment row f: KSTEPS reads, KSTEPS * R MFMAs) and, per 32 keys, ceil(ND / 3) PV groups (three 16-column slices of V: 6 transposed
  // reads, 3 * R MFMAs) -- and the fragment reads of group i+1 are requested before the MFMAs of group i (two register buffers of four fragments).
  constexpr int VG = (ND + 2) / 3, NGK = 4, NGV = 2 * VG;
  auto m_phase = [&](const T* Kn, const T* Vh, auto has_k, auto has_v) {
    constexpr bool HK = decltype(has_k)::value, HV = decltype(has_v)::value;
    constexpr int G0 = HK ? 0 : NGK, G1 = HV ? NGK + NGV : NGK;
    Pack8 fb[2][4];
    auto rd = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < NGK) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) fb[i & 1][ks] = lds_read16(Kn + (16 * i + l15) * KS + 32 * ks + 8 * g);
      } else {
        constexpr int j = i - NGK, s2 = j / VG, nd0 = (j % VG) * 3;
        const int krow = 32 * s2 + 4 * g + (l15 >> 2);
        const int dcol = 4 * (l15 & 3);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if (nd0 + t < ND) {
            const Pack4 lo = lds_read_tr16(Vh + krow * VS + 16 * (nd0 + t) + dcol);
            const Pack4 hi = lds_read_tr16(Vh + (krow + 16) * VS + 16 * (nd0 + t) + dcol);
            fb[i & 1][t].u = (u32x4){lo.u[0], lo.u[1], hi.u[0], hi.u[1]};
          }
        }
      }
    };
    auto mm = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < NGK) {
#pragma unroll
        for (int r = 0; r < R; ++r) s[r][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
          for (int r = 0; r < R; ++r) s[r][i] = mfma16<T>(fb[i & 1][ks], qfr[r][ks], s[r][i]);
      } else {
        constexpr int j = i - NGK, s2 = j / VG, nd0 = (j % VG) * 3;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (nd0 + t < ND) {
#pragma unroll
            for (int r = 0; r < R; ++r) oacc[r][nd0 + t] = mfma16<T>(fb[i & 1][t], s2 ? pf1[r] : pf0[r], oacc[r][nd0 + t]);
          }
      }
    };
    rd(std::integral_constant<int, G0>{});
    static_for<G1 - G0>([&](auto kc) {
      constexpr int i = G0 + decltype(kc)::value;
      __builtin_amdgcn_sched_barrier(AP_SB);
      if constexpr (i + 1 < G1) rd(std::integral_constant<int, i + 1>{});
      __builtin_amdgcn_sched_barrier(AP_SB);
      mm(std::integral_constant<int, i>{});
    });
  };
  // V phase: softmax of the scores in s -> pf0 / pf1, and this wave's share of the DMA
  auto v_phase = [&](bool first) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float mt = ATT_NEG;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[r][f][e]);
      mt *= p.scale_log2e;
      mt = rows_max(mt);
      const float m_new = fmaxf(m_run[r], mt);
      if (!first && __builtin_amdgcn_ballot_w64(m_new > m_run[r]) != 0) {  // exact: alpha == 1.0 when no lane's maximum grew (attention128.h); first unit: O is zero
        const float alpha = fast_exp2(m_run[r] - m_new);
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
          for (int e = 0; e < 4; ++e) oacc[r][nd][e] *= alpha;
      }
      m_run[r] = m_new;
    }
    issue_next();
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = fast_exp2(__builtin_fmaf(s[r][0][e], p.scale_log2e, -m_run[r]));
        const float a1 = fast_exp2(__builtin_fmaf(s[r][1][e], p.scale_log2e, -m_run[r]));
        const float b0 = fast_exp2(__builtin_fmaf(s[r][2][e], p.scale_log2e, -m_run[r]));
        const float b1 = fast_exp2(__builtin_fmaf(s[r][3][e], p.scale_log2e, -m_run[r]));
        float_to_elem<T>(pf0[r], e, a0);
        float_to_elem<T>(pf0[r], 4 + e, a1);
        float_to_elem<T>(pf1[r], e, b0);
        float_to_elem<T>(pf1[r], 4 + e, b1);
      }
  };
#if ATT_STAMP
  unsigned long long st_acc[4] = {0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#define AP_ST(i) { asm volatile("" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define AP_ST(i)
#endif
  auto phase_end = [&](int kind) {
    AP_ST(kind)
    __builtin_amdgcn_s_barrier();
    AP_ST(2)
  };

  load_q(blockIdx.x);
  if (group == 1) phase_end(3);  // group B runs one phase behind
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int qb = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      m_run[r] = ATT_NEG;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) oacc[r][nd] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // M(-1): S(0)
    m_phase(reinterpret_cast<const T*>(att_smem + slot * IMG), (const T*)nullptr, std::true_type{}, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_end(0);
    for (int u = 0; u < nu; ++u) {
      v_phase(u == 0);
      phase_end(1);
      const T* Vh = reinterpret_cast<const T*>(att_smem + slot * IMG) + UK * KS;
      const T* Kn = reinterpret_cast<const T*>(att_smem + ((slot + 1) & (AP_SLOTS - 1)) * IMG);
      if (u + 1 < nu) m_phase(Kn, Vh, std::true_type{}, std::true_type{});
      else m_phase(Kn, Vh, std::false_type{}, std::true_type{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of image u+2 (issued a phase ago): the other group reads them after the next barrier but one
      phase_end(0);
      slot = (slot + 1) & (AP_SLOTS - 1);
    }
    // E: normalise, store, next item's query fragments
#pragma unroll
    for (int r = 0; r < R; ++r) {
      constexpr int DL = DH & 15;
      const float den = __shfl(oacc[r][DH >> 4][DL & 3], l15 + 16 * (DL >> 2));
      const float inv = 1.0f / den;
      const int qloc = qb * QB + wave * (16 * R) + r * 16 + l15, qrow = p.q_off + qloc;
      if (qloc < q_cnt && qrow < p.Nq) {
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
    load_q(item + (int)gridDim.x);
    phase_end(3);
#if ATT_STAMP
    if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && item == (int)blockIdx.x) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + 8 + 4 * group;
      dbg[0] = st_acc[0]; dbg[1] = st_acc[1]; dbg[2] = st_acc[2]; dbg[3] = nu;
    }
#endif
  }
  if (group == 0) phase_end(3);
}

}  // namespace ecad
