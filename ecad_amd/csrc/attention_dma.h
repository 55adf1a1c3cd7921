// attention_dma.h -- attention128.h with the K / V chunks filled by LDS-DMA into a double-buffered image (round 3).
//
// attention128.h stages a 128-key chunk through registers: global loads right after the barrier, ds_write_b128 of the NEXT chunk between two barriers at the top
// of the following iteration.  By ablation (its header) that store phase -- during which no wave multiplies -- is 16 % of the FLUX launch, the second barrier
// 3.5 %, and the staging registers (32 VGPRs) are what a software-pipelined unit loop would need.  Here every wave writes its share of the next chunk straight
// into the OTHER half of a two-image LDS buffer with global_load_lds_dwordx4 (1 KiB per wave instruction, LDS destination = wave-uniform base + lane * 16) while
// the current image is being multiplied: one barrier per chunk, no register staging, no store phase.
//   * The images keep attention128.h's padded row strides (d_h 128: K 272 B = 17 pieces, V 288 B = 18 pieces; d_h 72: K 208 B = 13, V 160 B = 10), so the
//     fragment reads, the MFMA order and therefore the results are bit for bit attention128.h's.  A DMA instruction covers 64 consecutive 16-byte pieces of the
//     image, pads included: lanes that fall on a pad are switched off (EXEC), so the pads keep what the prologue put there once -- zeros, and 1.0 in the V
//     column d_h (the ones column that accumulates the softmax denominator).
//   * Which row and column a lane's piece is depends on the thread and the instruction only: the byte offset from the chunk's first row is computed once per
//     thread (NI registers) and added to a wave-uniform chunk base.
//   * Self / joint attention with Nk % 128 == 0 and no chunk across the FLUX [text | image] boundary (true for every model shape); everything else stays on
//     attention128.h.
// Measured (scripts/ab_attn_dma.py, one process, interleaved, results bit-identical in every variant; FLUX joint shape B 2 x 24 heads x 128, S 4608 / Sigma-1024
// self-attention Bt 8 x 16 heads x 72, N 4096 / alpha-512 N 1024 / alpha-256 N 256, microseconds per launch):
//     attention128.h (register staging, two barriers per chunk)               600-605 / 830-841 / 255-262 / 28.4-29.4
//     this kernel (LDS-DMA, one barrier per chunk)                             598-600 / 820-830 / 258-265 / 29.5-30.1     <- opt-in: ECAD_ATTN=dma, ECAD_FLUX_ATTN=dma
//     + fragment reads one batch ahead of their MFMAs (sched_barrier-pinned)   610-612 / 858      / 283-290 / 30.0
//     + the DMA pieces one per MFMA batch, branch-free (pads reload row data,
//       the ones piece reloads {1, 0, ...} from a constant block)             699-701 / 911-913 / 301-303 / 28.7          (242 VGPRs, 19 spilled SGPRs)
//     three-stage software pipeline over 64-key units (PV(u-1) | softmax(u) |
//       S(u+1) in one basic block, ring of four DMA images, barrier per unit)  661      / 947-965 / 301-308 / 29.5
//   So the store phase and the second barrier that the ablation of attention128.h priced at 16 % + 3.5 % were not what the launch waits for: removing them buys
//   0-1.5 %, and no re-ordering of the fragment reads, of the DMA issue or of the three streams that hipcc will compile beats the plain loop.  What does bound it
//   showed up by accident: with the DMA of every chunk after the first left out (a bug in a variant: stale keys, finite garbage) the same instruction stream
//   runs in 443 / 619 / 188 / 25.4 us -- 26 % less.  Moving 70 KB (46 KB) of K / V per 128-key chunk into each CU's LDS is what the other 26 % are, although the
//   rate (~7 B/clk per CU) is a tenth of what the L2 -> LDS path sustains in the GEMMs; DESIGN 5d discusses what is left (K / V staged once for 512 query rows,
//   which the 72-register accumulators of d_h 72 would allow and those of d_h 128 do not).  EXEC-masked lanes of global_load_lds_dwordx4 do NOT write their 16
//   bytes of LDS: the pads of the images, written once in the prologue, survive every refill (that is what makes this kernel's results exact).
#pragma once
#include "attention128.h"
#include "common.h"

namespace ecad {

// Timing-only ablations for scripts/probe_attention.hip (honoured only under -DECAD_DIAG, which the product library cannot be built with; results are wrong by design): bit 0 no exponential, 1 one K fragment
// read per K-step instead of four, 2 one V fragment read per 32 keys instead of ND, 3 no DMA after the first chunk, 4 no S MFMAs, 5 no PV MFMAs, 6 no barrier.
#if !defined(ECAD_DIAG) || !defined(ATT_ABL)
#undef ATT_ABL
#define ATT_ABL 0
#endif
// Probe switch (scripts/probe_attention.hip): ATT_KSWZ -- d_h 128 K image in unpadded 256-byte rows with the 16-byte column XOR-swizzled by the row (conflict-free
// ds_read_b128; the padded 272-byte rows are 2-way on one slot per lane group): -2 %.  Measured and removed again: the DMA pieces of waves 4-7 issued
// half a chunk later (+6 %), all K fragments of a unit requested in one batch in front of its S MFMAs (+4 %: the wait moves into the row-maximum phase).
#if !defined(ECAD_DIAG) || !defined(ATT_KSWZ)
#undef ATT_KSWZ
#define ATT_KSWZ 0
#endif
// ATT_STAMP: wave 0 of workgroup 0 sums s_memtime deltas per loop phase into p.bias (reused as a u64[8] debug buffer): wait+barrier, DMA issue, S, softmax, PV, units
#if !defined(ECAD_DIAG) || !defined(ATT_STAMP)
#undef ATT_STAMP
#define ATT_STAMP 0
#endif

template <int DHC> constexpr int attn_dma_image_bytes() { return A128_KC * (AttCfg<DHC>::KS + AttCfg<DHC>::VS) * 2; }
template <int DHC> constexpr int attn_dma_lds_bytes() { return 2 * attn_dma_image_bytes<DHC>(); }
static_assert(attn_dma_lds_bytes<128>() <= 160 * 1024, "two images of the d_h 128 chunk");

// R = query fragments (16 rows each) per wave: 2 (256 query rows per work item, as attention128.h) or 4 (512 rows: every K / V fragment read feeds four MFMAs and a
// chunk is staged once for twice the rows; fits the 72-wide head only -- d_h 128 would need 144 accumulator + 64 query + 64 score registers).  DESIGN 5d.
// NW = waves per workgroup: 8 (two per SIMD, <= 256 registers each) or 4 (one per SIMD with the whole 512-register file: eight query fragments per wave at d_h 72).
template <typename T, int DHC, int R = 2, int NW = 8>
__global__ __launch_bounds__(NW * 64) void attn_dma_kernel(const AttnArgs p) {
  constexpr bool KSWZ = ATT_KSWZ && DHC == 128;
  constexpr int KC = A128_KC, KS = KSWZ ? 128 : AttCfg<DHC>::KS, VS = AttCfg<DHC>::VS, DH = DHC, ND = AttCfg<DHC>::ND, KSTEPS = AttCfg<DHC>::KSTEPS;
  constexpr int NDO = DH / 16 + (DH % 16 ? 1 : 0);
  constexpr int KPR = KS / 8, VPR = VS / 8;        // 16-byte pieces per image row
  constexpr int KDATA = DH / 8, VDATA = DH / 8;    // of which carry data
  constexpr int K_INST = KC * KS * 2 / 1024, V_INST = KC * VS * 2 / 1024;  // 1-KiB DMA instructions per image: 34 + 36 (d_h 128), 26 + 20 (d_h 72)
  static_assert(KC * KS * 2 % 1024 == 0 && KC * VS * 2 % 1024 == 0, "image regions in whole DMA instructions");
  constexpr int N_INST = K_INST + V_INST;
  constexpr int NI = (N_INST + NW - 1) / NW;       // per wave (8 waves): 9 (d_h 128) / 6 (d_h 72)
  constexpr int IMG = attn_dma_image_bytes<DHC>();
  extern __shared__ __attribute__((aligned(16))) char att_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;

  const T* Q = reinterpret_cast<const T*>(p.q);
  const T* K = reinterpret_cast<const T*>(p.k);
  const T* V = reinterpret_cast<const T*>(p.v);
  T* O = reinterpret_cast<T*>(p.out);
  constexpr int QB = NW * 16 * R;  // query rows per work item
  const int q_cnt = p.q_cnt > 0 ? p.q_cnt : p.Nq;  // a launch may cover a slice of the query rows of every (sample, head): see launch_attn in flux.hip
  const int nqb = (q_cnt + QB - 1) / QB;
  const int nkc = p.Nk / KC;
  const int items = p.B * p.H * nqb;

  // both images: zeros everywhere, 1.0 in column DH of every V row; the DMA never touches a pad
  for (int i = tid; i < 2 * IMG / 16; i += NW * 64) reinterpret_cast<u32x4*>(att_smem)[i] = (u32x4){0, 0, 0, 0};
  __syncthreads();
  for (int i = tid; i < 2 * KC; i += NW * 64) {
    T* Vs = reinterpret_cast<T*>(att_smem + (i / KC) * IMG) + KC * KS;
    Vs[(i % KC) * VS + DH] = (T)1.0f;
  }

  // instruction j of this wave = image instruction wave + 8 j; its lane covers piece 64 * inst + lane of the K region (inst < K_INST) or of the V region
  unsigned off[NI];
  bool live[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int inst = wave + NW * j;
    const bool isv = inst >= K_INST;
    const int piece = (isv ? inst - K_INST : inst) * 64 + lane;
    const int row = piece / (isv ? VPR : KPR), col = piece - row * (isv ? VPR : KPR);
    live[j] = inst < N_INST && col < (isv ? VDATA : KDATA);
    off[j] = (unsigned)(((size_t)row * (isv ? p.ldv : p.ldk) + ((KSWZ && !isv) ? (col ^ (row & 15)) : col) * 8) * sizeof(T));
  }
  auto issue_chunk = [&](int item, int kc, int buf) {
    const int bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    const size_t row0 = att_row(b, kc * KC, p.Nk, p.seg_t, p.seg_bt);
    const char* kb = reinterpret_cast<const char*>(K + row0 * p.ldk + (size_t)h * DH);  // wave-uniform
    const char* vb = reinterpret_cast<const char*>(V + row0 * p.ldv + (size_t)h * DH);
    char* img = att_smem + buf * IMG;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int inst = wave + NW * j;
      if (inst < N_INST) {                                  // wave-uniform
        const char* src = (inst >= K_INST ? vb : kb) + off[j];
        if (live[j]) glds16(src, img + inst * 1024);        // pad lanes are off: their 16 bytes of the image keep the prologue's contents
      }
    }
  };

  // Workgroups b and b + 8 share an XCD (and its L2): give each XCD a contiguous run of work items, so that the query blocks of one (sample, head) -- which all
  // stream the same K / V -- meet in ONE L2 instead of pulling that K / V into all eight (PMC, Sigma-1024 self-attention: 1361 MB fetched per launch against 302 MB
  // algorithmic before this mapping).  Placement is a speed matter only.
  int item = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  int buf = 0;
  __syncthreads();  // ones columns in place before anybody reads an image
  if (item < items) issue_chunk(item, 0, 0);
  for (; item < items; item += gridDim.x) {
    const int qb = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    f32x4 oacc[R][ND];
    float m_run[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      m_run[r] = ATT_NEG;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) oacc[r][nd] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    Pack8 qfr[R][KSTEPS];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int qloc = qb * QB + wave * (16 * R) + r * 16 + l15, qrow = p.q_off + qloc;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        qfr[r][ks].u = (u32x4){0, 0, 0, 0};
        if (qloc < q_cnt && qrow < p.Nq && 32 * ks + 8 * g + 8 <= DH) qfr[r][ks] = gload16(Q + att_row(b, qrow, p.Nq, p.seg_t, p.seg_bt) * p.ldq + (size_t)h * DH + 32 * ks + 8 * g);
      }
    }
#if ATT_STAMP
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
#define ATT_ST(i) { asm volatile("" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
#else
#define ATT_ST(i)
#endif
    for (int kc = 0; kc < nkc; ++kc) {
      // this wave's pieces of chunk kc have landed (they are the newest DMA it issued); after the barrier everybody's have, and nobody reads the other image
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(ATT_ABL & 64)) __builtin_amdgcn_s_barrier();
      ATT_ST(0)
      const T* Ks = reinterpret_cast<const T*>(att_smem + buf * IMG);
      const T* Vs = Ks + KC * KS;
      const bool more = kc + 1 < nkc;
      const int n_item = more ? item : item + (int)gridDim.x, n_kc = more ? kc + 1 : 0;
      const bool any_next = n_item < items;
      const bool do_issue = any_next && !((ATT_ABL & 8) && (kc > 0 || item != (int)blockIdx.x));
      if (do_issue) issue_chunk(n_item, n_kc, buf ^ 1);

      ATT_ST(1)
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        const T* Kh = Ks + half * 64 * KS;
        const T* Vh = Vs + half * 64 * VS;
        f32x4 s[R][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
#pragma unroll
          for (int r = 0; r < R; ++r) s[r][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            const Pack8 kf = lds_read16(Kh + (16 * ((ATT_ABL & 2) ? 0 : f) + l15) * KS + (KSWZ ? ((4 * ks + g) ^ l15) * 8 : 32 * ks + 8 * g));
#pragma unroll
            for (int r = 0; r < R; ++r) {
              if (ATT_ABL & 16) s[r][f][ks & 3] += __builtin_bit_cast(float, kf.u[0] & 0x3f800000u) * __builtin_bit_cast(float, qfr[r][ks].u[1] & 0x3f800000u);
              else s[r][f] = mfma16<T>(kf, qfr[r][ks], s[r][f]);
            }
          }
        }
        ATT_ST(2)
        float m_new[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float mt = ATT_NEG;
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[r][f][e]);
          mt *= p.scale_log2e;
          mt = rows_max(mt);
          m_new[r] = fmaxf(m_run[r], mt);
          if (__builtin_amdgcn_ballot_w64(m_new[r] > m_run[r]) != 0) {  // exact: alpha == 1.0 when no lane's maximum grew (attention128.h)
            const float alpha = fast_exp2(m_run[r] - m_new[r]);
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
#pragma unroll
              for (int e = 0; e < 4; ++e) oacc[r][nd][e] *= alpha;
          }
          m_run[r] = m_new[r];
        }
        ATT_ST(3)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          Pack8 pf[R];
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float p0 = __builtin_fmaf(s[r][2 * s2][e], p.scale_log2e, -m_new[r]);
              float p1 = __builtin_fmaf(s[r][2 * s2 + 1][e], p.scale_log2e, -m_new[r]);
              if (!(ATT_ABL & 1)) { p0 = fast_exp2(p0); p1 = fast_exp2(p1); }
              float_to_elem<T>(pf[r], e, p0);
              float_to_elem<T>(pf[r], 4 + e, p1);
            }
          const int krow = 32 * s2 + 4 * g + (l15 >> 2);
          const int dcol = 4 * (l15 & 3);
#pragma unroll
          for (int nd = 0; nd < ND; ++nd) {
            const Pack4 lo = lds_read_tr16(Vh + krow * VS + 16 * ((ATT_ABL & 4) ? 0 : nd) + dcol);
            const Pack4 hi = lds_read_tr16(Vh + (krow + 16) * VS + 16 * ((ATT_ABL & 4) ? 0 : nd) + dcol);
            Pack8 vf;
            vf.u = (u32x4){lo.u[0], lo.u[1], hi.u[0], hi.u[1]};
#pragma unroll
            for (int r = 0; r < R; ++r) {
              if (ATT_ABL & 32) oacc[r][nd][0] += __builtin_bit_cast(float, vf.u[0] & 0x3f800000u) * __builtin_bit_cast(float, pf[r].u[nd & 3] & 0x3f800000u);
              else oacc[r][nd] = mfma16<T>(vf, pf[r], oacc[r][nd]);
            }
          }
        }
        ATT_ST(4)
      }
      buf ^= 1;
    }
#if ATT_STAMP
    if (blockIdx.x == 0 && tid == 0 && item == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias));
      for (int i = 0; i < 5; ++i) dbg[i] = st_acc[i];
      dbg[5] = 2ull * nkc;
      dbg[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
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
  }
}

}  // namespace ecad
