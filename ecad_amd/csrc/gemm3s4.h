// gemm3s4.h -- gemm3s.h's 256 x 144 tile with the read-modify-write epilogue (both attention out-projections and FF2 of the PixArt path: `cache <- out; h += gate . out`,
// cached_transformer_block.py:243-246,317-320,354-360,383-389) on FOUR waves, one per SIMD, stacked 4 x 1: a wave owns 64 rows x all 144 columns -- 4 A + 9 W fragments per
// 32-deep step for 36 MFMAs (0.36 fragment reads per MFMA against 0.61 on gemm3s's eight 32 x 144 waves) -- and the K loop is one hand-placed asm statement
// (gemm4_asm.h, scripts/gen_gemm4.py, BN 144 variant: three 50 KiB stages by LDS-DMA, one barrier per stage behind a counted vmcnt).  Why: a one-round launch of
// 256 tiles is what M = 8192 x N = 1152 offers, so the wide 256 x 288 tile of gemm2w4.h (same structure, 1.5x the MFMA work per operand byte) covers only half the
// chip there; this is the same hand-placed structure on the tile that fills it.
//
// The statement owns everything that touches the accumulators: it zeroes them, runs the loop, issues the 18 residual loads of this lane in the second-to-last stage
// (behind every operand piece: the HBM round trip travels under two stages of MFMAs) and rounds acc + bias into the row-major C tile in LDS (gemm2w4.h's lesson: hipcc
// treats the AGPRs as free registers once the statement is behind it).  The C++ behind it is gemm3s.h's row pass on the same values in the same order -- cache slot,
// gated residual, replayed cached components (GemmArgs::xsrc), row statistics for a LayerNorm folded into the next GEMM -- so a launch may run on either kernel:
// bit-identical (scripts/probe_gemm3s4.hip, tests/test_gpu_ops.py).
#pragma once
#include "common.h"
#include "gemm.h"
#include "gemm3s.h"
#include "gemm4_asm.h"

namespace ecad {

constexpr int G3S4_CST = GEMM4_CST144;               // bias[144] fp32 | gate table[144] fp32 | gate modulation[144] T, behind the three ring slots
constexpr int G3S4_LDS_BYTES = G3S4_CST + 576 + 576 + 288;
static_assert(G3S4_CST == G3_LDS_BYTES && G3S4_LDS_BYTES <= 160 * 1024, "ring + epilogue constants");
static_assert(G3_BM * (G3_BN + 8) * 2 <= 2 * G3_STAGE_BYTES, "the C tile lies in ring slots 0 and 1");

// whole 256-row tiles inside one sample each (the gate is one vector per tile), K in whole trips of the three-stage loop, at most G3_XSRC_MAX replayed components
template <typename T> inline bool g3s4_ok(const GemmArgs<T>& g) {
  const int rows = (g.m_end ? g.m_end : g.M) - g.m_begin;
  return g.resid && g.M % G3_BM == 0 && rows % G3_BM == 0 && g.m_begin % G3_BM == 0 && g.N % G3_BN == 0 && g.K % (GEMM4_PERIOD_144 * GEMM_BK) == 0 && !g.ln_stat &&
         g.n_xsrc <= G3_XSRC_MAX && (!g.gate_table || (g.rows_per_batch > 0 && g.rows_per_batch % G3_BM == 0)) && (size_t)g.ldr * sizeof(T) * 64 < (1u << 24) && g.ldr % 8 == 0 &&
         g.lda % 8 == 0 && g.ldw % 8 == 0;
}

#ifdef G3S4_STAMP
__device__ unsigned long long g3s4_stamp[4 * 1024];
#endif

template <typename T, int TAG>
__global__ __launch_bounds__(256) void gemm3s4_kernel(const GemmArgs<T> p) {
  constexpr int BM = G3_BM, BN = G3_BN, CS = BN + 8, CPR = BN / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef G3S4_STAMP  // diagnostic build only (scripts/probe_gemm3s4.hip): cycles before / in / behind the statement and the clock held, per workgroup
  unsigned long long st_a, st_b;
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int tiles_n = p.N / BN;
  const int tiles_m = ((p.m_end ? p.m_end : p.M) - p.m_begin) / BM;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // tile id -> origin exactly as gemm3s_kernel
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  tile_coords(bid, tiles_m, tiles_n, tm, tn);
  const int m0 = p.m_begin + tm * BM, n0 = tn * BN;
  const unsigned lds = __builtin_amdgcn_readfirstlane(lds_offset(smem));
  const bool gated = p.gate_table != nullptr, modulated = gated && p.gate_mod != nullptr;
  {
    // Epilogue constants into the LDS behind the ring BEFORE the K loop, by LDS-DMA (gemm2w4.h): older than every operand piece, so the loop's first counted wait
    // covers them and its barriers order them before the readers.  Wave 0: bias, wave 1: gate table, wave 2: the tile's sample's gate modulation.
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    if (wave == 0 && lane < 36) {
      if (p.bias) glds16_raw(p.bias + n0 + 4 * lane, lds + (unsigned)G3S4_CST);
      else reinterpret_cast<f32x4*>(smem + G3S4_CST)[lane] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the statement adds what it finds here
    }
    if (wave == 1 && lane < 36 && gated) glds16_raw(p.gate_table + n0 + 4 * lane, lds + (unsigned)G3S4_CST + 576u);
    if (wave == 2 && lane < 18 && modulated) glds16_raw(p.gate_mod + (size_t)(m0 / p.rows_per_batch) * p.gate_mod_stride + n0 + 8 * lane, lds + (unsigned)G3S4_CST + 1152u);
  }
  u32x4 hq[18];
#ifdef G3S4_STAMP
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_a)::"memory");
#endif
  {
    auto uniform = [](const void* q) {
      const unsigned long long u = (unsigned long long)q;
      return reinterpret_cast<const char*>((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u) |
                                           ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32)) << 32));
    };
    const char* ap = uniform(p.A + (size_t)m0 * p.lda);
    const char* wp = uniform(p.W + (size_t)n0 * p.ldw);
    const char* rbase = uniform(p.resid + (size_t)(m0 + wave * 64) * p.ldr + n0);
    const unsigned rsa = (unsigned)(p.lda * sizeof(T)), rsw = (unsigned)(p.ldw * sizeof(T)), ldrb = (unsigned)(p.ldr * sizeof(T));
    const unsigned trips = (unsigned)(p.K / (GEMM4_PERIOD_144 * GEMM_BK) - 1);
    const unsigned zero = 1u;
    u32x4* h = hq;
    if constexpr (std::is_same<T, bf16_t>::value) { GEMM4_ASM_BF16_144S(ap, wp, rsa, rsw, trips, lds, wave, zero, rbase, ldrb, h); }
    else { GEMM4_ASM_F16_144S(ap, wp, rsa, rsw, trips, lds, wave, zero, rbase, ldrb, h); }
  }
#ifdef G3S4_STAMP
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_b)::"memory");
#endif
  // Row pass.  The statement left round(acc + bias) of rows 64 wave .. 64 wave + 63 in the C tile (written and read by this wave only: no barrier) and this lane's
  // residual chunks in hq[]: mapping A (it < 16) row 4 it + (lane >> 4), chunk lane & 15; mapping B (it 16, 17) row 32 (it - 16) + (lane >> 1), chunk 16 + (lane & 1).
  int lane;  // re-derived behind the statement: nothing per-lane lives across it (gemm2w4.h)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  T* ctile = reinterpret_cast<T*>(smem);
  const char* cst = smem + G3S4_CST;
  const int rowA = lane >> 4, chA = lane & 15, rowB = lane >> 1, chB = 16 + (lane & 1);
  char* const hbase = reinterpret_cast<char*>(p.resid + (size_t)(m0 + wave * 64) * p.ldr + n0);
  char* const cbase = p.C ? reinterpret_cast<char*>(p.C + (size_t)(m0 + wave * 64) * p.ldc + n0) : nullptr;
  const unsigned hrow = (unsigned)p.ldr * (unsigned)sizeof(T), crow = (unsigned)p.ldc * (unsigned)sizeof(T);
  // cached components replayed in this epilogue (gemm3s.h): their chunks of this lane's rows, requested now so that they travel while the first rows are updated
  Pack8 xs[18][G3_XSRC_MAX];
  const int nx = p.n_xsrc;
  if (nx > 0) {
#pragma unroll
    for (int sx = 0; sx < G3_XSRC_MAX; ++sx) {
      if (sx >= nx) break;
      const char* xb = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.xsrc[sx].cache) + (size_t)(m0 + wave * 64) * p.ldr + n0);
#pragma unroll
      for (int it = 0; it < 18; ++it) {
        const int rl = it < 16 ? 4 * it + rowA : 32 * (it - 16) + rowB, ch = it < 16 ? chA : chB;
        xs[it][sx].u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xb + (unsigned)rl * hrow + (unsigned)ch * 16u));
      }
    }
  }
  // the gate of the lane's column chunk: table + modulation as one fp32 sum (gemm3s.h), once per mapping instead of once per row
  float gA[8], gB[8];
  if (gated) {
#pragma clang fp reassociate(off) contract(off)
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(cst + 576 + chA * 32), a1 = *reinterpret_cast<const f32x4*>(cst + 576 + chA * 32 + 16);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(cst + 576 + chB * 32), b1 = *reinterpret_cast<const f32x4*>(cst + 576 + chB * 32 + 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gA[e] = a0[e]; gA[4 + e] = a1[e]; gB[e] = b0[e]; gB[4 + e] = b1[e]; }
    if (modulated) {
      float ma[8], mb[8];
      unpack8<T>(lds_read16(cst + 1152 + chA * 16), ma);
      unpack8<T>(lds_read16(cst + 1152 + chB * 16), mb);
#pragma unroll
      for (int e = 0; e < 8; ++e) { gA[e] += ma[e]; gB[e] += mb[e]; }
    }
  }
#pragma unroll
  for (int it = 0; it < 18; ++it) {
    const int rl = it < 16 ? 4 * it + rowA : 32 * (it - 16) + rowB, ch = it < 16 ? chA : chB;
    const int ml = wave * 64 + rl;
    const Pack8 v = lds_read16(ctile + ml * CS + ch * 8);
    if (cbase) gstore16_nt(cbase + (unsigned)rl * crow + (unsigned)ch * 16u, v);
    Pack8 hv;
    hv.u = hq[it];
    float o[8], h[8];
    unpack8<T>(v, o);
    unpack8<T>(hv, h);
    if (gated) {
#pragma clang fp reassociate(off) contract(off)  // one fused multiply-add per element: gemm3s.h's update, bit for bit
      const float* gt = it < 16 ? gA : gB;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = __builtin_fmaf(gt[e], o[e], h[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] += o[e];
    }
    Pack8 hn = pack8<T>(h);
    if (nx > 0) {  // the replay kernel's arithmetic on top of the rounded update: h <- round(h + sum gate_i . cache_i), sources in plan order (gemm3s.h)
      float h2[8];
      unpack8<T>(hn, h2);
      const int n = n0 + ch * 8;
#pragma unroll
      for (int sx = 0; sx < G3_XSRC_MAX; ++sx) {
        if (sx >= nx) break;
        const SrcDesc d = p.xsrc[sx];
        float cf[8], gate[8];
        unpack8<T>(xs[it][sx], cf);
        if (d.gate_table) {
          unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate_mod) + (size_t)((m0 + ml) / p.rows_per_batch) * p.xsrc_mod_stride + n), gate);
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
    gstore16(hbase + (unsigned)rl * hrow + (unsigned)ch * 16u, hn);
    if (p.stat_out) *reinterpret_cast<u32x4*>(ctile + ml * CS + ch * 8) = hn.u;  // the wave's own rows of the C tile now hold the updated hidden states
  }
  if (p.stat_out) {
    // (sum x, sum x^2) of the tile's 144 columns of the updated rows for a LayerNorm folded into the next GEMM: lane l sums row l of the wave's 64 -- chunks 0..8, then
    // chunks 9..17, lower sum + upper sum: gemm3s.h's order (there two lanes hold the halves and meet through a shuffle), pinned
#pragma clang fp reassociate(off) contract(off)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sx[2] = {0.f, 0.f}, sxx[2] = {0.f, 0.f};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int c = 0; c < CPR / 2; ++c) {
        float x[8];
        unpack8<T>(lds_read16(ctile + (wave * 64 + lane) * CS + (hh * (CPR / 2) + c) * 8), x);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sx[hh] += x[e]; sxx[hh] = __builtin_fmaf(x[e], x[e], sxx[hh]); }
      }
    float* dst = p.stat_out + ((size_t)tn * p.M + m0 + wave * 64 + lane) * 2;
    dst[0] = sx[0] + sx[1];
    dst[1] = sxx[0] + sxx[1];
  }
#ifdef G3S4_STAMP
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = g3s4_stamp + 4 * blockIdx.x;
    o[0] = st_b - st_a; o[1] = st_c1 - st_b; o[2] = st_c1 - st_c0; o[3] = st_r1 - st_r0;
  }
#endif
}

}  // namespace ecad
