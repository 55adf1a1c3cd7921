// gemm2w4.h -- gemm2w.h's 256 x 288 tile (FF1 + GELU, the wide K,V projection, text K/V) on four waves, one per SIMD: 2 x 2 waves of 128 x 144, the K loop as a
// hand-placed asm statement (gemm4_asm.h, scripts/gen_gemm4.py: BN 288 variant -- 8 A + 9 W fragments per 32-deep step for 72 MFMAs, 14.4 bytes of LDS per KFLOP
// instead of 22; 64 of the wave's 72 accumulator fragments in a[0:255], the ninth column in eight "+v" operands).  Same tile order, LDS image, K order per accumulator
// and epilogue arithmetic as gemm2w_kernel: bit-identical results (scripts/probe_gemm2w4.hip).  The tile walk has no early first K-step (the statement owns its prologue);
// with the direct epilogue the next tile's DMA prologue starts while this tile's stores drain.
#pragma once
#include "common.h"
#include "gemm.h"
#include "gemm2w.h"
#include "gemm4.h"

namespace ecad {

#ifdef G2W4_STAMP
__device__ unsigned long long g2w4_stamp[4 * 1024];
#endif

constexpr int G2W4_CST = G2_TILE_BYTES;               // bias[288] | c1[288] | (rstd, -rstd * mean)[256]: 1152 + 1152 + 2048 bytes behind the operand ring / C tile
constexpr int G2W4_LDS_BYTES = G2W4_CST + 4352;
static_assert(G2W4_LDS_BYTES <= 160 * 1024, "ring + C tile + epilogue constants");

// read-modify-write epilogue on this tile: no replayed components (gemm3s.h carries those), the gate of one sample per tile, whole tiles of the residual
template <typename T> inline bool g2w4_rmw_ok(const GemmArgs<T>& g) {
  return g.resid && g.n_xsrc == 0 && !g.ln_stat && (!g.gate_table || (g.rows_per_batch > 0 && g.rows_per_batch % G2_BM == 0));
}

template <typename T> inline bool g2w4_shape_ok(const GemmArgs<T>& g) {
  const int rows = (g.m_end ? g.m_end : g.M) - g.m_begin;
  return g.M % G2_BM == 0 && rows % G2_BM == 0 && g.m_begin % G2_BM == 0 && g.N % G2_BN == 0 && g.K % (2 * GEMM_BK) == 0;
}

template <typename T, int EPI, int TAG>
__global__ __launch_bounds__(256) void gemm2w4_kernel(const GemmArgs<T> p) {
  constexpr int FM = 8, FN = 9;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int tiles_n = p.N / G2_BN;
  const int tiles_m = ((p.m_end ? p.m_end : p.M) - p.m_begin) / G2_BM;
  const int nwg = tiles_m * tiles_n;
  const unsigned rsa = (unsigned)(p.lda * sizeof(T)), rsw = (unsigned)(p.ldw * sizeof(T));
  const unsigned trips = (unsigned)(p.K / (2 * GEMM_BK) - 1);
  const unsigned lds = __builtin_amdgcn_readfirstlane(lds_offset(smem));
#ifdef G2W4_STAMP  // diagnostic build only (scripts/probe_gemm2w4.hip): cycles in the K loop / in the epilogue and the clock held, per workgroup
  unsigned long long st_loop = 0, st_epi = 0;
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
#ifdef G2W4_STAMP
    unsigned long long st_a, st_b, st_c;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_a)::"memory");
#endif
    int m0, n0;
    {  // tile id -> origin exactly as gemm2w_kernel: XCD-contiguous runs, then tile_coords
      const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, idx = vb >> 3;
      const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
      int tm, tn;
      tile_coords(bid, tiles_m, tiles_n, tm, tn);
      m0 = p.m_begin + tm * G2_BM;
      n0 = tn * G2_BN;
    }
    const bool fold = p.ln_stat != nullptr;
    // Epilogue constants of the tile, put into the LDS behind the ring BEFORE the K loop (nothing of the epilogue then waits on global memory between the tile's stores):
    // the 288 bias values (and the 288 c1 of a folded LayerNorm) by LDS-DMA -- older than every operand piece, so the loop's first counted wait covers them, and the
    // loop's barriers order them before the readers -- and the row coefficients rstd, -rstd * mean of the 256 rows (gemm2w.h) from the per-tile partial sums.
    {
      char* cst = smem + G2W4_CST;
      int t;  // the thread id from v_mbcnt, per tile: everything derived from threadIdx.x is loop-invariant to hipcc, gets hoisted above the tile loop and then has to live
              // across the statement -- in scratch, with an s_waitcnt vmcnt(0) per reload
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t));
      t += wave * 64;
      if (t < 72) {
        if (p.bias) glds16_raw(p.bias + n0 + 4 * t, lds + (unsigned)G2W4_CST + (unsigned)wave * 1024u);
        else if (EPI == EPI_BIAS_CACHE_RESID) reinterpret_cast<f32x4*>(cst)[t] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the statement's own C-tile staging adds what it finds here
        if (fold) glds16_raw(p.ln_c1 + n0 + 4 * t, lds + (unsigned)G2W4_CST + 1152u + (unsigned)wave * 1024u);
      }
      if (fold) {
        float ca[1], cb[1];
        ln_row_coeffs<1>(p, m0 + t, ca, cb);
        reinterpret_cast<f32x2*>(cst + 2304)[t] = (f32x2){ca[0], cb[0]};
      }
      if constexpr (EPI == EPI_BIAS_CACHE_RESID) {  // the gate of the tile's sample (g2w4_shape_ok: the tile lies inside one): table[288] fp32 | modulation[288] T, raw; added at use
        if (p.gate_table) {
          if (t < 72) glds16_raw(p.gate_table + n0 + 4 * t, lds + (unsigned)G2W4_CST + 1152u + (unsigned)wave * 1024u);
          if (p.gate_mod && t < 36) glds16_raw(p.gate_mod + (size_t)(m0 / p.rows_per_batch) * p.gate_mod_stride + n0 + 8 * t, lds + (unsigned)G2W4_CST + 2304u);
        }
      }
    }
    f32x4 c8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c8[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      // (scalar operands of the statement: said so explicitly -- hipcc does not always prove a tile origin uniform)
      auto uniform = [](const void* q) {
        const unsigned long long u = (unsigned long long)q;
        return reinterpret_cast<const char*>((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u) |
                                             ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32)) << 32));
      };
      const char* ap = uniform(p.A + (size_t)m0 * p.lda);
      const char* wp = uniform(p.W + (size_t)n0 * p.ldw);
      const unsigned zero = 1u;
      if constexpr (EPI == EPI_BIAS_CACHE_RESID) {  // K loop + round(acc + bias) into the C tile, all inside the statement (scripts/gen_gemm4.py, stage_c)
        const unsigned cst = lds + (unsigned)G2W4_CST;
        if constexpr (std::is_same<T, bf16_t>::value) { GEMM4_ASM_BF16_288S(ap, wp, rsa, rsw, trips, lds, wave, zero, cst); }
        else { GEMM4_ASM_F16_288S(ap, wp, rsa, rsw, trips, lds, wave, zero, cst); }
      } else if constexpr (std::is_same<T, bf16_t>::value) { GEMM4_ASM_BF16_288(ap, wp, rsa, rsw, trips, lds, wave, zero, c8); }
      else { GEMM4_ASM_F16_288(ap, wp, rsa, rsw, trips, lds, wave, zero, c8); }
    }
#ifdef G2W4_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_b)::"memory");
#endif
    // every per-lane value of the epilogue is derived from this laundered id, so none of it is computed above the K loop and carried across it in the 54 registers the
    // statement leaves (hipcc hoisted the row pointers and spilled them; each scratch reload is an `s_waitcnt vmcnt(0)` between the tile's stores)
    // (the lane id is re-derived by v_mbcnt behind the statement rather than kept: threadIdx.x carried across it was a scratch spill + reload with a full vmcnt(0))
    int lane_fresh;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_fresh));
    const int lane = lane_fresh, l15 = lane & 15, g = lane >> 4;
    const char* cst = smem + G2W4_CST;
    auto raw = [&](auto ic, auto jc) -> f32x4 {
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      if constexpr (j < 8) return g4_acc<i, j>();
      else return c8[i];
    };
    const bool has_bias = p.bias != nullptr;
    auto epilogue = [&](auto foldc) {
    constexpr bool FOLD = decltype(foldc)::value;
    f32x4 bvs[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      bvs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (has_bias) bvs[j] = *reinterpret_cast<const f32x4*>(cst + (wn * 144 + 16 * j + 4 * g) * 4);
    }
    // LayerNorm folded into this GEMM (gemm2w.h): acc <- rstd_m * acc - rstd_m * mean_m * c1[n]
    float ra[FM], rb[FM];
    f32x4 c1v[FN];
    if constexpr (FOLD) {
      const f32x2* rowab = reinterpret_cast<const f32x2*>(cst + 2304);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const f32x2 ab = rowab[wm * 128 + 16 * i + l15];
        ra[i] = ab[0];
        rb[i] = ab[1];
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) c1v[j] = *reinterpret_cast<const f32x4*>(cst + 1152 + (wn * 144 + 16 * j + 4 * g) * 4);
    }
    // (the operation order of gemm2w.h, pinned: under -ffast-math hipcc otherwise merges the bias into the fold as fma(c1, rb, bias) -- one rounding elsewhere)
    auto acc = [&](auto ic, auto jc) -> f32x4 {
#pragma clang fp reassociate(off) contract(off)
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      f32x4 v = raw(ic, jc);
      if constexpr (FOLD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], ra[i], rb[i] * c1v[j][r]);
      }
      return v;
    };
    auto cvt = [&](auto ic, auto jc) {
#pragma clang fp reassociate(off) contract(off)
      constexpr int j = decltype(jc)::value;
      const f32x4 av = acc(ic, jc);
      const f32x4 bv = bvs[j];
      Pack4 o;
      if constexpr (EPI == EPI_BIAS_GELU) {
        const f32x2 v01 = gelu_tanh2((f32x2){av[0] + bv[0], av[1] + bv[1]});
        const f32x2 v23 = gelu_tanh2((f32x2){av[2] + bv[2], av[3] + bv[3]});
        o.u[0] = pack2<T>(v01[0], v01[1]); o.u[1] = pack2<T>(v23[0], v23[1]);  // one v_cvt_pk per pair (same rounding as the scalar conversions)
      } else {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = av[r] + bv[r];
          if (EPI == EPI_BIAS_SILU) v[r] = silu(v[r]);
        }
        o.u[0] = pack2<T>(v[0], v[1]); o.u[1] = pack2<T>(v[2], v[3]);
      }
      return o;
    };
    if constexpr (EPI == EPI_BIAS_CACHE_RESID) {
      // cache <- out; h += gate . out (cached_transformer_block.py:243-246,317-320,355-360) and the row statistics of the updated h for a LayerNorm folded into the next
      // GEMM -- gemm3s.h's read-modify-write epilogue on this tile: the same operations on the same values in the same order (per 144-column half for the statistics),
      // so a launch may run on either kernel.  Wave w streams rows 64 w .. 64 w + 63: 36 sixteen-byte chunks per lane.
      constexpr int CPR = G2_BN / 8, NIT = 64 * CPR / 64;
      // The statement has already rounded acc + bias into the C tile (a C++ staging short of registers had its spills land in the AGPRs, i.e. in the accumulators).
      T* ctile = reinterpret_cast<T*>(smem);
      const int rlane = lane;
      // wave-uniform bases + 32-bit lane offsets: 36 (row, chunk) positions per lane would otherwise be 72 64-bit pointers beside the 144 registers of residual chunks
      char* const hbase = reinterpret_cast<char*>(p.resid + (size_t)(m0 + wave * 64) * p.ldr + n0);
      char* const cbase = p.C ? reinterpret_cast<char*>(p.C + (size_t)(m0 + wave * 64) * p.ldc + n0) : nullptr;
      const unsigned hrow = (unsigned)p.ldr * (unsigned)sizeof(T), crow = (unsigned)p.ldc * (unsigned)sizeof(T);
      Pack8 hpre[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int id = rlane + it * 64;
        const int rl = id / CPR, cchunk = id - rl * CPR;
        hpre[it] = gload16(hbase + (unsigned)rl * hrow + (unsigned)cchunk * 16u);
      }
      __syncthreads();
      const bool gated = p.gate_table != nullptr, modulated = gated && p.gate_mod != nullptr;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int id = rlane + it * 64;
        const int rl = id / CPR, cchunk = id - rl * CPR;
        const int ml = wave * 64 + rl;
        const Pack8 v = lds_read16(ctile + ml * G2_CS + cchunk * 8);
        if (cbase) gstore16_nt(cbase + (unsigned)rl * crow + (unsigned)cchunk * 16u, v);
        float o[8], h[8];
        unpack8<T>(v, o);
        unpack8<T>(hpre[it], h);
        if (gated) {
          // gemm3s.h: gate = table + modulation as one fp32 sum (it passes through LDS there), then h <- fma(gate, out, h), written as the builtin in both files: left to
          // -ffast-math the two kernels came out with different mixtures of fused and unfused updates (they differ on exact ties of the final rounding)
#pragma clang fp reassociate(off) contract(off)
          f32x4 g0 = *reinterpret_cast<const f32x4*>(cst + 1152 + cchunk * 32), g1 = *reinterpret_cast<const f32x4*>(cst + 1152 + cchunk * 32 + 16);
          if (modulated) {
            float gm[8];
            unpack8<T>(lds_read16(cst + 2304 + cchunk * 16), gm);
#pragma unroll
            for (int e = 0; e < 4; ++e) { g0[e] += gm[e]; g1[e] += gm[4 + e]; }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { h[e] = __builtin_fmaf(g0[e], o[e], h[e]); h[4 + e] = __builtin_fmaf(g1[e], o[4 + e], h[4 + e]); }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] += o[e];
        }
        const Pack8 hn = pack8<T>(h);
        gstore16(hbase + (unsigned)rl * hrow + (unsigned)cchunk * 16u, hn);
        if (p.stat_out) *reinterpret_cast<u32x4*>(ctile + ml * G2_CS + cchunk * 8) = hn.u;  // the wave's own rows of the C tile now hold the updated hidden states
      }
      if (p.stat_out) {
        // lane l: row l of the wave's 64, both 144-column halves; per half chunks 0..8 then 9..17 in order, lower sum + upper sum (gemm3s.h's order, pinned)
#pragma clang fp reassociate(off) contract(off)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int tn2 = 2 * (n0 / G2_BN);
#pragma unroll
        for (int part = 0; part < 2; ++part) {
          float sx[2] = {0.f, 0.f}, sxx[2] = {0.f, 0.f};
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
              float x[8];
              unpack8<T>(lds_read16(ctile + (wave * 64 + rlane) * G2_CS + (part * 18 + hh * 9 + c) * 8), x);
#pragma unroll
              for (int e = 0; e < 8; ++e) { sx[hh] += x[e]; sxx[hh] = __builtin_fmaf(x[e], x[e], sxx[hh]); }
            }
          float* dst = p.stat_out + ((size_t)(tn2 + part) * p.M + m0 + wave * 64 + rlane) * 2;
          dst[0] = sx[0] + sx[1];
          dst[1] = sxx[0] + sxx[1];
        }
      }
      return;
    }
    const bool direct = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && p.epi_direct > (EPI == EPI_BIAS_GELU ? 0 : 1);
    if (direct) {
      // rows leave straight from the registers: fragments j, j + 1 paired by v_permlane16_swap into 16-byte pieces (gemm2w.h), the ninth fragment in 8-byte stores
      static_for<FM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int m = m0 + wm * 128 + 16 * i + l15;
        T* crow = p.C + (size_t)m * p.ldc + n0 + wn * 144;
        static_for<FN / 2>([&](auto pc) {
          constexpr int j = 2 * decltype(pc)::value;
          Pack4 a = cvt(ic, std::integral_constant<int, j>{}), b = cvt(ic, std::integral_constant<int, j + 1>{});
          asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a.u[0]), "+v"(b.u[0]));
          asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a.u[1]), "+v"(b.u[1]));
          Pack8 v;
          v.u = (u32x4){a.u[0], a.u[1], b.u[0], b.u[1]};
          gstore16(crow + 16 * (j + (g & 1)) + 8 * (g >> 1), v);
        });
        const Pack4 o = cvt(ic, std::integral_constant<int, FN - 1>{});
        *reinterpret_cast<u32x2*>(crow + 16 * (FN - 1) + 4 * g) = o.u;
      });
      return;
    }
    // staged: the wave's 128 x 144 block -> row-major C tile in LDS, one barrier, wave w streams rows 64 w .. 64 w + 63 in 16-byte pieces
    T* ctile = reinterpret_cast<T*>(smem);
    static_for<FN>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int nl = wn * 144 + 16 * j + 4 * g;
      static_for<FM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int ml = wm * 128 + 16 * i + l15;
        const Pack4 o = cvt(ic, jc);
        *reinterpret_cast<u32x2*>(ctile + ml * G2_CS + nl) = o.u;
      });
    });
    __syncthreads();
    constexpr int CPR = G2_BN / 8;
    constexpr int CHUNKS = 64 * CPR;
    for (int id = lane; id < CHUNKS; id += 64) {
      const int rl = id / CPR, cchunk = id - rl * CPR;
      const int ml = wave * 64 + rl;
      gstore16(p.C + (size_t)(m0 + ml) * p.ldc + n0 + cchunk * 8, lds_read16(ctile + ml * G2_CS + cchunk * 8));
    }
    };
    if (fold) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if (vb + (int)gridDim.x < nwg) __syncthreads();  // the C tile and the constants have been read before the next tile overwrites them
#ifdef G2W4_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c)::"memory");
    st_loop += st_b - st_a; st_epi += st_c - st_b;
#endif
  }
#ifdef G2W4_STAMP
  if (threadIdx.x == 0) {
    const unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = g2w4_stamp + 4 * blockIdx.x;
    o[0] = st_loop; o[1] = st_epi; o[2] = st_c1 - st_c0; o[3] = st_r1 - st_r0;
  }
#endif
}

}  // namespace ecad
