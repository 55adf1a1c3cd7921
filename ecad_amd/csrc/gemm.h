// gemm.h -- MFMA GEMM for the dense contractions of the PixArt block (QKV / out-proj / cross-Q /
// text-KV / FF1 / FF2 / caption / adaLN / proj_out), with the cache gate fused into the epilogue.
//
//   C[M,N] = epilogue( A[M,K] (row-major, lda) @ W[N,K]^T (torch Linear layout, ldw) + bias[N] )
//
// Replaces the cuBLAS calls behind diffusers' Attention.to_q/to_k/to_v/to_out and FeedForward
// (reference call sites: ecad/transformer_blocks/cached_transformer_block.py:348-353,374-382) and the
// elementwise tail `cache <- out; hidden = gate * out + hidden` (:243-246,289,317-320,355-360,388-389).
//
// Structure (gfx950): BMxBN output tile per workgroup, BK=64; both operands are K-contiguous so A and W
// tiles share one LDS image: [rows][64] elements = 128-B rows, 16-B chunks XOR-swizzled by ((row>>1)&7)
// (conflict-free ds_read_b128 for 16 rows x one k-chunk).  Tiles arrive by LDS-DMA
// (global_load_lds_dwordx4; the swizzle is applied to the per-lane SOURCE address, the LDS image stays
// lane-linear), double-buffered.  MFMA 16x16x32 with the WEIGHT fragment as the A operand so every lane
// ends up with 4 consecutive output columns -> packed 8-byte LDS writes, then the C tile is re-read row-wise
// for 16-byte coalesced global stores (and 16-byte loads of the residual).
#pragma once
#include "common.h"

namespace ecad {

enum GemmEpi { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_SILU = 2, EPI_BIAS_CACHE_RESID = 3,
               EPI_BIAS_ATTN = 4, EPI_BIAS_ATTN_TAIL = 5,  // gemm3s.h only: the tile is a query projection, attention runs in its tail (gemm3s_attn.h)
               EPI_SPLITK = 6 };  // gemm3s.h latency tiles only: workgroup (x, y) multiplies K-slice y of tile x and leaves fp32 partial sums in `part` (no bias, no rounding)

// One cached sub-layer output to add to the hidden states: h += gate . cache (cached_transformer_block.py:243-246,289,317-320 with the cached tensor of :354-355,383-386)
//   attn1 / ff: gate = scale_shift_table[2|5] + timestep_mod[b][2|5];  attn2: gate = 1 (gate_table == nullptr)
struct SrcDesc {
  const void* cache;
  const float* gate_table;  // [D] fp32 or null
  const void* gate_mod;     // T[b * mod_stride + d] or null
};

template <typename T> struct GemmArgs {
  const T* A;
  const T* W;
  const float* bias;  // [N] fp32 (may be null)
  T* C;               // plain output, or the cache slot for EPI_BIAS_CACHE_RESID (may be null)
  T* resid;           // EPI_BIAS_CACHE_RESID: hidden states, updated in place: resid += gate * out
  const float* gate_table;  // [N] fp32 row of scale_shift_table (null => gate = 1, no mod)
  const T* gate_mod;        // adaLN-single modulation rows: gate_mod[b * gate_mod_stride + n]
  int64_t gate_mod_stride;  // 0 when one timestep is shared by the whole batch
  int rows_per_batch;       // image tokens per sample (row -> b)
  int M, N, K;
  int lda, ldw, ldc, ldr;
  // EPI_BIAS_ATTN / EPI_BIAS_ATTN_TAIL: keys / values of the tile's sample (rows b * att_nk + key, row stride att_ldk, head h at column h * 72),
  // additive per-key bias [Bt][att_nk] (TAIL), output rows [M][att_ldo]
  const T* att_k;
  const T* att_v;
  T* att_out;
  const float* att_bias;
  int att_ldk, att_ldo, att_nk;
  float att_scale_log2e;
  // LayerNorm folded into this GEMM (round 3; shared-timestep plans only).  A = the raw hidden states h, W = W (.) (1 + scale) rounded to T,
  // bias = c2 = shift . W^T + b, and the epilogue computes  rstd_m * acc - rstd_m * mean_m * c1[n] + c2[n]  with c1[n] = sum_k W'[n][k]:
  // ln_stat = per-row partial sums (sum x, sum x^2) over the ln_parts column tiles of h, [ln_parts][M][2] floats, written by the read-modify-write
  // epilogue that produced h (stat_out below); summed here in fixed order.  null: plain GEMM.
  const float* ln_stat;
  const float* ln_c1;
  int ln_parts;
  float ln_eps, ln_inv_d;
  // read-modify-write epilogue (gemm3s.h): also publish (sum x, sum x^2) of the tile's 144 columns of the UPDATED hidden states, [N / 144][M][2] floats
  float* stat_out;
  // read-modify-write epilogue (gemm3s.h), round 4: cached components that FOLLOW this GEMM in the plan with nothing reading the hidden states in between are
  // replayed right here -- h <- round(h + sum_i gate_i . cache_i) on top of this GEMM's own update, the arithmetic of the replay kernel (elementwise.h) -- so the
  // LayerNorm launch / replay flush that would have carried them disappears and the row statistics (stat_out) already describe the final h.  At most G3_XSRC_MAX.
  const struct SrcDesc* xsrc;
  int n_xsrc;
  int64_t xsrc_mod_stride;  // row stride of the sources' gate_mod (0 when one timestep is shared by the whole batch)
  int m_begin, m_end;  // this launch covers rows [m_begin, m_end) of the M-row problem (m_end 0: up to M); both multiples of 256 when set: engine.hip splits a launch whose
                       // last round of tiles would leave most CUs idle into whole rounds + a tail that takes the few-rows kernels
  // split-K for long-K GEMMs with few rows (round 4; FF2 at batch 1-4: 72 K-steps on 8-64 tiles were a 44 us latency chain): the launch has gridDim.y K-slices per
  // tile, slice y writes part[y][M][N] (fp32); splitk_rmw_kernel (gemm3s.h) then adds the slices in slice order, adds the bias and runs the read-modify-write epilogue.
  float* part;
  int epi_direct;  // gemm2w.h plain / GELU / SiLU epilogues: 1 = rows leave straight from the registers (permlane-transposed 16-byte stores), 0 = through a C tile in LDS
  int variant;  // K-loop schedule of gemm3s.h / gemm2w.h: 0 = every wave issues its LDS-DMA pieces in one block after the barrier (round 1),
                // 1 = staggered: pieces interleaved one per MFMA group, waves 0-3 in the first half of the step, waves 4-7 in the second
};

constexpr int GEMM_BK = 64;

// Linear tile id (ids are dealt to XCDs in contiguous runs, 32 consecutive ids = the tiles resident on one XCD's 32
// CUs at one time) -> tile coordinates.  4 x 8 patches of tiles instead of row-major runs: the resident set then
// shares 4 A row-panels and 8 W column-panels (5.0 MB per patch at K = 1152) instead of 1 + 32 (11.2 MB), which keeps
// the streamed operands inside the XCD's 4 MiB L2.
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
  if ((tiles_m & 3) == 0 && (tiles_n & 7) == 0) {
    const int patch = id >> 5, within = id & 31;
    const int pn = tiles_n >> 3;
    const int patch_m = patch / pn, patch_n = patch - patch_m * pn;
    tm = patch_m * 4 + (within >> 3);
    tn = patch_n * 8 + (within & 7);
  } else {
    tm = id / tiles_n;
    tn = id - tm * tiles_n;
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_kernel(const GemmArgs<T> p) {
  constexpr int NWAVES = WAVES_M * WAVES_N;
  constexpr int NT = NWAVES * 64;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // wave tile
  constexpr int FM = WTM / 16, FN = WTN / 16;            // 16x16 fragments per wave
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INST = BM / 8 / NWAVES, B_INST = BN / 8 / NWAVES;  // 1-KiB DMA instructions per wave
  constexpr int CS = BN + 8;                                          // C tile row stride (elements)
  static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "tile/wave mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l15 = lane & 15, g = lane >> 4;

  // ---- tile coordinates, XCD-aware: blocks b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous range of the row-major tile order so concurrently resident tiles share A rows / W rows.
  const int tiles_n = p.N / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane DMA source pointers (advance by 64 elements per K step)
  const T* a_src[A_INST];
  const T* b_src[B_INST];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int r = (wave * A_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int row = m0 + r;
    row = row < p.M ? row : p.M - 1;
    a_src[i] = p.A + (size_t)row * p.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < B_INST; ++i) {
    const int r = (wave * B_INST + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[i] = p.W + (size_t)(n0 + r) * p.ldw + c * 8;
  }

  auto stage = [&](int buf) {
    char* abase = smem + buf * STAGE_BYTES;
    char* bbase = abase + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      glds16(a_src[i], abase + (wave * A_INST + i) * 1024);
      a_src[i] += GEMM_BK;
    }
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
      glds16(b_src[i], bbase + (wave * B_INST + i) * 1024);
      b_src[i] += GEMM_BK;
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = base + 16*i + l15, logical chunk = kk*4 + g, physical = logical ^ ((row>>1)&7)
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wm * WTM + l15) * 128;
  const int b_row_off = (wn * WTN + l15) * 128;

  const int nk = p.K / GEMM_BK;
  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage(buf ^ 1);
    const char* abase = smem + buf * STAGE_BYTES;
    const char* bbase = abase + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + g) ^ swz) * 16;
      Pack8 af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = lds_read16(abase + a_row_off + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = lds_read16(bbase + b_row_off + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<T>(wf[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue 1: acc[i][j][r] = C[m = wm*WTM+16i+l15][n = wn*WTN+16j+4g+r]; bias/activation, round to T,
  // pack 4 consecutive n -> 8-byte LDS write into the row-major C tile (reuses the operand buffers).
  T* ctile = reinterpret_cast<T*>(smem);
  f32x4 bvs[FN];  // one batch of loads (gemm3s.h: hipcc otherwise serializes a load -> wait -> use round trip per column group)
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    bvs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bvs[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * WTN + 16 * j + 4 * g);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = wn * WTN + 16 * j + 4 * g;
    const f32x4 bv = bvs[j];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ml = wm * WTM + 16 * i + l15;
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
  __syncthreads();

  // ---- epilogue 2: row-wise 16-byte pass over the C tile
  constexpr int CPR = BN / 8;            // 16-byte chunks per row
  constexpr int ROWS_PER_PASS = NT / CPR;
  const int cchunk = tid % CPR, crow0 = tid / CPR;
  const int n = n0 + cchunk * 8;
  float gate_t[8];
  if (EPI == EPI_BIAS_CACHE_RESID) {
#pragma unroll
    for (int e = 0; e < 8; ++e) gate_t[e] = p.gate_table ? p.gate_table[n + e] : 1.0f;
  }
#pragma unroll 4
  for (int pass = 0; pass < BM / ROWS_PER_PASS; ++pass) {
    const int ml = pass * ROWS_PER_PASS + crow0;
    const int m = m0 + ml;
    if (m >= p.M) break;
    Pack8 v = lds_read16(ctile + ml * CS + cchunk * 8);
    if (EPI == EPI_BIAS_CACHE_RESID) {
      if (p.C) gstore16(p.C + (size_t)m * p.ldc + n, v);  // cache <- un-gated sub-layer output
      T* rp = p.resid + (size_t)m * p.ldr + n;
      Pack8 hv = gload16(rp);
      float o[8], h[8], gm[8];
      unpack8<T>(v, o);
      unpack8<T>(hv, h);
      if (p.gate_mod) {
        const int b = m / p.rows_per_batch;
        Pack8 gp = gload16(p.gate_mod + (size_t)b * p.gate_mod_stride + n);
        unpack8<T>(gp, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] += (gate_t[e] + gm[e]) * o[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] += gate_t[e] * o[e];
      }
      gstore16(rp, pack8<T>(h));
    } else {
      gstore16(p.C + (size_t)m * p.ldc + n, v);
    }
  }
}

template <int BM, int BN> constexpr int gemm_lds_bytes() {
  return (2 * (BM + BN) * 128) > (BM * (BN + 8) * 2) ? (2 * (BM + BN) * 128) : (BM * (BN + 8) * 2);
}

}  // namespace ecad
