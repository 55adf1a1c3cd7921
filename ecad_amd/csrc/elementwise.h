// elementwise.h -- the HBM-bound kernels of the hot path:
//   * cache replay: hidden += sum_i gate_i * cached_i        (the "skip" side of the cache gate)
//   * LayerNorm(no affine) + adaLN-single modulate, with pending cache replays folded in
//   * patch-embed (conv k2 s2 + bias + sincos table), timestep sinusoid, unpatchify,
//     CFG combine + learned-sigma drop + DPM-Solver++(2M) update
// Reference call sites are cited per kernel.
#pragma once
#include "common.h"
#include "fgemm8.h"
#include "gemm.h"  // SrcDesc

namespace ecad {

struct FlushArgs {
  void* h;               // [M, D] hidden states (updated in place when n_src > 0)
  void* y;               // [M, D] LN+modulate output (ln kernel only)
  const SrcDesc* srcs;   // device array
  int n_src;
  int M, D;
  int64_t mod_stride;    // elements between batch entries of the sources' gate_mod pointers (0 = shared timestep)
  int64_t ln_mod_stride; // same for mod_shift / mod_scale
  int rows_per_batch;
  // LN + modulate (ln kernel only): y = LN(h) * (1 + scale) + shift
  const float* table_shift;
  const float* table_scale;
  const void* mod_shift;
  const void* mod_scale;
  // shared-timestep plans (ecad_run_schedule): table + modulation summed once per step into one model-dtype vector each (comb_mod_kernel),
  // as the reference's `scale_shift_table[None] + timestep.reshape(B, 6, -1)` is; non-null => the four pointers above are not read
  const void* comb_shift;
  const void* comb_scale;
  float eps;
  uint8_t* y8;      // opt-in fp8 mode: the LN output leaves as e4m3 rows (stride D bytes) + one fp32 scale per row instead of y
  float* y8_scale;  //   (its consumers are fp8 GEMMs: gemm3s8.h), saving the separate row-quantiser pass
};

// ---- cache replay, flat: each thread owns one 16-byte chunk of the hidden states across all sources.
// Fully cached step: n_src = 84, one streaming pass over the cache arena.
// TAG only separates the kernel symbols (0: mid-step replay, 1: whole cache-hit step) for rocprof's per-kernel statistics.
template <typename T, int TAG = 0>
__global__ __launch_bounds__(256) void cache_accumulate_kernel(const FlushArgs p) {
  const int cpr = p.D >> 3;
  const size_t total = (size_t)p.M * cpr;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int row = (int)(idx / cpr), c = (int)(idx - (size_t)row * cpr);
    const int b = row / p.rows_per_batch;
    const size_t off = (size_t)row * p.D + c * 8;
    T* hp = reinterpret_cast<T*>(p.h) + off;
    float acc[8];
    unpack8<T>(gload16(hp), acc);
    int s = 0;
    for (; s + 4 <= p.n_src; s += 4) {  // 4 sources in flight
      Pack8 cv[4];
      SrcDesc d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = p.srcs[s + u];
        cv[u] = gload16(reinterpret_cast<const T*>(d[u].cache) + off);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float cf[8];
        unpack8<T>(cv[u], cf);
        if (d[u].gate_table) {
          float gm[8];
          unpack8<T>(gload16(reinterpret_cast<const T*>(d[u].gate_mod) + (size_t)b * p.mod_stride + c * 8), gm);
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(d[u].gate_table + c * 8);
          const f32x4 t1 = *reinterpret_cast<const f32x4*>(d[u].gate_table + c * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[e] += (t0[e] + gm[e]) * cf[e];
            acc[4 + e] += (t1[e] + gm[4 + e]) * cf[4 + e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += cf[e];
        }
      }
    }
    for (; s < p.n_src; ++s) {
      const SrcDesc d = p.srcs[s];
      float cf[8];
      unpack8<T>(gload16(reinterpret_cast<const T*>(d.cache) + off), cf);
      if (d.gate_table) {
        float gm[8];
        unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate_mod) + (size_t)b * p.mod_stride + c * 8), gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (d.gate_table[c * 8 + e] + gm[e]) * cf[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += cf[e];
      }
    }
    gstore16(hp, pack8<T>(acc));
  }
}

// ---- cache replay, R rows per thread: a thread owns chunk c of R consecutive rows of one sample, so the gate of (source, sample,
// column) -- 48 bytes of L1/L2 traffic next to every 16-byte cache chunk in the flat kernel -- is fetched once per R chunks, and the
// cache stream is read with non-temporal loads (each slot is read once per step; 1.6 GB never fits the Infinity Cache).
// Probe (scripts/probe_hbm_read.hip): this access pattern without the gate traffic streams at 6.9 TB/s on one MI355X.
template <typename T, int TAG, int R>
__global__ __launch_bounds__(256) void cache_accumulate_rows_kernel(const FlushArgs p) {
  const int cpr = p.D >> 3;
  const unsigned total = (unsigned)(p.M / R) * cpr;
  for (unsigned idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const unsigned g = idx / cpr, c = idx - g * cpr;
    const unsigned row0 = g * R;
    const int b = row0 / p.rows_per_batch;
    const size_t off = (size_t)row0 * p.D + c * 8;
    T* hp = reinterpret_cast<T*>(p.h) + off;
    float acc[R][8];
#pragma unroll
    for (int j = 0; j < R; ++j) unpack8<T>(gload16(hp + (size_t)j * p.D), acc[j]);
    for (int s = 0; s < p.n_src; ++s) {
      const SrcDesc d = p.srcs[s];
      const T* cp = reinterpret_cast<const T*>(d.cache) + off;
      Pack8 cv[R];
#pragma unroll
      for (int j = 0; j < R; ++j) cv[j].u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(cp + (size_t)j * p.D));
      float gate[8];
      if (d.gate_table) {
        unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate_mod) + (size_t)b * p.mod_stride + c * 8), gate);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(d.gate_table + c * 8);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(d.gate_table + c * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { gate[e] += t0[e]; gate[4 + e] += t1[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) gate[e] = 1.0f;
      }
#pragma unroll
      for (int j = 0; j < R; ++j) {
        float cf[8];
        unpack8<T>(cv[j], cf);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] += gate[e] * cf[e];
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) gstore16(hp + (size_t)j * p.D, pack8<T>(acc[j]));
  }
}

// ---- ECAD_NATIVE_CFG_MEAN_CACHE: the cache slot of a CFG-doubled batch becomes the mean of its unconditional and its text half, in both halves (what
// compute_attn_tgate caches at its gate step, cached_transformer_block.py:443-452: `(hidden_uncond + hidden_pred_text) / 2`): fp32 mean, one rounding.
template <typename T>
__global__ __launch_bounds__(256) void cfg_mean_halves_kernel(T* slot, size_t half_elems) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= half_elems) return;
  float a[8], b[8];
  unpack8<T>(gload16(slot + i), a);
  unpack8<T>(gload16(slot + half_elems + i), b);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.5f * (a[e] + b[e]);
  const Pack8 m = pack8<T>(a);
  gstore16(slot + i, m);
  gstore16(slot + half_elems + i, m);
}

// ---- LayerNorm (eps, no affine) + modulate, one wave per row, pending cache replays folded in.
// norm1/norm2 + `x * (1 + scale) + shift` (cached_transformer_block.py:207-215,306-310) and norm_out + modulate
// (pixart_transformer_2d_edited.py:340-348).  MAXC = 16-byte chunks per lane (D <= 512 * MAXC).
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const FlushArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int cpr = p.D >> 3;
  const int b = row / p.rows_per_batch;
  T* hrow = reinterpret_cast<T*>(p.h) + (size_t)row * p.D;
  float x[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < cpr) unpack8<T>(gload16(hrow + c * 8), x[i]);
  }
  if (p.n_src > 0) {
    for (int s = 0; s < p.n_src; ++s) {
      const SrcDesc d = p.srcs[s];
      const T* crow = reinterpret_cast<const T*>(d.cache) + (size_t)row * p.D;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < cpr) {
          float cf[8];
          unpack8<T>(gload16(crow + c * 8), cf);
          if (d.gate_table) {
            float gm[8];
            unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate_mod) + (size_t)b * p.mod_stride + c * 8), gm);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] += (d.gate_table[c * 8 + e] + gm[e]) * cf[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] += cf[e];
          }
        }
      }
    }
    // hidden states are materialised in the storage type between sub-layers
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < cpr) {
        Pack8 o = pack8<T>(x[i]);
        gstore16(hrow + c * 8, o);
        unpack8<T>(o, x[i]);
      }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < cpr)
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += x[i][e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)p.D;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < cpr)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dlt = x[i][e] - mean;
        var += dlt * dlt;
      }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
  const float rstd = rsqrtf(var / (float)p.D + p.eps);

  T* yrow = reinterpret_cast<T*>(p.y) + (size_t)row * p.D;
  const T* msh = reinterpret_cast<const T*>(p.mod_shift) + (size_t)b * p.ln_mod_stride;
  const T* msc = reinterpret_cast<const T*>(p.mod_scale) + (size_t)b * p.ln_mod_stride;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < cpr) {
      float sh[8], sc[8], o[8];
      unpack8<T>(gload16(msh + c * 8), sh);
      unpack8<T>(gload16(msc + c * 8), sc);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float n = (x[i][e] - mean) * rstd;
        o[e] = n * (1.0f + p.table_scale[c * 8 + e] + sc[e]) + (p.table_shift[c * 8 + e] + sh[e]);
      }
      gstore16(yrow + c * 8, pack8<T>(o));
    }
  }
}

// ---- same operation, 16 lanes per row (CPL = D / 128 chunks of 16 bytes per lane, every lane busy: D = 1152 -> 9), so a wave carries 4
// rows with CPL independent loads in flight per lane and the reductions are 4 shuffle steps instead of 6.
// COMB: shift / scale come pre-summed (FlushArgs::comb_*), 32 bytes of modulation per 16-byte chunk instead of 96 -- the loads are L1 / L2
// hits, but the pass is bound by L1 traffic.  A template parameter, not a run-time branch: a branch inside the chunk loop keeps hipcc from
// batching the loads of the 9 chunks (11.7 -> 19.9 us per launch when it was one).
template <typename T, int CPL, bool COMB = false>
__global__ __launch_bounds__(256) void ln_modulate16_kernel(const FlushArgs p) {
  const int l16 = threadIdx.x & 15;
  const int stride = gridDim.x * 16;
  int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= p.M) return;
  // rows of this 16-lane group: row, row + stride, ...; the next row's loads are in flight while the current one is reduced and
  // stored, so the launch is not one read phase followed by one write phase
  Pack8 nx[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) nx[i] = gload16(reinterpret_cast<const T*>(p.h) + (size_t)row * p.D + (l16 + 16 * i) * 8);
  for (; row < p.M; row += stride) {
    const int b = row / p.rows_per_batch;
    T* hrow = reinterpret_cast<T*>(p.h) + (size_t)row * p.D;
    float x[CPL][8];
#pragma unroll
    for (int i = 0; i < CPL; ++i) unpack8<T>(nx[i], x[i]);
    if (row + stride < p.M) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) nx[i] = gload16(reinterpret_cast<const T*>(p.h) + (size_t)(row + stride) * p.D + (l16 + 16 * i) * 8);
    }
    const T* msh = reinterpret_cast<const T*>(p.mod_shift) + (size_t)b * p.ln_mod_stride;
    const T* msc = reinterpret_cast<const T*>(p.mod_scale) + (size_t)b * p.ln_mod_stride;
    if (p.n_src > 0) {
      for (int s = 0; s < p.n_src; ++s) {
        const SrcDesc d = p.srcs[s];
        const T* crow = reinterpret_cast<const T*>(d.cache) + (size_t)row * p.D;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int c = l16 + 16 * i;
          float cf[8];
          unpack8<T>(gload16(crow + c * 8), cf);
          if (d.gate_table) {
            float gm[8];
            unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate_mod) + (size_t)b * p.mod_stride + c * 8), gm);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] += (d.gate_table[c * 8 + e] + gm[e]) * cf[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] += cf[e];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        Pack8 o = pack8<T>(x[i]);
        gstore16(hrow + (l16 + 16 * i) * 8, o);
        unpack8<T>(o, x[i]);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += x[i][e];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)p.D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dlt = x[i][e] - mean;
        var += dlt * dlt;
      }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = rsqrtf(var / (float)p.D + p.eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = l16 + 16 * i;
      float sh[8], sc[8];
      f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0, u0 = t0, u1 = t0;
      if (COMB) {
        unpack8<T>(gload16(reinterpret_cast<const T*>(p.comb_shift) + c * 8), sh);
        unpack8<T>(gload16(reinterpret_cast<const T*>(p.comb_scale) + c * 8), sc);
      } else {
        unpack8<T>(gload16(msh + c * 8), sh);
        unpack8<T>(gload16(msc + c * 8), sc);
        t0 = *reinterpret_cast<const f32x4*>(p.table_scale + c * 8); t1 = *reinterpret_cast<const f32x4*>(p.table_scale + c * 8 + 4);
        u0 = *reinterpret_cast<const f32x4*>(p.table_shift + c * 8); u1 = *reinterpret_cast<const f32x4*>(p.table_shift + c * 8 + 4);
      }
      const float tsc[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]}, tsh[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x[i][e] = round_elem<T>((x[i][e] - mean) * rstd * (1.0f + tsc[e] + sc[e]) + (tsh[e] + sh[e]));
        amax = fmaxf(amax, fabsf(x[i][e]));
      }
    }
    if (p.y8) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
      const float scl = amax > 0.f ? amax / FP8_MAX : 1.0f;
      const float inv = 1.0f / scl;
      if (l16 == 0) p.y8_scale[row] = scl;
      u32x2* d = reinterpret_cast<u32x2*>(p.y8 + (size_t)row * p.D);
#pragma unroll
      for (int i = 0; i < CPL; ++i)
        d[l16 + 16 * i] = (u32x2){pack_fp8x4(x[i][0] * inv, x[i][1] * inv, x[i][2] * inv, x[i][3] * inv), pack_fp8x4(x[i][4] * inv, x[i][5] * inv, x[i][6] * inv, x[i][7] * inv)};
    } else {
      T* yrow = reinterpret_cast<T*>(p.y) + (size_t)row * p.D;
#pragma unroll
      for (int i = 0; i < CPL; ++i) gstore16(yrow + (l16 + 16 * i) * 8, pack8<T>(x[i]));
    }
  }
}

// ---- patch embed: Conv2d(C_in, D, k = p, s = p) + bias + fixed sincos table  (pixart_transformer_2d_edited.py:306)
// grid: (grid_h, Bt); block: 128 threads; each block produces one row of patches.  K = CIN*P*P is tiny (16):
// bandwidth-bound on the [M, D] output, so plain FMAs with the filter taps in registers.
template <typename T, int CIN, int P>
__global__ __launch_bounds__(128) void patch_embed_kernel(const T* __restrict__ latents, const T* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ pos, T* __restrict__ out, int Hl, int Wl, int D) {
  extern __shared__ float xs[];  // [CIN][P][Wl]
  constexpr int KK = CIN * P * P;
  const int gi = blockIdx.x, b = blockIdx.y;
  const int gw = Wl / P, gh = gridDim.x;
  for (int i = threadIdx.x; i < CIN * P * Wl; i += 128) {
    const int c = i / (P * Wl), rem = i - c * (P * Wl), pr = rem / Wl, col = rem - pr * Wl;
    xs[i] = (float)latents[(((size_t)b * CIN + c) * Hl + gi * P + pr) * Wl + col];
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 128) {
    float wv[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) wv[k] = (float)w[(size_t)d * KK + k];
    const float bv = bias[d];
    for (int gj = 0; gj < gw; ++gj) {
      float acc = bv;
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int pr = 0; pr < P; ++pr)
#pragma unroll
          for (int pc = 0; pc < P; ++pc) acc += wv[(c * P + pr) * P + pc] * xs[(c * P + pr) * Wl + gj * P + pc];
      const int tok = gi * gw + gj;
      out[((size_t)b * gh * gw + tok) * D + d] = (T)(acc + pos[(size_t)tok * D + d]);
    }
  }
}

// ---- patch embed, column-stationary: block = 128 output columns x 256 tokens of one sample; thread (cc, ts) keeps the 8 x 16 filter taps of
// its 8 columns in registers and walks the block's tokens ts, ts + 16, ..., one 16-byte store per token.  The kernel above re-reads nothing
// either, but writes 2 bytes per lane per store from 4 waves per CU (43 us for the 18.9 MB output at Bt 32); this one is 16-byte stores with
// 128 independent FMAs between them: 43 -> 22 us per launch (64 / 128 / 256 tokens per block: equal).  P = 2, CIN = 4 (K = 16).
// grid: (D / 128, Bt * ceil(Ntok / TOKB)).
// SOLVE (whole-loop plan, steps 1 ..): the latents this patch embed reads do not exist yet -- they are the CFG + sigma-drop + DPM-Solver++ update of the PREVIOUS step
// (cfg_solver_kernel below, same operations in the same order), computed here on the fly from that step's noise prediction and the solver state, so the separate
// solver launch between two transformer forwards disappears (20 -> 1 per batch).  Every block recomputes the few values it gathers (4 channels x 2 x 2 pixels per token:
// nothing next to the 1152 x 16 MACs per token); the blocks of column block 0 and of the first (unconditional) half of the CFG batch also write the new state, into
// the OTHER buffer pair (ping-pong per step: every reader of this launch sees the old state).
struct PatchSolve {
  const void* noise;      // [2B, Cout, H, W], unconditional half first
  const float* lat_old;   // [B, 4, H, W] fp32
  const float* x0_old;
  float* lat_new;
  float* x0_new;
  const float* coef;      // {alpha_s, sigma_s, c_x, c_m0, c_m1}
  float guidance;
  int B, Cout;
};
template <typename T, int TOKB, bool SOLVE = false>
__global__ __launch_bounds__(256) void patch_embed_cols_kernel(const T* __restrict__ latents, const T* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ pos, T* __restrict__ out, int Hl, int Wl, int D, const PatchSolve sv = PatchSolve{}) {
  constexpr int KK = 16;
  __shared__ float xs[TOKB][KK];
  const int gw = Wl >> 1, ntok = (Hl >> 1) * gw;
  const int tblocks = (ntok + TOKB - 1) / TOKB;
  const int b = blockIdx.y / tblocks, tok0 = (blockIdx.y - b * tblocks) * TOKB;
  const int tid = threadIdx.x;
  // gather: pair (pc = 0, 1) of tap row cp = c * 2 + pr of token t -> xs[t][cp * 2 .. cp * 2 + 1]
  for (int i = tid; i < TOKB * 8; i += 256) {
    const int t = i >> 3, cp = i & 7, tok = tok0 + t;
    float x0 = 0.f, x1 = 0.f;
    if (tok < ntok) {
      const int gi = tok / gw, gj = tok - gi * gw;
      if constexpr (SOLVE) {
        const int bb = b < sv.B ? b : b - sv.B, c = cp >> 1, HW = Hl * Wl, hw = (gi * 2 + (cp & 1)) * Wl + gj * 2;
        const size_t idx = ((size_t)bb * 4 + c) * HW + hw;
        const T* nu = reinterpret_cast<const T*>(sv.noise) + ((size_t)bb * sv.Cout + c) * HW + hw;
        const T* nt = reinterpret_cast<const T*>(sv.noise) + ((size_t)(sv.B + bb) * sv.Cout + c) * HW + hw;
        const float alpha_s = sv.coef[0], sigma_s = sv.coef[1], c_x = sv.coef[2], c0 = sv.coef[3], c1 = sv.coef[4];
        float nxv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float u = (float)nu[e], t = (float)nt[e];
          const float eps = round_elem<T>(u + sv.guidance * (t - u));
          const float x = sv.lat_old[idx + e];
          const float xz = (x - sigma_s * eps) / alpha_s;
          float nx = c_x * x + c0 * xz;
          if (c1 != 0.f) nx += c1 * sv.x0_old[idx + e];
          nx = round_elem<T>(nx);
          if (blockIdx.x == 0 && b < sv.B) { sv.x0_new[idx + e] = xz; sv.lat_new[idx + e] = nx; }
          nxv[e] = nx;
        }
        x0 = nxv[0];
        x1 = nxv[1];
      } else {
      const T* src = latents + (((size_t)b * 4 + (cp >> 1)) * Hl + gi * 2 + (cp & 1)) * Wl + gj * 2;
      x0 = (float)src[0];
      x1 = (float)src[1];
      }
    }
    xs[t][cp * 2] = x0;
    xs[t][cp * 2 + 1] = x1;
  }
  const int cc = tid & 15, ts = tid >> 4;
  const int n0 = blockIdx.x * 128 + cc * 8;
  float wv[8][KK];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float lo[8], hi[8];
    unpack8<T>(gload16(w + (size_t)(n0 + j) * KK), lo);
    unpack8<T>(gload16(w + (size_t)(n0 + j) * KK + 8), hi);
#pragma unroll
    for (int k = 0; k < 8; ++k) { wv[j][k] = lo[k]; wv[j][8 + k] = hi[k]; }
  }
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + n0), b1 = *reinterpret_cast<const f32x4*>(bias + n0 + 4);
  __syncthreads();
  for (int t = ts; t < TOKB; t += 16) {
    const int tok = tok0 + t;
    if (tok >= ntok) break;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(pos + (size_t)tok * D + n0), p1 = *reinterpret_cast<const f32x4*>(pos + (size_t)tok * D + n0 + 4);
    float x[KK];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&xs[t][q * 4]);
      x[q * 4] = v[0]; x[q * 4 + 1] = v[1]; x[q * 4 + 2] = v[2]; x[q * 4 + 3] = v[3];
    }
    float acc[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wv[j][k] * x[k];
    const float o[8] = {acc[0] + p0[0], acc[1] + p0[1], acc[2] + p0[2], acc[3] + p0[3], acc[4] + p1[0], acc[5] + p1[1], acc[6] + p1[2], acc[7] + p1[3]};
    gstore16(out + ((size_t)b * ntok + tok) * D + n0, pack8<T>(o));
  }
}

// comb[s][l][j][:] = T(scale_shift_table_l[slot_j][:] + mod_all[s][slot_j][:]), slots (shift_msa, scale_msa, shift_mlp, scale_mlp) = rows 0, 1, 3, 4
// of the 6-row tables.  grid (steps, L); sst_stride = floats between the tables of consecutive blocks.
template <typename T>
__global__ void comb_mod_kernel(const float* __restrict__ sst0, int64_t sst_stride, const T* __restrict__ mod_all, T* __restrict__ comb, int L, int D) {
  const int s = blockIdx.x, l = blockIdx.y;
  const float* sst = sst0 + (size_t)l * sst_stride;
  const T* mod = mod_all + (size_t)s * 6 * D;
  T* out = comb + ((size_t)s * L + l) * 4 * D;
  for (int i = threadIdx.x; i < 4 * D; i += blockDim.x) {
    const int j = i / D, d = i - j * D;
    const int slot = j < 2 ? j : j + 1;
    out[i] = (T)(sst[slot * D + d] + (float)mod[slot * D + d]);
  }
}

// LayerNorm folded into the consumer GEMM (gemm.h `ln_stat`): per (step, block, component) of a shared-timestep plan
//     Wp[n][k] = T(W[n][k] * (1 + scale[k])),   c1[n] = sum_k Wp[n][k] (of the ROUNDED values),   c2[n] = sum_k shift[k] * W[n][k] + bias[n]
// so that  LN(h)(1 + scale) + shift  times W^T + bias  ==  rstd * (h Wp^T) - rstd * mean * c1 + c2.  One workgroup per output row, fixed-order reduction.
struct FoldDesc {
  const void* W;      // [rows][K]
  const float* bias;  // [rows]
  const void* shift;  // [K] (comb vectors of the step / block, model dtype)
  const void* scale;
  void* Wp;           // [rows][K]
  float* c1;          // [rows]
  float* c2;
  int rows;
};
template <typename T> __global__ __launch_bounds__(256) void ln_fold_weights_kernel(const FoldDesc* __restrict__ descs, int K) {
  const FoldDesc d = descs[blockIdx.y];
  const int n = blockIdx.x;
  if (n >= d.rows) return;
  const T* w = reinterpret_cast<const T*>(d.W) + (size_t)n * K;
  const T* sh = reinterpret_cast<const T*>(d.shift);
  const T* sc = reinterpret_cast<const T*>(d.scale);
  T* wp = reinterpret_cast<T*>(d.Wp) + (size_t)n * K;
  float s1 = 0.f, s2 = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float wv = (float)w[k];
    const T r = (T)(wv * (1.0f + (float)sc[k]));
    wp[k] = r;
    s1 += (float)r;
    s2 = __builtin_fmaf((float)sh[k], wv, s2);
  }
  __shared__ float red[2][256];
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { d.c1[n] = red[0][0]; d.c2[n] = red[1][0] + (d.bias ? d.bias[n] : 0.f); }
}

template <typename T> __global__ void silu_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (T)silu((float)x[i]);
}

// mask (1 keep / 0 discard) -> additive bias (1 - m) * -10000  (pixart_transformer_2d_edited.py:282-289); a bias handed over as such (`user`, :271-279: a 3-D
// encoder_attention_mask is added as given) takes its place
static __global__ void mask_bias_kernel(const uint8_t* __restrict__ mask, const float* __restrict__ user, float* __restrict__ bias, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) bias[i] = user ? user[i] : mask ? (1.0f - (float)(mask[i] != 0)) * -10000.0f : 0.f;
}

// ---- Timesteps(256, flip_sin_to_cos, shift 0): out[m] = [cos(t*f_i) | sin(t*f_i)], f_i = exp(-ln(1e4) i / 128)
template <typename T> __global__ void timestep_sinusoid_kernel(const float* __restrict__ t, T* __restrict__ out, int M) {
  const int m = blockIdx.x, i = threadIdx.x;  // 128 threads
  if (m >= M) return;
  const float f = __expf(-9.210340371976184f * (float)i / 128.0f);
  const float a = t[m] * f;
  out[(size_t)m * 256 + i] = (T)cosf(a);
  out[(size_t)m * 256 + 128 + i] = (T)sinf(a);
}

// ---- PixArtAlphaCombinedTimestepSizeEmbeddings [ext], the size half: block (b, k), k = 0 height, 1 width (resolution_embedder), 2 aspect ratio
// (aspect_ratio_embedder): out[b][k*E .. (k+1)*E) = W2 @ silu(W1 @ sinusoid256(cond[b][k]) + b1) + b2, every stage rounded to T as the
// reference's half-precision modules do.  Step-invariant, 0.74 MMAC per sample: one wave per output element, lanes across K.
template <typename T>
__global__ __launch_bounds__(256) void size_embed_kernel(const float* __restrict__ cond, const T* __restrict__ w1r, const float* __restrict__ b1r, const T* __restrict__ w2r,
                                                         const float* __restrict__ b2r, const T* __restrict__ w1a, const float* __restrict__ b1a, const T* __restrict__ w2a,
                                                         const float* __restrict__ b2a, T* __restrict__ out, int E) {
  extern __shared__ float se_lds[];
  float* sinus = se_lds;        // [256]
  float* hid = se_lds + 256;    // [E]
  const int b = blockIdx.x / 3, k = blockIdx.x % 3;
  const T* w1 = k == 2 ? w1a : w1r;
  const T* w2 = k == 2 ? w2a : w2r;
  const float* b1 = k == 2 ? b1a : b1r;
  const float* b2 = k == 2 ? b2a : b2r;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 128) {
    const float a = cond[blockIdx.x] * __expf(-9.210340371976184f * (float)tid / 128.0f);
    sinus[tid] = (float)(T)cosf(a);
    sinus[128 + tid] = (float)(T)sinf(a);
  }
  __syncthreads();
  for (int j = wave; j < E; j += 4) {
    float acc = 0.f;
    for (int i = lane; i < 256; i += 64) acc += (float)w1[(size_t)j * 256 + i] * sinus[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
      const float v = (float)(T)(acc + b1[j]);
      hid[j] = (float)(T)(v / (1.0f + __expf(-v)));
    }
  }
  __syncthreads();
  for (int j = wave; j < E; j += 4) {
    float acc = 0.f;
    for (int i = lane; i < E; i += 64) acc += (float)w2[(size_t)j * E + i] * hid[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[(size_t)b * 3 * E + (size_t)k * E + j] = (T)(acc + b2[j]);
  }
}

// temb[r][:] += size_emb[bt ? r % bt : 0][:]   (conditioning = timesteps_emb + size_emb)
template <typename T> __global__ void add_size_emb_kernel(T* __restrict__ temb, const T* __restrict__ size_emb, int D, int bt) {
  const int r = blockIdx.x;
  const T* se = size_emb + (size_t)(bt ? r % bt : 0) * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) temb[(size_t)r * D + c] = (T)((float)temb[(size_t)r * D + c] + (float)se[c]);
}

// ---- unpatchify: proj[M, ldp] (first P*P*Cout columns, order (p, q, c)) -> out[Bt, Cout, H, W]
// einsum "nhwpqc->nchpwq" (pixart_transformer_2d_edited.py:352-371)
template <typename T> __global__ void unpatchify_kernel(const T* __restrict__ proj, T* __restrict__ out, int Bt, int gh, int gw, int P, int Cout, int ldp) {
  const int total = Bt * Cout * gh * P * gw * P;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int x = idx;
    const int col = x % (gw * P);
    x /= gw * P;
    const int rowp = x % (gh * P);
    x /= gh * P;
    const int c = x % Cout;
    const int b = x / Cout;
    const int i = rowp / P, pr = rowp % P, j = col / P, pc = col % P;
    out[idx] = proj[((size_t)b * gh * gw + i * gw + j) * ldp + (pr * P + pc) * Cout + c];
  }
}

// ---- guidance + learned-sigma drop + DPM-Solver++(2M) update (pass_through.py:341-370; scheduler [ext]).
// noise: [2B, Cout, H, W] (uncond half first); latents fp32 [B, C, H, W] in place; x0_prev fp32.
// coef = {alpha_s, sigma_s, c_x, c_m0, c_m1}.  Also refreshes the CFG-doubled transformer input for the next step.
// (whole-loop plan: only the LAST step's update runs as this launch -- the others ride in the next step's patch embed, PatchSolve above; lat_in / x0_in are then the
// buffer pair that step left current, and model_in is null: nothing reads it any more)
template <typename T>
__global__ void cfg_solver_kernel(const T* __restrict__ noise, const float* lat_in, const float* x0_in, float* latents, float* x0_prev, T* __restrict__ model_in,
                                  const float* __restrict__ coef, float guidance, int B, int C, int Cout, int HW) {
  const int total = B * C * HW;
  const float alpha_s = coef[0], sigma_s = coef[1], c_x = coef[2], c0 = coef[3], c1 = coef[4];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int hw = idx % HW, c = (idx / HW) % C, b = idx / (HW * C);
    const float u = (float)noise[((size_t)b * Cout + c) * HW + hw];
    const float t = (float)noise[((size_t)(B + b) * Cout + c) * HW + hw];
    const float eps = round_elem<T>(u + guidance * (t - u));
    const float x = lat_in[idx];
    const float x0 = (x - sigma_s * eps) / alpha_s;
    float nx = c_x * x + c0 * x0;
    if (c1 != 0.f) nx += c1 * x0_in[idx];
    nx = round_elem<T>(nx);  // the reference keeps latents in half precision between steps
    x0_prev[idx] = x0;
    latents[idx] = nx;
    if (model_in) {
      model_in[idx] = (T)nx;
      model_in[(size_t)total + idx] = (T)nx;
    }
  }
}

template <typename T> __global__ void cfg_duplicate_kernel(float* __restrict__ latents, T* __restrict__ model_in, int total) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const float x = round_elem<T>(latents[idx]);
    latents[idx] = x;
    model_in[idx] = (T)x;
    model_in[(size_t)total + idx] = (T)x;
  }
}

template <typename TS, typename TD> __global__ void convert_kernel(const TS* __restrict__ src, TD* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (TD)(float)src[i];
}

}  // namespace ecad
