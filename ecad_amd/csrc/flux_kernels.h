// flux_kernels.h -- the HBM-bound kernels of the FLUX blocks.
//   * LayerNorm (no affine) + adaLN-Zero modulate `LN(x) * (1 + scale[b]) + shift[b]`, pending gated cache replays folded in
//     (cached_flux_transformer_block.py:106,238-248,259-262,276-280; AdaLayerNormZero / ZeroSingle / Continuous [ext])
//   * gated cache replay `hidden += gate[b] * cached`          (the "skip" side of the gate: :128-129,256-257,267-269,273-274,285-287)
//   * per-head RMSNorm(q), RMSNorm(k) + RoPE, in place on the fused QKV buffer  (FluxAttnProcessor2_0 [ext])
//   * flow-match Euler update on the packed latents               (FlowMatchEulerDiscreteScheduler.step [ext])
// Row -> sample index is segment aware (see fgemm.h): rows [0, seg0) are text rows, the rest image rows.
#pragma once
#include "common.h"
#include "fgemm.h"
#include "fgemm8.h"

namespace ecad {

struct FluxSrc {
  const void* cache;  // [M, D] rows aligned with the hidden-state rows this launch covers
  const void* gate;   // T[b * gate_stride + d] or null (gate 1)
  int64_t clip16;     // 1: the sum with this source is a FLUX block's output: fp16 saturates at +-65504 (cached_flux_transformer_block.py:127-128,288-289)
};

struct FluxRowArgs {
  void* h;             // [M, D] hidden states (updated in place when n_src > 0)
  void* y;             // LN output (ln kernel only)
  uint8_t* y8;         // fp8 mode: the LN output goes out as e4m3 rows (row stride D bytes) with one fp32 scale per row instead of y,
  float* y8_scale;     //   so the consuming fp8 GEMM needs no separate quantisation pass (fgemm8.h)
  const FluxSrc* srcs; // device array
  int n_src;
  int M, D;
  int seg0, rows_per_batch0, rows_per_batch1;
  int64_t gate_stride;  // elements between samples of every source's gate pointer
  const void* shift;    // T[b * mod_stride + d]
  const void* scale;
  int64_t mod_stride;
  float eps;
};

// one wave per row; MAXC 16-byte chunks per lane (D <= 512 * MAXC)
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void flux_ln_modulate_kernel(const FluxRowArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int cpr = p.D >> 3;
  const int b = seg_batch(row, p.seg0, p.rows_per_batch0, p.rows_per_batch1);
  T* hrow = reinterpret_cast<T*>(p.h) + (size_t)row * p.D;
  float x[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < cpr) unpack8<T>(gload16(hrow + c * 8), x[i]);
  }
  if (p.n_src > 0) {
    for (int s = 0; s < p.n_src; ++s) {
      const FluxSrc d = p.srcs[s];
      const T* crow = reinterpret_cast<const T*>(d.cache) + (size_t)row * p.D;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < cpr) {
          float cf[8];
          unpack8<T>(gload16(crow + c * 8), cf);
          if (d.gate) {
            float gm[8];
            unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate) + (size_t)b * p.gate_stride + c * 8), gm);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] = round_elem<T>(x[i][e] + round_elem<T>(gm[e] * cf[e]));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] = round_elem<T>(x[i][e] + cf[e]);
          }
          if (d.clip16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] = clip_half<T>(x[i][e]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < cpr) gstore16(hrow + c * 8, pack8<T>(x[i]));
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

  const T* msh = reinterpret_cast<const T*>(p.shift) + (size_t)b * p.mod_stride;
  const T* msc = reinterpret_cast<const T*>(p.scale) + (size_t)b * p.mod_stride;
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < cpr) {
      float sh[8], sc[8];
      unpack8<T>(gload16(msh + c * 8), sh);
      unpack8<T>(gload16(msc + c * 8), sc);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x[i][e] = round_elem<T>((x[i][e] - mean) * rstd * (1.0f + sc[e]) + sh[e]);
        amax = fmaxf(amax, fabsf(x[i][e]));
      }
    }
  }
  if (p.y8) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float scl = amax > 0.f ? amax / FP8_MAX : 1.0f;
    const float inv = 1.0f / scl;
    if (lane == 0) p.y8_scale[row] = scl;
    u32x2* d = reinterpret_cast<u32x2*>(p.y8 + (size_t)row * p.D);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < cpr)
        d[c] = (u32x2){pack_fp8x4(x[i][0] * inv, x[i][1] * inv, x[i][2] * inv, x[i][3] * inv), pack_fp8x4(x[i][4] * inv, x[i][5] * inv, x[i][6] * inv, x[i][7] * inv)};
    }
  } else {
    T* yrow = reinterpret_cast<T*>(p.y) + (size_t)row * p.D;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < cpr) gstore16(yrow + c * 8, pack8<T>(x[i]));
    }
  }
}

// flat gated replay: each thread owns one 16-byte chunk of the hidden states across all sources
template <typename T>
__global__ __launch_bounds__(256) void flux_replay_kernel(const FluxRowArgs p) {
  const int cpr = p.D >> 3;
  const size_t total = (size_t)p.M * cpr;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int row = (int)(idx / cpr), c = (int)(idx - (size_t)row * cpr);
    const int b = seg_batch(row, p.seg0, p.rows_per_batch0, p.rows_per_batch1);
    const size_t off = (size_t)row * p.D + c * 8;
    T* hp = reinterpret_cast<T*>(p.h) + off;
    float acc[8];
    unpack8<T>(gload16(hp), acc);
    for (int s = 0; s < p.n_src; ++s) {
      const FluxSrc d = p.srcs[s];
      float cf[8];
      unpack8<T>(gload16(reinterpret_cast<const T*>(d.cache) + off), cf);
      if (d.gate) {
        float gm[8];
        unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate) + (size_t)b * p.gate_stride + c * 8), gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = round_elem<T>(acc[e] + round_elem<T>(gm[e] * cf[e]));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = round_elem<T>(acc[e] + cf[e]);
      }
      if (d.clip16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = clip_half<T>(acc[e]);
      }
    }
    gstore16(hp, pack8<T>(acc));
  }
}

// R rows per thread (same chunk c of R consecutive rows of one sample and segment): the gate of (source, sample, column) is fetched once
// per R cache chunks and the cache stream is read non-temporally -- the PixArt replay went from 4.2 to 6.3 TB/s with this layout
// (elementwise.h).  Same arithmetic and rounding order as flux_replay_kernel.  Needs seg0, rows_per_batch0/1 and M divisible by R.
template <typename T, int R>
__global__ __launch_bounds__(256) void flux_replay_rows_kernel(const FluxRowArgs p) {
  const int cpr = p.D >> 3;
  const unsigned total = (unsigned)(p.M / R) * cpr;
  for (unsigned idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const unsigned g = idx / cpr, c = idx - g * cpr;
    const int row0 = (int)(g * R);
    const int b = seg_batch(row0, p.seg0, p.rows_per_batch0, p.rows_per_batch1);
    const size_t off = (size_t)row0 * p.D + c * 8;
    T* hp = reinterpret_cast<T*>(p.h) + off;
    float acc[R][8];
#pragma unroll
    for (int j = 0; j < R; ++j) unpack8<T>(gload16(hp + (size_t)j * p.D), acc[j]);
    for (int s = 0; s < p.n_src; ++s) {
      const FluxSrc d = p.srcs[s];
      const T* cp = reinterpret_cast<const T*>(d.cache) + off;
      Pack8 cv[R];
#pragma unroll
      for (int j = 0; j < R; ++j) cv[j].u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(cp + (size_t)j * p.D));
      if (d.gate) {
        float gm[8];
        unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate) + (size_t)b * p.gate_stride + c * 8), gm);
#pragma unroll
        for (int j = 0; j < R; ++j) {
          float cf[8];
          unpack8<T>(cv[j], cf);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j][e] = round_elem<T>(acc[j][e] + round_elem<T>(gm[e] * cf[e]));
        }
      } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          float cf[8];
          unpack8<T>(cv[j], cf);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j][e] = round_elem<T>(acc[j][e] + cf[e]);
        }
      }
      if (d.clip16) {
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j][e] = clip_half<T>(acc[j][e]);
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) gstore16(hp + (size_t)j * p.D, pack8<T>(acc[j]));
  }
}

// ---- q <- rope(rmsnorm(q) * wq), k <- rope(rmsnorm(k) * wk), in place on qkv[rows][3D] (q | k | v thirds).
// 16 lanes per (row, head, q|k): lane owns 8 consecutive head-dim elements = 4 RoPE pairs.  rope: [S][dh/2][2] = (cos, sin)
// indexed by the row's sequence position (text rows first).  Text rows use (wq_txt, wk_txt) when given (joint blocks:
// norm_added_q / norm_added_k), image rows (wq, wk).
template <typename T>
__global__ __launch_bounds__(256) void flux_qk_norm_rope_kernel(T* __restrict__ qkv, int rows, int heads, int dh, const float* __restrict__ wq,
                                                                const float* __restrict__ wk, const float* __restrict__ wq_txt,
                                                                const float* __restrict__ wk_txt, const float* __restrict__ rope, int seg0, int T_txt,
                                                                int N_img, float eps, float q_scale) {
  const int l16 = threadIdx.x & 15;
  const size_t item = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const size_t total = (size_t)rows * heads * 2;
  const bool live = item < total;
  const size_t it = live ? item : total - 1;
  const int which = (int)(it % 2);
  const int head = (int)((it / 2) % heads);
  const int row = (int)(it / (2 * (size_t)heads));
  const int D = heads * dh;
  const bool txt = row < seg0;
  const int pos = txt ? row % T_txt : T_txt + (row - seg0) % N_img;
  T* ptr = qkv + (size_t)row * 3 * D + (size_t)which * D + (size_t)head * dh + l16 * 8;
  float x[8];
  unpack8<T>(gload16(ptr), x);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float r = rsqrtf(ss / (float)dh + eps);
  const float* w = which == 0 ? (txt && wq_txt ? wq_txt : wq) : (txt && wk_txt ? wk_txt : wk);
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + l16 * 8), w1 = *reinterpret_cast<const f32x4*>(w + l16 * 8 + 4);
  const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = round_elem<T>(round_elem<T>(x[e] * r) * wv[e]);
  const float* cs = rope + ((size_t)pos * (dh / 2) + l16 * 4) * 2;
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs), c1 = *reinterpret_cast<const f32x4*>(cs + 4);
  const float cosv[4] = {c0[0], c0[2], c1[0], c1[2]}, sinv[4] = {c0[1], c0[3], c1[1], c1[3]};
  float o[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[2 * j] = cosv[j] * x[2 * j] - sinv[j] * x[2 * j + 1];
    o[2 * j + 1] = sinv[j] * x[2 * j] + cosv[j] * x[2 * j + 1];
  }
  if (which == 0 && q_scale != 0.f) {  // softmax scale * log2 e folded into Q before its one rounding (attention_w4.h)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= q_scale;
  }
  if (live) gstore16(ptr, pack8<T>(o));
}

// ---- flow-match Euler: x <- round(x + dt * v); also refreshes the model-dtype copy the next x_embedder GEMM reads
template <typename T>
__global__ void flux_euler_kernel(float* __restrict__ latents, const T* __restrict__ v, int64_t ldv, T* __restrict__ model_in, const float* __restrict__ dt, int rows,
                                  int cols) {
  const size_t total = (size_t)rows * cols;
  const float d = dt ? dt[0] : 0.f;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t r = idx / cols, c = idx - r * cols;
    float x = latents[idx];
    if (dt) x += d * (float)v[r * ldv + c];
    x = round_elem<T>(x);
    latents[idx] = x;
    model_in[idx] = (T)x;
  }
}

template <typename T> __global__ void flux_add3_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, T* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = (float)a[i];
    if (b) v = round_elem<T>(v + (float)b[i]);
    v = round_elem<T>(v + (float)c[i]);
    out[i] = (T)v;
  }
}

}  // namespace ecad
