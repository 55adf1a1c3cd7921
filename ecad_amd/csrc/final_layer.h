// final_layer.h -- `_create_output` of the transformer in one launch (pixart_transformer_2d_edited.py:338-371), pending cache replays included:
//   h += sum_i gate_i * cached_i   (what the skipped sub-layers at the end of the step contribute; cached_transformer_block.py:243-246, 317-320)
//   norm_out (LayerNorm, no affine)  ->  * (1 + scale) + shift   with   shift, scale = scale_shift_table[2, D] + embedded_timestep   (:340-346)
//   ->  proj_out: Linear(D, p * p * C_out)   (:347-349)   ->  unpatchify "nhwpqc->nchpwq"   (:352-371)
// As separate launches (ln_modulate 16.5 us + a 32-tile GEMM padded to 144 columns 20 us + unpatchify 4.8 us per step at Bt 32) this tail keeps an eighth of
// the chip busy: the projection has 32 real output columns.  Here a workgroup of four waves owns 16 token rows; wave w takes the 32-wide k-slices w, w + 4, ...
// (9 of 36 at D = 1152) as MFMA A fragments -- lane (row l15, g) holds k = 32 ks + 8 g ..+7 -- adds the pending replays in the LayerNorm kernel's order and
// rounds to the storage type as it does, the four waves exchange row sums through LDS (mean, then centred squares: the two-pass variance of the LayerNorm
// kernel), each normalises + modulates + rounds its slices (the values the separate LayerNorm wrote to HBM) and multiplies them by the matching slices of the
// 32 rows of proj_out.weight (B fragments straight from L2); the partial 16 x 32 products are summed through LDS in wave order, the bias added, and the result
// scattered into the [Bt, C_out, H, W] noise prediction.  The hidden states are not written back: nothing reads them after the last layer of a step.
// A first version with one wave per 16 rows (512 waves, two per CU, each a chain of dependent L2 round trips) took 41-44 us; this one has 2 048 waves.
#pragma once
#include "common.h"
#include "elementwise.h"

namespace ecad {

struct FinalProj {
  const void* w;      // proj_out.weight [p * p * C_out = 32][D]
  const float* bias;  // [32]
  void* out;          // [Bt, C_out, gh * P, gw * P]
  int gh, gw, P, Cout;
};

// f: the LayerNorm's arguments (h, pending sources, table_shift / table_scale, mod_shift = mod_scale = embedded timestep rows, eps); f.y is not used.
template <typename T, int KSTEPS>
__global__ __launch_bounds__(256) void final_layer_kernel(const FlushArgs f, const FinalProj q) {
  constexpr int D = KSTEPS * 32, NW = 4, PER = KSTEPS / NW;
  static_assert(KSTEPS % NW == 0, "k-slices split evenly over the four waves");
  __shared__ float red[2][NW][16];
  __shared__ f32x4 accs[NW][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int m_base = blockIdx.x * 16;
  const int m = m_base + l15 < f.M ? m_base + l15 : f.M - 1;  // rows past the end repeat the last one and are not stored
  const int b = m / f.rows_per_batch;
  const int kw = 32 * wave + 8 * g;  // this lane's column inside its wave's first k-slice; slice i starts 128 i further
  const T* hrow = reinterpret_cast<const T*>(f.h) + (size_t)m * D + kw;
  float x[PER][8];
#pragma unroll
  for (int i = 0; i < PER; ++i) unpack8<T>(gload16(hrow + 128 * i), x[i]);
  if (f.n_src > 0) {
    for (int s = 0; s < f.n_src; ++s) {
      const SrcDesc d = f.srcs[s];
      const T* crow = reinterpret_cast<const T*>(d.cache) + (size_t)m * D + kw;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        float cf[8];
        unpack8<T>(gload16(crow + 128 * i), cf);
        if (d.gate_table) {
          float gm[8];
          unpack8<T>(gload16(reinterpret_cast<const T*>(d.gate_mod) + (size_t)b * f.mod_stride + kw + 128 * i), gm);
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(d.gate_table + kw + 128 * i), t1 = *reinterpret_cast<const f32x4*>(d.gate_table + kw + 128 * i + 4);
          const float gt[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) x[i][e] += (gt[e] + gm[e]) * cf[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[i][e] += cf[e];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) x[i][e] = round_elem<T>(x[i][e]);  // the LayerNorm kernel stores the updated hidden states before it normalises them
  }
  // modulation and weight slices: independent of the statistics, requested before the exchanges
  const T* emb = reinterpret_cast<const T*>(f.mod_shift) + (size_t)b * f.ln_mod_stride + kw;
  const T* w0 = reinterpret_cast<const T*>(q.w) + (size_t)l15 * D + kw;  // B fragments: lane (column l15, g) holds W[column][k ..+7]
  const T* w1 = w0 + (size_t)16 * D;
  Pack8 mo[PER], wf0[PER], wf1[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    mo[i] = gload16(emb + 128 * i);
    wf0[i] = gload16(w0 + 128 * i);
    wf1[i] = gload16(w1 + 128 * i);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += x[i][e];
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  if (g == 0) red[0][wave][l15] = sum;
  __syncthreads();
  const float mean = (red[0][0][l15] + red[0][1][l15] + red[0][2][l15] + red[0][3][l15]) / (float)D;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dlt = x[i][e] - mean;
      var += dlt * dlt;
    }
  var += __shfl_xor(var, 16);
  var += __shfl_xor(var, 32);
  if (g == 0) red[1][wave][l15] = var;
  __syncthreads();
  const float rstd = rsqrtf((red[1][0][l15] + red[1][1][l15] + red[1][2][l15] + red[1][3][l15]) / (float)D + f.eps);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int k0 = kw + 128 * i;
    float me[8];
    unpack8<T>(mo[i], me);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(f.table_scale + k0), c1 = *reinterpret_cast<const f32x4*>(f.table_scale + k0 + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(f.table_shift + k0), s1 = *reinterpret_cast<const f32x4*>(f.table_shift + k0 + 4);
    const float tsc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, tsh[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (x[i][e] - mean) * rstd * (1.0f + tsc[e] + me[e]) + (tsh[e] + me[e]);
    const Pack8 af = pack8<T>(y);
    acc0 = mfma16<T>(af, wf0[i], acc0);
    acc1 = mfma16<T>(af, wf1[i], acc1);
  }
  accs[wave][0][lane] = acc0;
  accs[wave][1][lane] = acc1;
  __syncthreads();
  if (wave != 0) return;
  // acc[r] = out[row m_base + 4 g + r][column l15 (+ 16)]; column = (pr * P + pc) * C_out + c
  T* out = reinterpret_cast<T*>(q.out);
  const int H = q.gh * q.P, W = q.gw * q.P;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    const f32x4 a = ((accs[0][nf][lane] + accs[1][nf][lane]) + accs[2][nf][lane]) + accs[3][nf][lane];
    const int col = 16 * nf + l15;
    const int c = col % q.Cout, pq = col / q.Cout, pr = pq / q.P, pc = pq - pr * q.P;
    const float bv = q.bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mr = m_base + 4 * g + r;
      if (mr >= f.M) continue;
      const int bb = mr / f.rows_per_batch, tok = mr - bb * f.rows_per_batch;
      const int i = tok / q.gw, j = tok - i * q.gw;
      out[(((size_t)bb * q.Cout + c) * H + i * q.P + pr) * W + j * q.P + pc] = (T)(a[r] + bv);
    }
  }
}

}  // namespace ecad
