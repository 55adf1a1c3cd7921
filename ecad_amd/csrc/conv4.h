// conv4.h -- the decoder's convolutions on the one-wave-per-SIMD K loops of gemm4_asm.h (round 6; SURVEY 8f-1, reference: `self.vae.decode` inside the timed region of
// ecad/pipelines/pass_through.py:382-396).  conv3s_kernel (vae_kernels.h) gathers its A rows per tap with a per-lane address and a zero page for the padding, which a
// generated loop with one uniform A pointer cannot do.  Here the INPUT of a 3 x 3 convolution is laid out with a zero border -- [B][H + 2][W + 2][C], written by the
// GroupNorm apply pass or the upsample pass that produces it anyway -- so that
//   * the three taps of one kernel row are 3 C contiguous channels starting at pixel (y + dy, x - 1): a K segment of 3 C elements, no padding test;
//   * from one kernel row to the next every lane's source moves by the same (W + 2) C elements: a uniform jump of the A pointer every 3 C / 64 stages.
// (That is conv3s_kernel's K order -- kernel row, tap, channel -- kept as p.korder 1; the default since the PMC counters showed every input byte fetched three times is
// chunk of 64 channels, kernel row, tap: the same statement, another walk of its A pointer and a matching weight layout, see conv4_kernel.)
// The M index still runs over the B H W real pixels (whole 256-row tiles, the same tile counts as conv3s), so a lane's eight DMA pieces have gathered byte offsets
// (operands %[ao0..7] of the GEMM4_ASM_*G statements); outputs, residuals and 1 x 1 inputs stay plain [B H W][C].  K order per accumulator (row of taps, tap, channel,
// fp32) and the epilogue arithmetic (bias in fp32, round, C tile through LDS, residual added to the rounded value, round) are conv3s_kernel's: the outputs are its bytes.
//
// Upsampler convolutions (p.up == 2, four workgroups per tile: the phases): conv3x3(nearest2x(x)) at output pixel (2 y + a, 2 x + b) reads only the 2 x 2 input pixels (y - 1 + a .. y + a,
// x - 1 + b .. x + b), each through the SUM of the kernel taps that land on it -- four 2 x 2 convolutions over the low-resolution image with pre-summed weights
// (vae_phase_weights_kernel: fp32 sums of the fp16 taps, rounded once), 4/9 of the multiplications and no upsampled image in memory.  Same function, another rounding
// (weights summed before instead of products after): held to the oracle's tolerance, not to conv3s's bytes (ECAD_VAE_UP=nine keeps the nine-tap form).
#pragma once
#include "common.h"
#include "gemm4.h"
#include "vae_kernels.h"

#if defined(ECAD_PRODUCT_BUILD) && (defined(CONV4_PROBE_NO_EPILOGUE) || defined(CONV4_PROBE_NO_LOOP))
#error "CONV4_PROBE_* cut phases out of conv4_kernel for scripts/probe_conv4.hip: never in the product library"
#endif

namespace ecad {

// the statement takes its pointers in SGPRs: pin what the compiler may have computed per lane (tile coordinates are workgroup-uniform, hipcc does not always see it)
ECAD_DEV const char* conv4_uniform(const char* ptr) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
// the A pointer's walk (gen_gemm4.py, gathered A): da bytes per stage, db into stages kw and 2 kw of a cycle, the 64-bit dc into stage kt (= stage 0 of the next cycle)
struct Conv4Walk { unsigned da, db, dc, dch, kw, kw2, kt; };
template <typename T, int BN> ECAD_DEV void conv4_statement(const char* ap, const char* wp, unsigned rsw, unsigned trips, unsigned lds, unsigned wave, const Conv4Walk& wk,
                                                            const unsigned (&ao)[8]) {
  const unsigned zero = 1u;
  ap = conv4_uniform(ap);
  wp = conv4_uniform(wp);
  rsw = __builtin_amdgcn_readfirstlane(rsw);
  trips = __builtin_amdgcn_readfirstlane(trips);
  const unsigned da = __builtin_amdgcn_readfirstlane(wk.da), db = __builtin_amdgcn_readfirstlane(wk.db), dc = __builtin_amdgcn_readfirstlane(wk.dc),
                 dch = __builtin_amdgcn_readfirstlane(wk.dch), kw = __builtin_amdgcn_readfirstlane(wk.kw), kw2 = __builtin_amdgcn_readfirstlane(wk.kw2),
                 kt = __builtin_amdgcn_readfirstlane(wk.kt);
  if constexpr (std::is_same<T, bf16_t>::value && BN == 256) { GEMM4_ASM_BF16_256G(ap, wp, rsw, trips, lds, wave, zero, da, db, dc, dch, kw, kw2, kt, ao); }
  else if constexpr (std::is_same<T, bf16_t>::value) { GEMM4_ASM_BF16_128G(ap, wp, rsw, trips, lds, wave, zero, da, db, dc, dch, kw, kw2, kt, ao); }
  else if constexpr (BN == 256) { GEMM4_ASM_F16_256G(ap, wp, rsw, trips, lds, wave, zero, da, db, dc, dch, kw, kw2, kt, ao); }
  else { GEMM4_ASM_F16_128G(ap, wp, rsw, trips, lds, wave, zero, da, db, dc, dch, kw, kw2, kt, ao); }
}

// p.in: [B][H + 2][W + 2][Cin] with a zero border when p.taps == 9, plain [B H W][Cin] when p.taps == 1; B H W % 256 == 0, Cout % BN == 0, taps Cin / 64 a whole
// number of trips of the loop (vae.hip: conv4_bn)
// RES: out = round(round(conv + bias) + resid).  The tile's residual chunks are requested right behind the K loop, all at once (16 or 32 sixteen-byte loads per lane), so
// that they travel while the accumulators are staged: the row pass of conv3s_kernel asks for four at a time, which at one workgroup per CU is a latency-bound stream
// (the conv2 launches took 75-120 us longer than their conv1 twins, 268 MB of residual at ~2 TB/s).
template <typename T, int BN, bool RES>
__global__ __launch_bounds__(256) void conv4_kernel(const ConvArgs p) {
  constexpr int FM = 8, FN = BN / 32, CS = BN + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hs = p.up == 2 ? p.H >> 1 : p.H, Ws = p.up == 2 ? p.W >> 1 : p.W;   // the image the M index runs over (low resolution for the phase form)
  const int M = p.B * Hs * Ws;
  int tm, tn, phase = 0;
  {  // workgroups of one XCD (blockIdx % 8) take a contiguous run of tiles: neighbours share their halo rows (and, along n, all their A rows) in that XCD's L2
    const int tiles_n = p.Cout / BN, nph = p.up == 2 ? 4 : 1, nwg = (M / 256) * tiles_n * nph;
    int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.up == 2) {   // the four phases of a tile are neighbours in the run: they read the same 3 x 3 neighbourhoods and write the four pixels of the same 2 x 2 blocks
      phase = bid & 3;
      bid >>= 2;
    }
    tm = bid / tiles_n;
    tn = bid - tm * tiles_n;
  }
  const int pa = phase >> 1, pb = phase & 1;
  const int m0 = tm * 256, n0 = tn * BN;
  // p.tile2d: the 256 rows of a tile are a 16 x 16 patch of one image (row r = pixel (ty0 + r / 16, tx0 + r % 16)) instead of 256 consecutive pixels: 18 x 18 input
  // pixels under the nine taps instead of 3 (or 4, 6) whole image rows -- 2.4x fewer distinct A bytes per tile at W = 256 for the L2 to keep
  int tb = 0, ty0 = 0, tx0 = 0;
  if (p.tile2d) {
    const int tiles_x = Ws >> 4, tpi = (Hs >> 4) * tiles_x;
    tb = tm / tpi;
    const int t = tm - tb * tpi, trow = t / tiles_x;
    ty0 = trow << 4;
    tx0 = (t - trow * tiles_x) << 4;
  }
  const int K = (p.up == 2 ? 4 : p.taps) * p.Cin;
  unsigned ao[8];
  {
    // rows 8 (8 wave + q) + (lane >> 3), q = 0..7: one division pair for q = 0, then eight rows further per piece (two integer divisions per piece were 1.5 us of
    // a 22 us tile on the 18-stage convolutions)
    const int HW = Hs * Ws, Wp = Ws + 2;
    const int r0 = 64 * wave + (lane >> 3);
    int b = tb, y = 0, x = 0;
    if (p.taps == 9 && !p.tile2d) {
      const int m = m0 + r0;
      b = m / HW;
      const int rem = m - b * HW;
      y = rem / Ws;
      x = rem - y * Ws;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = r0 + 8 * q;
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      if (p.tile2d) { y = ty0 + (r >> 4); x = tx0 + (r & 15); }
      // 3 x 3: the pixel above-left of (y, x) in the bordered image, first tap of the first kernel row (phase form: (y - 1 + a, x - 1 + b) of the low-resolution image)
      const unsigned pix = p.taps == 9 ? (unsigned)((b * (Hs + 2) + y + pa) * Wp + x + pb) : (unsigned)(m0 + r);
      ao[q] = pix * (unsigned)(p.Cin * 2) + (unsigned)(c * 16);
      if (!p.tile2d) {
        x += 8;
        while (x >= Ws) { x -= Ws; ++y; }
        if (y >= Hs) { y -= Hs; ++b; }
      }
    }
  }
  // K order (p.korder 0, the default): 64-channel chunk, kernel row, tap -- the nine (four) visits of a tile to the same chunk of its pixels are consecutive stages, so the
  // chunk comes from HBM once (tap-major, p.korder 1 = conv3s_kernel's order kernel row, tap, channel: every input byte was fetched about three times, PMC: 16.5 GB of reads
  // per decode for 5.2 GB of inputs).  The weights are laid out to match (vae.hip).
  Conv4Walk wk;
  {
    const int kwid = p.up == 2 ? 2 : 3, khei = kwid;   // taps per kernel row, kernel rows
    const long long pix = (long long)p.Cin * 2, row = (long long)(Ws + 2) * pix;
    if (p.taps != 9) {                 // 1 x 1: plain rows, 128 bytes per stage
      wk = Conv4Walk{128u, 128u, 128u, 0u, 1u, 2u, 1u};
    } else if (p.korder == 1) {        // kernel row, tap, channel: kwid * Cin / 64 contiguous stages, then to the next image row; the cycle is the whole K loop
      const unsigned seg = (unsigned)(kwid * p.Cin / 64);
      wk = Conv4Walk{128u, (unsigned)(row - (kwid - 1) * pix - (pix - 128)), 128u, 0u, seg, 2 * seg, (unsigned)khei * seg};
    } else {                           // chunk, kernel row, tap
      const long long back = 128 - ((khei - 1) * row + (kwid - 1) * pix);
      wk = Conv4Walk{(unsigned)pix, (unsigned)(row - (kwid - 1) * pix), (unsigned)(unsigned long long)back, (unsigned)((unsigned long long)back >> 32), (unsigned)kwid, (unsigned)(2 * kwid),
                     (unsigned)(kwid * khei)};
    }
  }
  constexpr int period = G4Period<T, BN>::value;
#ifdef CONV4_PROBE_NO_LOOP
  if (p.bias == reinterpret_cast<const float*>(8))
#endif
  conv4_statement<T, BN>(reinterpret_cast<const char*>(p.in), conv4_uniform(reinterpret_cast<const char*>(p.w) + ((size_t)phase * p.Cout + n0) * K * sizeof(T)), (unsigned)(K * sizeof(T)),
                         (unsigned)(K / 64 / period - 1), __builtin_amdgcn_readfirstlane(lds_offset(smem)), (unsigned)wave, wk, ao);

#ifdef CONV4_PROBE_NO_EPILOGUE   // scripts/probe_conv4.hip only (timing of the tile's phases; the product build never defines it)
  if (p.bias != reinterpret_cast<const float*>(8)) return;
#endif
  // epilogue of conv3s_kernel on this wave geometry (2 x 2 waves of 128 x BN/2): the statement ends behind a barrier, the ring is free
  const int wm = wave & 1, wn = wave >> 1;
  const int l15 = lane & 15, g = lane >> 4;
  constexpr int CPR = BN / 8, RPP = 256 / CPR, NPASS = 256 / RPP;   // 16-byte chunks per tile row, rows per pass of the 256 threads, passes
  const int cchunk = tid % CPR, crow0 = tid / CPR;
  T* out = reinterpret_cast<T*>(p.out);
  const T* res = reinterpret_cast<const T*>(p.resid);
  // element offset of this thread's chunk of tile row ml = pass * RPP + crow0 in the plain [B][H][W][Cout] output (and residual)
  int ub = 0, uy = 0, ux = 0;   // phase form on consecutive pixels: low-resolution pixel of this thread's row of the current pass
  if (p.up == 2 && !p.tile2d) {
    const int m = m0 + crow0, HW = Hs * Ws;
    ub = m / HW;
    const int rem = m - ub * HW;
    uy = rem / Ws;
    ux = rem - uy * Ws;
  }
  size_t ooff[NPASS];
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const int ml = pass * RPP + crow0;
    size_t pix;
    if (p.tile2d) {
      const int y = ty0 + (ml >> 4), x = tx0 + (ml & 15);
      pix = p.up == 2 ? (size_t)(tb * p.H + 2 * y + pa) * p.W + 2 * x + pb : (size_t)(tb * Hs + y) * Ws + x;
    } else if (p.up == 2) {
      pix = (size_t)(ub * p.H + 2 * uy + pa) * p.W + 2 * ux + pb;
      ux += RPP;
      while (ux >= Ws) { ux -= Ws; ++uy; }
      if (uy >= Hs) { uy -= Hs; ++ub; }
    } else {
      pix = (size_t)(m0 + ml);
    }
    ooff[pass] = pix * p.Cout + n0 + cchunk * 8;
  }
  Pack8 rv[RES ? NPASS : 1];
  if constexpr (RES) {
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) rv[pass] = gload16(res + ooff[pass]);
  }
  T* ctile = reinterpret_cast<T*>(smem);
  static_for<FN>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int nl = wn * (BN / 2) + 16 * j + 4 * g;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + nl);
    static_for<FM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int ml = wm * 128 + 16 * i + l15;
      const f32x4 av = g4_acc<i, j>();
      Pack4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) float_to_elem4<T>(o, r, av[r] + bv[r]);
      *reinterpret_cast<u32x2*>(ctile + ml * CS + nl) = o.u;
    });
  });
  __syncthreads();
  f32x2 ss[4], sq[4];   // GroupNorm partial sums of this lane's eight channels over its NPASS rows (p.gn_part), channel pairs on the packed fp32 pipe
#pragma unroll
  for (int e = 0; e < 4; ++e) ss[e] = sq[e] = (f32x2){0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const int ml = pass * RPP + crow0;
    Pack8 v = lds_read16(ctile + ml * CS + cchunk * 8);
    if constexpr (RES) {
      float a[8], b[8];
      unpack8<T>(v, a);
      unpack8<T>(rv[pass], b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
      v = pack8<T>(a);
    }
    gstore16(out + ooff[pass], v);
    if (p.gn_part) {   // of the values as stored (rounded): what gn_stats_kernel would read back
      float f[8];
      unpack8<T>(v, f);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 f2 = (f32x2){f[2 * e], f[2 * e + 1]};
        ss[e] += f2;
        sq[e] += f2 * f2;
      }
    }
  }
  if (p.gn_part) {
    // fixed-order fold (gn_stats_kernel's scheme): every thread leaves (low half, high half) x (sum, sum of squares) of its chunk in LDS behind the C tile, thread
    // t < groups-of-the-tile adds its group's partials in row order (all reads first: a chain of dependent LDS round trips per tile is time a one-wave-per-SIMD
    // kernel cannot hide); one row of partials per 256-pixel tile, added in tile order by gn_finalize_kernel
    f32x4* part = reinterpret_cast<f32x4*>(smem + 256 * CS * sizeof(T));
    const f32x2 lo_s = ss[0] + ss[1], lo_q = sq[0] + sq[1], hi_s = ss[2] + ss[3], hi_q = sq[2] + sq[3];
    part[tid] = (f32x4){lo_s[0] + lo_s[1], lo_q[0] + lo_q[1], hi_s[0] + hi_s[1], hi_q[0] + hi_q[1]};
    __syncthreads();
    const int cg = p.Cout >> 5;   // channels per group (32 groups): 4, 8 or 16
    if (tid < BN / cg) {
      const int gl = tid;
      const int cc = cg == 4 ? gl >> 1 : gl * cg / 8;      // first (or only) chunk of the group
      const bool two = cg == 16, upper = cg == 4 && (gl & 1);
      f32x4 a[RPP], b[RPP];
#pragma unroll
      for (int r = 0; r < RPP; ++r) {
        a[r] = part[r * CPR + cc];
        b[r] = part[r * CPR + cc + (two ? 1 : 0)];
      }
      float ts = 0.f, tq = 0.f;
#pragma unroll
      for (int r = 0; r < RPP; ++r) {
        if (cg == 4) {
          ts += upper ? a[r][2] : a[r][0];
          tq += upper ? a[r][3] : a[r][1];
        } else {
          ts += a[r][0] + a[r][2];
          tq += a[r][1] + a[r][3];
          if (two) {
            ts += b[r][0] + b[r][2];
            tq += b[r][1] + b[r][3];
          }
        }
      }
      // tile tm = image b, tile (m0 % HW) / 256 of it: [b][tile][group][2] when HW % 256 == 0; phase form: [b][phase][tile of the low-resolution image]
      size_t row = (size_t)tm;
      if (p.up == 2) {
        const int tpi = (Hs * Ws) >> 8, b = tm / tpi;
        row = ((size_t)b * 4 + phase) * tpi + (tm - b * tpi);
      }
      *reinterpret_cast<f32x2*>(p.gn_part + (row * 32 + n0 / cg + gl) * 2) = (f32x2){ts, tq};
    }
  }
}
constexpr int CONV4_PART_BYTES = 256 * 16;   // LDS behind the C tile for the GroupNorm fold
template <int BN> constexpr int conv4_lds_bytes() { return (g4_lds_bytes<BN>() > 256 * (BN + 8) * 2 + CONV4_PART_BYTES) ? g4_lds_bytes<BN>() : 256 * (BN + 8) * 2 + CONV4_PART_BYTES; }

// GroupNorm apply (+ SiLU) from plain [B H W][C] into the zero-bordered layout [B][H + 2][W + 2][C] a conv4_kernel reads; the border chunks are written as zeros here,
// every time, so that the buffer carries no state from the geometry it held before
template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_pad_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                           size_t total_chunks, int H, int W, int C) {
  const int cpr = C >> 3, Wp = W + 2, HWp = (H + 2) * Wp;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total_chunks; idx += (size_t)gridDim.x * 256) {
    const size_t pp = idx / cpr;
    const int c = (int)(idx - pp * cpr) * 8;
    const int b = (int)(pp / HWp);
    const int rem = (int)(pp - (size_t)b * HWp);
    const int yy = rem / Wp, xx = rem - yy * Wp;
    Pack8 o;
    o.u = (u32x4){0u, 0u, 0u, 0u};
    if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
      float v[8];
      unpack8<T>(gload16(x + (((size_t)b * H + (yy - 1)) * W + (xx - 1)) * C + c), v);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c), s1 = *reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + (size_t)b * C + c), h1 = *reinterpret_cast<const f32x4*>(shift + (size_t)b * C + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = v[e] * s0[e] + h0[e];
        v[4 + e] = v[4 + e] * s1[e] + h1[e];
      }
      if (SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
      }
      o = pack8<T>(v);
    }
    gstore16(y + idx * 8, o);
  }
}

// plain [B][H >> sh][W >> sh][C] into the zero-bordered [B][H + 2][W + 2][C]: sh = 1, the nearest-2x upsampled image in front of an upsampler's nine-tap conv4_kernel;
// sh = 0, a bordered copy in front of its phase form
template <typename T>
__global__ __launch_bounds__(256) void upsample_pad_kernel(const T* __restrict__ x, T* __restrict__ y, size_t total_chunks, int H, int W, int C, int sh) {
  const int cpr = C >> 3, Wp = W + 2, HWp = (H + 2) * Wp, Hin = H >> sh, Win = W >> sh;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total_chunks; idx += (size_t)gridDim.x * 256) {
    const size_t pp = idx / cpr;
    const int c = (int)(idx - pp * cpr) * 8;
    const int b = (int)(pp / HWp);
    const int rem = (int)(pp - (size_t)b * HWp);
    const int yy = rem / Wp, xx = rem - yy * Wp;
    Pack8 o;
    o.u = (u32x4){0u, 0u, 0u, 0u};
    if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) o = gload16(x + (((size_t)b * Hin + ((yy - 1) >> sh)) * Win + ((xx - 1) >> sh)) * C + c);
    gstore16(y + idx * 8, o);
  }
}

// w9: [Cout][3][3][Cin] -> w4: [phase = 2 a + b][Cout][2][2][Cin]: tap row r of phase a sums kernel rows {0} / {1, 2} (a = 0) or {0, 1} / {2} (a = 1), columns likewise
template <typename T>
__global__ void vae_phase_weights_kernel(const T* __restrict__ w9, T* __restrict__ w4, int Cout, int Cin) {
  const size_t n = (size_t)4 * Cout * 4 * Cin;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int t = (int)((i / Cin) % 4), r = t >> 1, c = t & 1;
    const int co = (int)((i / ((size_t)4 * Cin)) % Cout);
    const int ph = (int)(i / ((size_t)4 * Cin * Cout)), a = ph >> 1, b = ph & 1;
    const int ky0 = a == 0 ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), ky1 = a == 0 ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
    const int kx0 = b == 0 ? (c == 0 ? 0 : 1) : (c == 0 ? 0 : 2), kx1 = b == 0 ? (c == 0 ? 0 : 2) : (c == 0 ? 1 : 2);
    float acc = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx) acc += (float)w9[((size_t)co * 9 + ky * 3 + kx) * Cin + ci];
    w4[i] = (T)acc;
  }
}

// src [rows][taps][Cin] -> dst [rows][Cin / 64][taps][64]: the weight layout of conv4_kernel's default K order (chunk of 64 channels, kernel row, tap)
template <typename T>
__global__ void vae_chunk_major_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t rows, int taps, int Cin) {
  const size_t n = rows * taps * Cin;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % 64);
    const int tap = (int)((i / 64) % taps);
    const int chunk = (int)((i / ((size_t)64 * taps)) % (Cin / 64));
    const size_t row = i / ((size_t)taps * Cin);
    dst[i] = src[(row * taps + tap) * Cin + chunk * 64 + c];
  }
}

}  // namespace ecad
