// common.h -- element types, vector types, MFMA / LDS-DMA wrappers for gfx950 (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace ecad {

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define ECAD_DEV __device__ __forceinline__

// compile-time unrolled loop with a constexpr index (needed by __builtin_amdgcn_sched_group_barrier and register rings)
template <int... Q, typename F> ECAD_DEV void static_for_impl(std::integer_sequence<int, Q...>, F&& f) { (f(std::integral_constant<int, Q>{}), ...); }
template <int N, typename F> ECAD_DEV void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <typename T> struct Vec8;
template <> struct Vec8<f16_t> { typedef f16x8 type; };
template <> struct Vec8<bf16_t> { typedef bf16x8 type; };

// 8 elements (16 bytes) as a raw register quad; all global/LDS traffic moves in these.
union Pack8 {
  u32x4 u;
  i16x8 s;
  f16x8 h;
  bf16x8 b;
  f16_t he[8];
  bf16_t be[8];
};
union Pack4 {
  u32x2 u;
  i16x4 s;
  f16_t he[4];
  bf16_t be[4];
};

template <typename T> ECAD_DEV float elem_to_float(const Pack8& p, int i);
template <> ECAD_DEV float elem_to_float<f16_t>(const Pack8& p, int i) { return (float)p.he[i]; }
template <> ECAD_DEV float elem_to_float<bf16_t>(const Pack8& p, int i) { return (float)p.be[i]; }
template <typename T> ECAD_DEV void float_to_elem(Pack8& p, int i, float v);
template <> ECAD_DEV void float_to_elem<f16_t>(Pack8& p, int i, float v) { p.he[i] = (f16_t)v; }
template <> ECAD_DEV void float_to_elem<bf16_t>(Pack8& p, int i, float v) { p.be[i] = (bf16_t)v; }
template <typename T> ECAD_DEV void float_to_elem4(Pack4& p, int i, float v);
template <> ECAD_DEV void float_to_elem4<f16_t>(Pack4& p, int i, float v) { p.he[i] = (f16_t)v; }
template <> ECAD_DEV void float_to_elem4<bf16_t>(Pack4& p, int i, float v) { p.be[i] = (bf16_t)v; }

// two floats -> one register of two storage elements, as ONE v_cvt_pk_{f16,bf16}_f32 (round to nearest even, what the scalar conversions do): written as a vector
// conversion because element-wise stores into a Pack4 came out as v_cvt + v_cvt (SDWA) + v_or per pair in the register-resident epilogues (FF1: 288 values per lane)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <typename T> ECAD_DEV unsigned pack2(float a, float b);
template <> ECAD_DEV unsigned pack2<f16_t>(float a, float b) {
  const f16x2_t h = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);
  return __builtin_bit_cast(unsigned, h);
}
template <> ECAD_DEV unsigned pack2<bf16_t>(float a, float b) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  return __builtin_bit_cast(unsigned, h);
}

template <typename T> ECAD_DEV void unpack8(const Pack8& p, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = elem_to_float<T>(p, i);
}
template <typename T> ECAD_DEV Pack8 pack8(const float (&f)[8]) {
  Pack8 p;
#pragma unroll
  for (int i = 0; i < 8; ++i) float_to_elem<T>(p, i, f[i]);
  return p;
}
// round-trip through the storage type (what the reference's half tensors do)
template <typename T> ECAD_DEV float round_elem(float v);
template <> ECAD_DEV float round_elem<f16_t>(float v) { return (float)(f16_t)v; }
template <> ECAD_DEV float round_elem<bf16_t>(float v) { return (float)(bf16_t)v; }

// fp16 saturation of the FLUX block outputs: `if hidden_states.dtype == torch.float16: hidden_states = hidden_states.clip(-65504, 65504)`
// (cached_flux_transformer_block.py:127-128,288-289): an overflowed sum (+-inf after rounding to fp16) becomes +-65504; bf16 is left alone; NaN stays NaN.
template <typename T> ECAD_DEV float clip_half(float v) {
  if constexpr (std::is_same<T, f16_t>::value) return v > 65504.0f ? 65504.0f : (v < -65504.0f ? -65504.0f : v);
  else return v;
}

// D(16x16) += A(16x32) * B(32x16); lane l holds A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15];
// D[row 4(l>>4)+r][col l&15] in element r.
template <typename T> ECAD_DEV f32x4 mfma16(const Pack8& a, const Pack8& b, f32x4 c);
template <> ECAD_DEV f32x4 mfma16<f16_t>(const Pack8& a, const Pack8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, c, 0, 0, 0);
}
template <> ECAD_DEV f32x4 mfma16<bf16_t>(const Pack8& a, const Pack8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// 16-byte global -> LDS DMA.  LDS destination = wave-uniform base + lane*16.
ECAD_DEV void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
}

// The same DMA as inline asm.  hipcc's waitcnt pass tracks the builtin as an LDS store and puts `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove
// disjoint from it -- every ds_read_b64_tr_b16 (the intrinsic carries no alias information), and plain reads of a ring slot picked at run time -- which turns a
// prefetch that should fly for a whole tile into a round trip in the middle of the tile before it (round 5: found in the ISA of attn_dma_kernel, where the wait sat
// in front of the first V fragment read of every chunk).  An asm statement is invisible to that pass: the kernel then owns the wait (its own counted
// `s_waitcnt vmcnt(N)` + barrier before anybody reads the slot).  M0 carries the LDS destination and is not preserved around asm by the compiler: saved and
// restored here (cdna_hip_programming.md 5.7).  lds_off = wave-uniform LDS byte address (lds_offset() of the slot pointer).
ECAD_DEV unsigned lds_offset(const void* p) { return (unsigned)(size_t)(lds_void*)p; }
ECAD_DEV void glds16_raw(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_off) : "memory");
}

ECAD_DEV Pack8 lds_read16(const void* p) {
  Pack8 r;
  r.u = *reinterpret_cast<const u32x4*>(p);
  return r;
}
ECAD_DEV Pack8 gload16(const void* p) {
  Pack8 r;
  r.u = *reinterpret_cast<const u32x4*>(p);
  return r;
}
ECAD_DEV void gstore16(void* p, const Pack8& v) { *reinterpret_cast<u32x4*>(p) = v.u; }
// cache-slot writes: read again one or more denoising steps later, never soon -- keep them out of the way of h / xn in L2 and the Infinity Cache
ECAD_DEV void gstore16_nt(void* p, const Pack8& v) { __builtin_nontemporal_store(v.u, reinterpret_cast<u32x4*>(p)); }

// 4 rows x 16 cols transposed LDS read (gfx950 ds_read_b64_tr_b16): lane 4q+p of a 16-lane group supplies
// the address of row q, cols 4p..4p+3; lane i receives column i of the 4 rows.  EXEC must be all ones.
ECAD_DEV Pack4 lds_read_tr16(const void* p) {
  Pack4 r;
  r.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)p);
  return r;
}

// gelu(x, approximate="tanh") = 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)
//                              = x * sigmoid(2u) = x / (1 + exp2(-x (a + b x^2)))
// 5 VALU + v_exp_f32 + v_rcp_f32 per element (the textbook form costs ~14 VALU + 2 transcendentals; the GELU
// epilogue of FF1 is 144 elements per lane and measured 8-14 % of that GEMM).
ECAD_DEV float gelu_tanh(float x) {
  const float a = 2.0f * 0.7978845608028654f * 1.4426950408889634f;
  const float b = a * 0.044715f;
  const float t = __builtin_fmaf(x * x, b, a);
  const float e = __builtin_amdgcn_exp2f(-x * t);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// two elements per VALU instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32; the two transcendentals stay scalar): for VALU-bound epilogues
// (FF1: 144 GELUs per lane, stamped 7.6k cycles per tile).  Same formula and operation order as gelu_tanh, so the values are identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));
ECAD_DEV f32x2 gelu_tanh2(f32x2 x) {
  const float a = 2.0f * 0.7978845608028654f * 1.4426950408889634f;
  const float b = a * 0.044715f;
  const f32x2 t = __builtin_elementwise_fma(x * x, (f32x2){b, b}, (f32x2){a, a});
  const f32x2 z = -x * t;
  const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + (f32x2){1.0f, 1.0f};
  return x * (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
// max over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} (the four 16-lane rows of a wave: the lanes that hold one query row's scores in the attention kernels) without
// the LDS: __shfl_xor compiles to ds_bpermute_b32, a round trip through the LDS queue behind every fragment read and DMA write of the workgroup (stamped: the
// row-maximum phase of a 64-key unit took ~1000 cycles).  v_permlane16_swap_b32 a, b exchanges the odd rows of a with the even rows of b, v_permlane32_swap_b32
// the upper half of a with the lower half of b (gfx950); with b a copy of a, max(a, b) afterwards is the xor-16 / xor-32 butterfly.  Inline asm because the
// clang builtin returns the first result for both elements here (ROCm 7.2); the s_nop covers the VALU-write -> permlane-read hazard the assembler cannot see.
ECAD_DEV float rows_max(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a = fmaxf(a, b);
  b = a;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
ECAD_DEV float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
ECAD_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

}  // namespace ecad
