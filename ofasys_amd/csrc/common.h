// Shared device/host helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ofasys_amd.h"

namespace ofa {

constexpr int WAVE = 64;

// ---------------------------------------------------------------- error plumbing
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define OFA_REQUIRE(cond, code, ...)     \
  do {                                   \
    if (!(cond)) {                       \
      ::ofa::set_error(__VA_ARGS__);     \
      return (code);                     \
    }                                    \
  } while (0)

// ---------------------------------------------------------------- bf16 <-> fp32 (round-to-nearest-even, as torch)
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// fp32 -> bf16 is the hardware conversion (v_cvt_pk_bf16_f32, round-to-nearest-even, NaN preserved)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// fp16: the dtype the reference routes to its fused-softmax extensions (multihead_attention.py:83-91); _Float16 <-> float are the
// hardware conversions (v_cvt_f32_f16 / v_cvt_f16_f32, round-to-nearest-even)
typedef _Float16 f16_t;

template <typename T> struct Vec;  // 16-byte vector of T
template <> struct Vec<float> { static constexpr int N = 4; typedef float4 type; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; typedef uint4 type; };
template <> struct Vec<f16_t> { static constexpr int N = 8; typedef uint4 type; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
// the 8 (4) storage elements of a 16-byte register quad as floats -- kernels that load raw uint4 use this instead of load_vec
template <typename T> __device__ __forceinline__ void unpack16(const uint4& r, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& r, float* out) {
  out[0] = __uint_as_float(r.x); out[1] = __uint_as_float(r.y); out[2] = __uint_as_float(r.z); out[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& r, float* out) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(w[i] << 16);
    out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& r, float* out) {
  const f32x8_t f = __builtin_convertvector(__builtin_bit_cast(f16x8_t, r), f32x8_t);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = f[i];
}
// two floats -> one 32-bit pair of storage elements (16-bit types only)
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }

// load/store N consecutive elements (16-byte aligned) as floats
template <typename T> __device__ __forceinline__ void load_vec(const T* p, float* out);
template <> __device__ __forceinline__ void load_vec<float>(const float* p, float* out) {
  float4 v = *reinterpret_cast<const float4*>(p);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <> __device__ __forceinline__ void load_vec<bf16_t>(const bf16_t* p, float* out) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(w[i] << 16);
    out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void load_vec<f16_t>(const f16_t* p, float* out) {
  unpack16<f16_t>(*reinterpret_cast<const uint4*>(p), out);
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float* in);
template <> __device__ __forceinline__ void store_vec<f16_t>(f16_t* p, const float* in) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_f16x2(in[2 * i], in[2 * i + 1]);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
template <> __device__ __forceinline__ void store_vec<float>(float* p, const float* in) {
  *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
}
template <> __device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float* in) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(in[2 * i], in[2 * i + 1]);
  *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ld1<f16_t>(const f16_t* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<f16_t>(f16_t* p, float v) { *p = (f16_t)v; }
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// ---------------------------------------------------------------- element-type dispatch of the C-ABI entry points
#define OFA_DT_OK(dt) ((dt) == OFA_F32 || (dt) == OFA_BF16 || (dt) == OFA_F16)
__host__ __device__ inline int dt_vecn(int dt) { return dt == OFA_F32 ? 4 : 8; }   // elements per 16-byte vector

// ---------------------------------------------------------------- wave64 reductions
// DPP data sharing instead of __shfl_xor: a shuffle compiles to ds_bpermute_b32 -- an LDS-pipe instruction with ~100+
// cycles of latency, six of them back to back per reduction -- while the DPP forms are ordinary VALU operands (quad_perm,
// row_half_mirror, row_mirror inside a 16-lane row; row_bcast15 / row_bcast31 across the four rows, gfx9 encodings).  The
// row kernels (LayerNorm, residual join, softmax, criterion) do 2-6 wave reductions per row and were latency-bound on them.
// Summation order is fixed (deterministic); every lane receives the result (v_readlane of lane 63).
#ifndef OFA_WAVE_REDUCE_SHFL
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1, 0xf>(0.f, v);          // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xf>(0.f, v);          // quad_perm [2,3,0,1]  -> every quad holds its sum
  v += dpp_f<0x141, 0xf>(0.f, v);         // row_half_mirror      -> every 8 lanes
  v += dpp_f<0x140, 0xf>(0.f, v);         // row_mirror           -> every row of 16
  v += dpp_f<0x142, 0xa>(0.f, v);         // row_bcast15 into rows 1, 3
  v += dpp_f<0x143, 0xc>(0.f, v);         // row_bcast31 into rows 2, 3 -> lanes 48..63 hold the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x4E, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x141, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x140, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(v, v));  // (lanes outside the row mask keep `old` = v: max(v, v) = v)
  v = fmaxf(v, dpp_f<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#else
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif

// ---------------------------------------------------------------- Philox4x32-10 (counter-based dropout masks)
// 32 x 32 -> 64-bit product in ONE quarter-rate instruction (hipcc splits `(uint64_t)a * b` with a constant operand into v_mul_hi_u32 +
// v_mul_lo_u32, two quarter-rate instructions: the generator is 40 such products per call and the residual-join kernels are bound by it)
__device__ __forceinline__ uint64_t mul_wide(uint32_t k, uint32_t x) {
  uint64_t r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "s"(k), "v"(x));
  return r;
}
struct Philox {
  uint32_t k0, k1;
  __device__ __forceinline__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __device__ __forceinline__ uint4 operator()(uint64_t ctr) const {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t p0 = mul_wide(0xD2511F53u, c0), p1 = mul_wide(0xCD9E8D57u, c2);
      const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
      uint32_t n0 = hi1 ^ c1 ^ a, n1 = lo1, n2 = hi0 ^ c3 ^ b, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
  }
};
// Dropout masks: element e draws 16 bits -- Philox counter (offset + e/8), halfword e%8 of the 128-bit output -- and is
// dropped iff they are below round(p * 65536).  One Philox call (40 32-bit multiplies) decides 8 elements: with 32 bits
// per element the residual-join kernels were VALU-bound on the generator, not on HBM.  The drop probability is exact to
// 2^-17 (p = 0.1 -> 0.100006).
__device__ __forceinline__ uint32_t philox_thresh(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ bool philox_keep(const uint4& r, int j, uint32_t thresh) {     // j = e % 8
  const uint32_t w = (j >> 1) == 0 ? r.x : ((j >> 1) == 1 ? r.y : ((j >> 1) == 2 ? r.z : r.w));
  return ((j & 1 ? w >> 16 : w & 0xffffu)) >= thresh;
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float kInvSqrt2Pi = 0.39894228040143267794f;
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}

// Gaussian cdf Phi(x) and e = exp(-x^2/2) together (gelu = x*Phi, gelu' = Phi + x*e/sqrt(2 pi)).  EXACT: libm erff (fp32
// kernels).  Otherwise Abramowitz-Stegun 7.1.26 on the SAME exponential: |error| < 1.5e-7 absolute, one v_exp + one v_rcp
// + 6 FMAs instead of erff's ~40 instructions -- the fused GELU+LayerNorm kernels were VALU-bound on erff (14336 x 3072:
// 94 us of issue cycles against 41 us of HBM time); used by the bf16 kernels, whose 2^-9 output rounding is four orders
// of magnitude coarser.  The tail uses 0.5*poly*e directly (no 1 - erf cancellation for negative x).
template <bool EXACT> __device__ __forceinline__ void gauss_cdf(float x, float& cdf, float& e) {
  e = __expf(-0.5f * x * x);
  if (EXACT) {
    cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return;
  }
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float q = 0.5f * poly * e;
  cdf = x > 0.f ? 1.0f - q : q;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Workgroups are dispatched in linear order (x fastest, then y, then z) round-robin over the 8 XCDs, and each XCD has its own L2.
// xcd_remap is the bijective "each XCD gets a contiguous chunk of the n logical ids" map: workgroups whose logical ids are
// neighbours (the tiles of one GEMM operand panel, the query tiles of one (sample, head)) then run on the SAME XCD and share its L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, xcd = id & 7, local = id >> 3;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + local;
}

}  // namespace ofa
