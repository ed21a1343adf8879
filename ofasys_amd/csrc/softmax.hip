// Row softmax family for gfx950, wave64: one wavefront per row, fp32 math, row held in registers (sk <= 4096).
//
// These are re-derivations (not hipify output) of the reference's three CUDA extensions, SURVEY.md section 2a:
//   scaled_softmax                      fused_kernels/scaled_masked_softmax.h:98-201   (fwd), :329-423 (bwd)
//   scaled_masked_softmax               scaled_masked_softmax.h:209-327
//   scaled_upper_triang_masked_softmax  scaled_upper_triang_masked_softmax.h:113-230
// plus the slow-path attention softmax of multihead_attention.py:311-334 (bias + causal + key-padding, fp32).
// The reference maps one 32-lane warp to WARP_BATCH rows; here one 64-lane wave owns one row and reduces with
// cross-lane shuffles (5+1 butterfly steps).
#include "common.h"

namespace ofa {

constexpr int SM_MAX_PER_LANE = 64;  // 64 lanes * 64 = 4096 columns

enum { SM_PLAIN = 0, SM_MASKED = 1, SM_CAUSAL = 2, SM_ATTN = 3 };

template <typename T, int MODE, int NI>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* __restrict__ x, const T* __restrict__ bias,
                                                          const uint8_t* __restrict__ mask, T* __restrict__ y,
                                                          float scale, int64_t rows, int sq, int sk, int np, int mask_b,
                                                          int causal) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * sk;
  T* yr = y + row * sk;
  const int q = (int)(row % sq);         // query index inside the [sq, sk] matrix
  const int64_t ab = row / sq;           // attention batch (b*np + head)
  const int b = (int)(ab / np);
  const uint8_t* mr = nullptr;
  if (MODE == SM_MASKED) mr = mask + ((int64_t)(mask_b == 1 ? 0 : b) * sq + q) * sk;   // [mask_b,1,sq,sk]
  if (MODE == SM_ATTN && mask) mr = mask + (int64_t)b * sk;                            // key padding [B,S]
  const T* br = (MODE == SM_ATTN && bias) ? bias + row * sk : nullptr;
  float v[NI];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = i * 64 + lane;
    float t = -INFINITY;
    if (c < sk) {
      t = ld1<T>(xr + c) * scale;
      if (MODE == SM_MASKED) { if (mr[c]) t = -10000.0f; }
      if (MODE == SM_CAUSAL) { if (c > q) t = -INFINITY; }
      if (MODE == SM_ATTN) {
        if (br) t += ld1<T>(br + c);
        if (causal) { if (isnan(t)) t = 0.f; if (c > q) t = -INFINITY; }  // nan_to_num then += triu(-inf)
        if (mr && mr[c]) t = -INFINITY;
      }
    }
    v[i] = t;
    mx = fmaxf(mx, t);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = i * 64 + lane;
    float e = 0.f;
    if (c < sk) e = (MODE == SM_CAUSAL && c > q) ? 0.f : expf(v[i] - mx);
    v[i] = e;
    s += e;
  }
  s = wave_sum(s);
  const float inv = 1.0f / s;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = i * 64 + lane;
    if (c < sk) st1<T>(yr + c, v[i] * inv);
  }
}

// dy and dx are NOT __restrict__: the reference's backward is "completely in place" on the output-gradient buffer
// (scaled_masked_softmax_cuda.cu:98-117, scaled_upper_triang_masked_softmax_cuda.cu:68-95) and callers may pass dx == dy; each
// lane reads all of its own elements of the row before the first store, so aliasing is safe.
// causal != 0 (sq == sk): columns above the diagonal are never read and their gradient is written as zero
// (scaled_upper_triang_masked_softmax.h:232-329 -- the forward's outputs are zero there).
template <typename T, int NI>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* dy, const T* __restrict__ y, T* dx, float scale,
                                                          int64_t rows, int sk, int causal) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* dyr = dy + row * sk;
  const T* yr = y + row * sk;
  T* dxr = dx + row * sk;
  const int last = causal ? (int)(row % sk) : sk - 1;       // last column with a non-zero probability
  float g[NI], p[NI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = i * 64 + lane;
    g[i] = p[i] = 0.f;
    if (c <= last) {
      p[i] = ld1<T>(yr + c);
      g[i] = ld1<T>(dyr + c) * p[i];
      s += g[i];
    }
  }
  s = wave_sum(s);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = i * 64 + lane;
    if (c < sk) st1<T>(dxr + c, scale * (g[i] - p[i] * s));
  }
}

template <int MODE>
static int softmax_launch(const void* x, const void* bias, const uint8_t* mask, void* y, float scale, int64_t rows,
                          int sq, int sk, int np, int mask_b, int causal, int dtype, hipStream_t st) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "softmax: bad dtype %d", dtype);
  OFA_REQUIRE(sk > 0 && sk <= 64 * SM_MAX_PER_LANE, OFA_ERR_UNSUPPORTED, "softmax: sk=%d must be in [1,4096]", sk);
  OFA_REQUIRE(x && y, OFA_ERR_INVALID, "softmax: null pointer");
  if (rows == 0) return 0;
  dim3 grid(cdiv(rows, 4)), block(256);
  const int ni = (sk + 63) / 64;
#define SM_CASE(NI)                                                                                                  \
  do {                                                                                                               \
    if (dtype == OFA_F32)                                                                                            \
      hipLaunchKernelGGL((softmax_fwd_kernel<float, MODE, NI>), grid, block, 0, st, (const float*)x,                 \
                         (const float*)bias, mask, (float*)y, scale, rows, sq, sk, np, mask_b, causal);              \
    else if (dtype == OFA_BF16)                                                                                      \
      hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t, MODE, NI>), grid, block, 0, st, (const bf16_t*)x,               \
                         (const bf16_t*)bias, mask, (bf16_t*)y, scale, rows, sq, sk, np, mask_b, causal);            \
    else                                                                                                             \
      hipLaunchKernelGGL((softmax_fwd_kernel<f16_t, MODE, NI>), grid, block, 0, st, (const f16_t*)x,                 \
                         (const f16_t*)bias, mask, (f16_t*)y, scale, rows, sq, sk, np, mask_b, causal);              \
  } while (0)
  if (ni <= 1) SM_CASE(1);
  else if (ni <= 2) SM_CASE(2);
  else if (ni <= 4) SM_CASE(4);
  else if (ni <= 8) SM_CASE(8);
  else if (ni <= 16) SM_CASE(16);
  else if (ni <= 32) SM_CASE(32);
  else SM_CASE(64);
#undef SM_CASE
  return check_launch("softmax_fwd");
}

}  // namespace ofa
using namespace ofa;

extern "C" int ofa_scaled_softmax_fwd(const void* x, void* y, float scale, int b, int np, int sq, int sk, int dtype,
                                      void* stream) {
  return softmax_launch<SM_PLAIN>(x, nullptr, nullptr, y, scale, (int64_t)b * np * sq, sq, sk, np, 0, 0, dtype,
                                  (hipStream_t)stream);
}

static int softmax_bwd_launch(const void* dy, const void* y, void* dx, float scale, int64_t rows, int sk, int causal,
                              int dtype, hipStream_t st) {
  OFA_REQUIRE(dtype == OFA_F32 || dtype == OFA_BF16 || dtype == OFA_F16, OFA_ERR_INVALID, "softmax_bwd: bad dtype %d", dtype);
  OFA_REQUIRE(sk > 0 && sk <= 64 * SM_MAX_PER_LANE, OFA_ERR_UNSUPPORTED, "softmax_bwd: sk=%d must be in [1,4096]", sk);
  OFA_REQUIRE(rows >= 0 && dy && y && dx, OFA_ERR_INVALID, "softmax_bwd: null pointer / negative size");
  if (rows == 0) return 0;
  dim3 grid(cdiv(rows, 4)), block(256);
  const int ni = (sk + 63) / 64;
#define SM_CASE(NI)                                                                                              \
  do {                                                                                                           \
    if (dtype == OFA_F32)                                                                                        \
      hipLaunchKernelGGL((softmax_bwd_kernel<float, NI>), grid, block, 0, st, (const float*)dy, (const float*)y, \
                         (float*)dx, scale, rows, sk, causal);                                                   \
    else if (dtype == OFA_BF16)                                                                                  \
      hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t, NI>), grid, block, 0, st, (const bf16_t*)dy,                \
                         (const bf16_t*)y, (bf16_t*)dx, scale, rows, sk, causal);                                \
    else                                                                                                         \
      hipLaunchKernelGGL((softmax_bwd_kernel<f16_t, NI>), grid, block, 0, st, (const f16_t*)dy,                  \
                         (const f16_t*)y, (f16_t*)dx, scale, rows, sk, causal);                                  \
  } while (0)
  if (ni <= 1) SM_CASE(1);
  else if (ni <= 2) SM_CASE(2);
  else if (ni <= 4) SM_CASE(4);
  else if (ni <= 8) SM_CASE(8);
  else if (ni <= 16) SM_CASE(16);
  else if (ni <= 32) SM_CASE(32);
  else SM_CASE(64);
#undef SM_CASE
  return check_launch("softmax_bwd");
}

extern "C" int ofa_scaled_softmax_bwd(const void* dy, const void* y, void* dx, float scale, int b, int np, int sq, int sk,
                                      int dtype, void* stream) {
  return softmax_bwd_launch(dy, y, dx, scale, (int64_t)b * np * sq, sk, 0, dtype, (hipStream_t)stream);
}

extern "C" int ofa_scaled_masked_softmax_bwd(const void* dy, const void* y, void* dx, float scale, int b, int np, int sq,
                                             int sk, int dtype, void* stream) {
  // scaled_masked_softmax.cpp:60-77 -> scaled_masked_softmax_cuda.cu:84-117: the mask does not enter the backward (masked
  // probabilities are ~exp(-10000) = 0 and carry their own zero)
  return softmax_bwd_launch(dy, y, dx, scale, (int64_t)b * np * sq, sk, 0, dtype, (hipStream_t)stream);
}

extern "C" int ofa_scaled_upper_triang_masked_softmax_bwd(const void* dy, const void* y, void* dx, float scale,
                                                          int attn_batches, int sq, int dtype, void* stream) {
  // scaled_upper_triang_masked_softmax.cpp:49-64 -> _cuda.cu:68-95: [attn_batches, sq, sq], zero above the diagonal
  return softmax_bwd_launch(dy, y, dx, scale, (int64_t)attn_batches * sq, sq, 1, dtype, (hipStream_t)stream);
}

extern "C" int ofa_scaled_masked_softmax_fwd(const void* x, const uint8_t* mask, void* y, float scale, int b, int np,
                                             int sq, int sk, int mask_b, int dtype, void* stream) {
  OFA_REQUIRE(mask, OFA_ERR_INVALID, "scaled_masked_softmax: null mask");
  OFA_REQUIRE(mask_b == 1 || mask_b == b, OFA_ERR_INVALID, "scaled_masked_softmax: mask batch %d must be 1 or %d",
              mask_b, b);
  return softmax_launch<SM_MASKED>(x, nullptr, mask, y, scale, (int64_t)b * np * sq, sq, sk, np, mask_b, 0, dtype,
                                   (hipStream_t)stream);
}

extern "C" int ofa_scaled_upper_triang_masked_softmax_fwd(const void* x, void* y, float scale, int attn_batches, int sq,
                                                          int dtype, void* stream) {
  return softmax_launch<SM_CAUSAL>(x, nullptr, nullptr, y, scale, (int64_t)attn_batches * sq, sq, sq, 1, 0, 1, dtype,
                                   (hipStream_t)stream);
}

extern "C" int ofa_attn_softmax_fwd(const void* x, const void* bias, const uint8_t* kpm, void* p, float scale, int BA,
                                    int heads, int T, int S, int causal, int dtype, void* stream) {
  OFA_REQUIRE(heads > 0 && BA % heads == 0, OFA_ERR_INVALID, "attn_softmax: BA=%d not a multiple of heads=%d", BA, heads);
  return softmax_launch<SM_ATTN>(x, bias, kpm, p, scale, (int64_t)BA * T, T, S, heads, 0, causal, dtype,
                                 (hipStream_t)stream);
}

extern "C" int ofa_get_batch_per_block(int sq, int sk, int b, int np) {
  // scaled_masked_softmax.h:426-438 -- the reference's 32-lane-warp launch arithmetic, kept so callers that size
  // their batches with it (fused_softmax.py:131-143) see the same answer.
  (void)sq; (void)b; (void)np;
  int log2 = 0;
  while ((1 << log2) < sk) ++log2;
  const int pow2 = 1 << log2;
  const int warp_size = pow2 < 32 ? pow2 : 32;
  const int batches_per_warp = pow2 <= 128 ? 2 : 1;
  return (128 / warp_size) * batches_per_warp;
}
