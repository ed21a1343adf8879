// Small attention-side kernels: rowsum(dO*O) for the fused backward, head-mean of attention weights, [B,T,C]->[B,C,Tpad].
#include "common.h"

namespace ofa {
constexpr int HD = 64;

// ------------------------------------------------------------------------------------------------ backward: delta
// delta[bh, q] = sum_d dout[q,h,d] * out[q,h,d]   (row-sum of dO*O; invariant under the c_attn output scale)
template <typename E>
__global__ __launch_bounds__(256) void attn_delta_kernel(const E* __restrict__ dout, const E* __restrict__ out,
                                                         float* __restrict__ delta, int B, int heads, int T, int Tpad,
                                                         int64_t ldo) {
  // 8 lanes per (row, head): each lane 8 elements
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t item = gid >> 3;
  const int sub = (int)(gid & 7);
  const int64_t total = (int64_t)B * T * heads;
  float s = 0.f;
  int64_t row = 0;
  int h = 0;
  if (item < total) {
    row = item / heads;
    h = (int)(item % heads);
    float x[8], y[8];
    load_vec<E>(dout + row * ldo + h * HD + sub * 8, x);
    load_vec<E>(out + row * ldo + h * HD + sub * 8, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j] * y[j];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (item < total && sub == 0) {
    const int64_t b = row / T, t = row % T;
    delta[(b * heads + h) * Tpad + t] = s;
  }
}

// ------------------------------------------------------------------------------------------------ [B,T,C] -> [B,C,Tpad]
template <typename T>
__global__ __launch_bounds__(256) void transpose_heads_kernel(const T* __restrict__ x, T* __restrict__ xt, int Tn, int C,
                                                             int Tpad, int64_t ld) {
  __shared__ T tile[64][66];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int t = t0 + r, c = c0 + tx;
    T v = 0;
    if (t < Tn && c < C) v = x[((int64_t)b * Tn + t) * ld + c];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, t = t0 + tx;
    if (c < C && t < Tpad) xt[((int64_t)b * C + c) * Tpad + t] = tile[tx][r];
  }
}

// ------------------------------------------------------------------------------------------------ backward: d c_attn
// dc[h] (+)= sum_b sum_{t<T} delta[(b*heads+h)*ld + t] / c[h]: one 1024-thread block per head, fixed summation order.
template <typename T>
__global__ __launch_bounds__(1024) void c_attn_grad_kernel(const float* __restrict__ delta, const T* __restrict__ c,
                                                           T* __restrict__ dc, int B, int heads, int Tq, int64_t ld,
                                                           int accumulate) {
  __shared__ float sw[16];
  const int h = blockIdx.x;
  float s = 0.f;
  const int64_t n = (int64_t)B * Tq;                  // one flat index over (b, t): independent loads, fixed order
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const int64_t b = i / Tq, t = i - b * Tq;
    s += delta[(b * heads + h) * ld + t];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += sw[i];
    tot /= ld1<T>(c + h);
    if (accumulate) tot += ld1<T>(dc + h);
    st1<T>(dc + h, tot);
  }
}
}  // namespace ofa
using namespace ofa;

extern "C" int ofa_attn_bwd_prep(const void* dout, const void* out, float* delta, int B, int heads, int T, int Tpad,
                                 int64_t ldo, int dtype, void* stream) {
  OFA_REQUIRE(dtype == OFA_BF16 || dtype == OFA_F16, OFA_ERR_UNSUPPORTED, "attn_bwd_prep: bf16 / fp16 only");
  OFA_REQUIRE(dout && out && delta && (ldo % 8) == 0 && Tpad >= T, OFA_ERR_INVALID, "attn_bwd_prep: bad argument");
  const int64_t threads = (int64_t)B * T * heads * 8;
  if (dtype == OFA_BF16)
    hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, dim3(cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                       (const bf16_t*)out, delta, B, heads, T, Tpad, ldo);
  else
    hipLaunchKernelGGL(attn_delta_kernel<f16_t>, dim3(cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)dout,
                       (const f16_t*)out, delta, B, heads, T, Tpad, ldo);
  return check_launch("attn_bwd_prep");
}

// out[b][i] = mean_a p[b][a][i]  (head-averaged attention weights, multihead_attention.py:347-351)
namespace ofa {
template <typename T>
__global__ __launch_bounds__(256) void mean_heads_kernel(const T* __restrict__ p, T* __restrict__ out, int heads, int64_t n) {
  const int b = blockIdx.y;
  const float inv = 1.0f / (float)heads;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int a = 0; a < heads; ++a) s += ld1<T>(p + ((int64_t)b * heads + a) * n + i);
    st1<T>(out + (int64_t)b * n + i, s * inv);
  }
}
}  // namespace ofa

extern "C" int ofa_mean_heads(const void* p, void* out, int B, int heads, int64_t n, int dtype, void* stream) {
  OFA_REQUIRE(p && out && B > 0 && heads > 0 && n > 0, OFA_ERR_INVALID, "mean_heads: bad argument");
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "mean_heads: bad dtype %d", dtype);
  int64_t gx = (n + 255) / 256;
  dim3 grid((unsigned)(gx > 1024 ? 1024 : gx), B), block(256);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((mean_heads_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)p, (float*)out, heads, n);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((mean_heads_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)p, (bf16_t*)out, heads, n);
  else
    hipLaunchKernelGGL((mean_heads_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, (const f16_t*)p, (f16_t*)out, heads, n);
  return check_launch("mean_heads");
}

extern "C" int ofa_transpose_heads(const void* x, void* xt, int B, int T, int C, int Tpad, int64_t ld, int dtype,
                                   void* stream) {
  OFA_REQUIRE(x && xt && B > 0 && T > 0 && C > 0 && Tpad >= T && ld >= C, OFA_ERR_INVALID, "transpose_heads: bad argument");
  dim3 grid(cdiv(Tpad, 64), cdiv(C, 64), B), block(256);
  if (dtype != OFA_F32)                                      // (a pure 16-bit move: bf16 and fp16 alike)
    hipLaunchKernelGGL((transpose_heads_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)xt,
                       T, C, Tpad, ld);
  else
    hipLaunchKernelGGL((transpose_heads_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)xt, T,
                       C, Tpad, ld);
  return check_launch("transpose_heads");
}

extern "C" int ofa_c_attn_grad(const float* delta, const void* c_attn, void* dc, int B, int heads, int T, int64_t ld,
                               int accumulate, int c_attn_dtype, void* stream) {
  OFA_REQUIRE(delta && c_attn && dc && B > 0 && heads > 0 && T > 0 && ld >= T, OFA_ERR_INVALID, "c_attn_grad: bad argument");
  OFA_REQUIRE(OFA_DT_OK(c_attn_dtype), OFA_ERR_INVALID, "c_attn_grad: bad dtype %d", c_attn_dtype);
  hipStream_t st = (hipStream_t)stream;
  if (c_attn_dtype == OFA_F32)
    hipLaunchKernelGGL((c_attn_grad_kernel<float>), dim3(heads), dim3(1024), 0, st, delta, (const float*)c_attn, (float*)dc, B, heads,
                       T, ld, accumulate);
  else if (c_attn_dtype == OFA_BF16)
    hipLaunchKernelGGL((c_attn_grad_kernel<bf16_t>), dim3(heads), dim3(1024), 0, st, delta, (const bf16_t*)c_attn, (bf16_t*)dc, B,
                       heads, T, ld, accumulate);
  else
    hipLaunchKernelGGL((c_attn_grad_kernel<f16_t>), dim3(heads), dim3(1024), 0, st, delta, (const f16_t*)c_attn, (f16_t*)dc, B,
                       heads, T, ld, accumulate);
  return check_launch("c_attn_grad");
}
