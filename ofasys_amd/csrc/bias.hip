// Attention-bias assembly for gfx950 (HBM-bound): the per-layer, per-slot relative-position bias is added onto the
// diagonal block of the absolute-position bias (reference: adaptor/general.py:265-280,
//   self_attn_bias[:, :, s:e, s:e] += slot_bias   with slot_bias = values[T,T,A] expanded over batch, base.py:242-256).
// Forward adds in place on the caller's clone; backward reduces the block over the batch.
#include "common.h"

namespace ofa {

// bias[b][a][s+i][s+j] += values[i][j][a]        one thread per (b,a,i,j), j fastest (coalesced on bias)
template <typename T>
__global__ __launch_bounds__(256) void bias_block_add_kernel(T* __restrict__ bias, const T* __restrict__ values, int B, int A,
                                                             int Tt, int s, int n) {
  const int64_t total = (int64_t)B * A * n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % n);
    const int i = (int)((e / n) % n);
    const int a = (int)((e / ((int64_t)n * n)) % A);
    const int64_t b = e / ((int64_t)n * n * A);
    T* p = bias + (((b * A + a) * Tt) + s + i) * Tt + s + j;
    st1<T>(p, ld1<T>(p) + ld1<T>(values + ((int64_t)i * n + j) * A + a));
  }
}

// dvalues[i][j][a] = sum_b dbias[b][a][s+i][s+j]   one thread per (i,j,a); deterministic (b ascending)
template <typename T>
__global__ __launch_bounds__(256) void bias_block_grad_kernel(const T* __restrict__ dbias, T* __restrict__ dvalues, int B,
                                                              int A, int Tt, int s, int n) {
  const int64_t total = (int64_t)A * n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % n);
    const int i = (int)((e / n) % n);
    const int a = (int)(e / ((int64_t)n * n));
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += ld1<T>(dbias + ((((int64_t)b * A + a) * Tt) + s + i) * Tt + s + j);
    st1<T>(dvalues + ((int64_t)i * n + j) * A + a, acc);
  }
}

}  // namespace ofa
using namespace ofa;

static inline int bias_grid(int64_t work) {
  int64_t g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int ofa_bias_block_add(void* bias, const void* values, int B, int A, int T, int start, int n, int dtype,
                                  void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "bias_block_add: bad dtype %d", dtype);
  OFA_REQUIRE(bias && values && B > 0 && A > 0 && n > 0 && start >= 0 && start + n <= T, OFA_ERR_INVALID,
              "bias_block_add: bad argument (T=%d start=%d n=%d)", T, start, n);
  const int64_t total = (int64_t)B * A * n * n;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((bias_block_add_kernel<float>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (float*)bias, (const float*)values, B, A, T, start, n);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((bias_block_add_kernel<bf16_t>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)bias, (const bf16_t*)values, B, A, T, start, n);
  else
    hipLaunchKernelGGL((bias_block_add_kernel<f16_t>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (f16_t*)bias, (const f16_t*)values, B, A, T, start, n);
  return check_launch("bias_block_add");
}

extern "C" int ofa_bias_block_grad(const void* dbias, void* dvalues, int B, int A, int T, int start, int n, int dtype,
                                   void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "bias_block_grad: bad dtype %d", dtype);
  OFA_REQUIRE(dbias && dvalues && B > 0 && A > 0 && n > 0 && start >= 0 && start + n <= T, OFA_ERR_INVALID,
              "bias_block_grad: bad argument (T=%d start=%d n=%d)", T, start, n);
  const int64_t total = (int64_t)A * n * n;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((bias_block_grad_kernel<float>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dbias, (float*)dvalues, B, A, T, start, n);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((bias_block_grad_kernel<bf16_t>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dbias, (bf16_t*)dvalues, B, A, T, start, n);
  else
    hipLaunchKernelGGL((bias_block_grad_kernel<f16_t>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const f16_t*)dbias, (f16_t*)dvalues, B, A, T, start, n);
  return check_launch("bias_block_grad");
}
