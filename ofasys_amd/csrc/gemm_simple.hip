// Exact-tier GEMM: C = alpha * op(A) op(B) (+bias) (+C), fp32 FMA accumulation on the vector ALU.
// Used for fp32 parity runs (bit-for-bit an fp32 fmaf chain in k order per output element) and as the always-correct
// fallback for shapes the MFMA kernel does not take (unaligned leading dimensions).  64x64 tile, 256 threads,
// 4x4 outputs per thread, BK = 16, LDS tiles stored k-major so the inner product reads are conflict-free.
// Replaces torch.addmm/bmm at multihead_attention.py:199-217,308,338,346 and transformer_layer.py:194,202.
#include "gemm.h"

namespace ofa {

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gemm_simple_kernel(GemmArgs g) {
  constexpr int BM = 64, BN = 64, BK = 16;
  __shared__ float sA[BK][BM + 4];
  __shared__ float sB[BK][BN + 4];
  const int bz = blockIdx.z;
  const T* A = (const T*)g.A + batch_off(bz, g.batch_inner, g.strideA, g.strideA2);
  const T* B = (const T*)g.B + batch_off(bz, g.batch_inner, g.strideB, g.strideB2);
  TO* C = (TO*)g.C + batch_off(bz, g.batch_inner, g.strideC, g.strideC2);
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, each 4 (m) x 4 (n)
  float acc[4][4] = {};
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    // stage: 64*16 = 1024 elements per operand, 4 per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + 256 * i;
      {
        // A tile element (m = e % 64 or e / 16 depending on contiguity)
        int m, k;
        if (g.transA) { m = e & 63; k = e >> 6; } else { k = e & 15; m = e >> 4; }
        float v = 0.f;
        if (m0 + m < g.M && k0 + k < g.K)
          v = g.transA ? ld1<T>(A + (int64_t)(k0 + k) * g.lda + (m0 + m)) : ld1<T>(A + (int64_t)(m0 + m) * g.lda + (k0 + k));
        sA[k][m] = v;
      }
      {
        int n, k;
        if (g.transB) { k = e & 15; n = e >> 4; } else { n = e & 63; k = e >> 6; }
        float v = 0.f;
        if (n0 + n < g.N && k0 + k < g.K)
          v = g.transB ? ld1<T>(B + (int64_t)(n0 + n) * g.ldb + (k0 + k)) : ld1<T>(B + (int64_t)(k0 + k) * g.ldb + (n0 + n));
        sB[k][n] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sA[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = sB[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= g.N) continue;
      float v = acc[i][j];
      if (g.flags & OFA_GEMM_BIAS_COL) v += ld1<T>((const T*)g.bias + n);
      if (g.flags & OFA_GEMM_BIAS_ROW) v += ld1<T>((const T*)g.bias + m);
      v *= g.alpha;
      TO* p = C + (int64_t)m * g.ldc + n;
      if (g.flags & OFA_GEMM_ACCUM) v += ld1<TO>(p);
      st1<TO>(p, v);
    }
  }
}

int gemm_simple_launch(const GemmArgs& g, int batch, int dtype, hipStream_t st) {
  dim3 grid(cdiv(g.N, 64), cdiv(g.M, 64), batch), block(256);
  const bool of = (g.flags & OFA_GEMM_OUT_F32) != 0;
  if (dtype == OFA_F32) hipLaunchKernelGGL((gemm_simple_kernel<float, float>), grid, block, 0, st, g);
  else if (dtype == OFA_BF16 && of) hipLaunchKernelGGL((gemm_simple_kernel<bf16_t, float>), grid, block, 0, st, g);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((gemm_simple_kernel<bf16_t, bf16_t>), grid, block, 0, st, g);
  else if (of) hipLaunchKernelGGL((gemm_simple_kernel<f16_t, float>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((gemm_simple_kernel<f16_t, f16_t>), grid, block, 0, st, g);
  return check_launch("gemm_simple");
}

}  // namespace ofa
