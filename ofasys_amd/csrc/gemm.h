// Shared GEMM argument block (exact VALU kernel + MFMA kernel).
#pragma once
#include "common.h"

namespace ofa {
struct GemmArgs {
  const void* A; const void* B; void* C; const void* bias;
  int M, N, K, transA, transB;
  int64_t lda, ldb, ldc, strideA, strideB, strideC;
  float alpha; int flags;
  int batch_inner;                       // batch index z -> (z / batch_inner, z % batch_inner)
  int64_t strideA2, strideB2, strideC2;  // outer strides (elements)
  int b_krows;                           // valid k rows of an m-major B (== K unless the contraction is zero-padded)
  double* colstat;                       // optional: per-column (sum, sum of squares) of the ROUNDED output, one partial row per wave
                                         // block of rows -- [groups][2][N] fp64, groups = ceil(M / colstat_rows) (epilogue_lds; BatchNorm)
  int colstat_rows;                      // rows per partial row (the launcher fills it: 32 * accumulator tiles per wave along M)
  int a_krows;                           // valid k rows of an m-major A (weight gradients over a row count that is not a multiple of 64)
};
__host__ __device__ inline int64_t batch_off(int z, int inner, int64_t s_in, int64_t s_out) {
  return (int64_t)(z / inner) * s_out + (int64_t)(z % inner) * s_in;
}
int gemm_simple_launch(const GemmArgs& g, int batch, int dtype, hipStream_t st);
int gemm_mfma_launch(const GemmArgs& g, int batch, void* ws, int64_t ws_bytes, hipStream_t st, bool f16 = false);
bool gemm_mfma_supported(const GemmArgs& g);
int gemm_mfma_splits(const GemmArgs& g, int batch, int64_t ws_bytes);
// gemm_pp.hip: the ping-pong main loop on the 256 x 256 / 192 x 256 tile (tm = 4 / 3); false: shape or variant not built
bool gemm_pp_launch(int variant, const GemmArgs& g, int batch, int tm, int splits, int ksplit, float* ws, hipStream_t st, bool f16);
}  // namespace ofa
