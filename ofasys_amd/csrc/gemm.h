// Shared GEMM argument block (exact VALU kernel + MFMA kernel).
#pragma once
#include "common.h"

namespace ofa {
struct GemmArgs {
  const void* A; const void* B; void* C; const void* bias;
  int M, N, K, transA, transB;
  int64_t lda, ldb, ldc, strideA, strideB, strideC;
  float alpha; int flags;
};
int gemm_simple_launch(const GemmArgs& g, int batch, int dtype, hipStream_t st);
int gemm_mfma_launch(const GemmArgs& g, int batch, void* ws, int64_t ws_bytes, hipStream_t st);
bool gemm_mfma_supported(const GemmArgs& g);
}  // namespace ofa
