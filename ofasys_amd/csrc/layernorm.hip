// LayerNorm forward/backward for gfx950: one 64-lane wavefront per row, the row lives in registers
// (16-byte vector loads, two-pass mean/variance in fp32), optional fused erf-GELU in front of it
// (libm erff in the fp32 kernels, a 1.5e-7-accurate rational form in the bf16 kernels: gauss_cdf in common.h).
//
// Reference arithmetic: torch.nn.LayerNorm(eps=1e-5, affine)  (module/layer_norm.py:27-32); the GELU+LN pair is
// transformer_layer.py:194-197 / :480-483 (activation_fn(fc1(x)) -> ffn_layernorm), GELU per module/gelu.py:18-19.
// HBM-bound: algorithmic bytes = 2 * rows * cols * sizeof(T) forward (read x, write y).
#include "common.h"

namespace ofa {

template <typename T, int NV, bool GELU>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                     const T* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int cols, float eps) {
  constexpr int N = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * cols;
  float v[NV][N];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
    if (c < cols) {
      load_vec<T>(xr + c, v[i]);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (GELU) {
          float cdf, e;
          gauss_cdf<!__is_same(T, bf16_t)>(v[i][j], cdf, e);
          v[i][j] *= cdf;
        }
        s += v[i][j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) v[i][j] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
    if (c < cols) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float d = v[i][j] - mu;
        q += d * d;
      }
    }
  }
  const float var = wave_sum(q) / (float)cols;
  const float rs = 1.0f / sqrtf(var + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  T* yr = y + row * cols;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
    if (c < cols) {
      float g[N], b[N], o[N];
      load_vec<T>(gamma + c, g);
      load_vec<T>(beta + c, b);
#pragma unroll
      for (int j = 0; j < N; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
      store_vec<T>(yr + c, o);
    }
  }
}

// Backward.  One 1024-thread block (16 waves) per CU walks rows with a grid stride -- 4 waves per SIMD keep enough
// loads in flight to cover HBM latency (the earlier 256-thread / 1024-slot version ran ONE wave per SIMD: 2.9 TB/s).
// A row is split over WPR (1, 2, 4 or 8) waves -- wide rows (the 4D FFN LayerNorm) spread over four waves so that each
// lane keeps <= 2 vectors of x, dy, gamma AND its running dgamma/dbeta(/dbias) in registers; the two row statistics
// cross the waves through a small LDS exchange.  With GELU the activation and its derivative share one erf
// (gelu = x*Phi, gelu' = Phi + x*phi) and the column sums of dh (= the gradient of the bias of the Linear that
// produced h, transformer_layer.py:194) come for free.  At the end the block folds its waves' partials through LDS
// and writes ONE partial row per block to ws[q][block][cols]; a second small kernel folds the <= 256 block rows
// (deterministic: fixed row->wave assignment and summation order).
//
// When a row is split over waves the block synchronises twice per row, so no wave can run ahead and the load latency
// of every iteration would be exposed (14336 x 3072: 28 iterations x ~3 us); the raw 16-byte vectors of the NEXT row are
// therefore fetched before the current row's arithmetic (software prefetch, 8 registers per vector pair).  Rows of 6 x 512
// columns (the base model's 3072-wide FFN LayerNorm) use 12 waves per block, 6 per row: every lane owns exactly one
// vector (with 8 parts only 48 of 64 lanes would) and the 768-thread block has 170 registers per lane for the prefetch.
#ifndef LN_PF_DEPTH
#define LN_PF_DEPTH 2
#endif
constexpr int LN_WPB = 16;          // waves per block (default)
constexpr int LN_BWD_BLOCKS = 256;  // one block per CU; also the workspace row count

template <int I, int E, typename F> __device__ __forceinline__ void static_for_ln(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for_ln<I + 1, E>(f);
  }
}

template <typename T> __device__ __forceinline__ void unpack_vec(const uint4& r, float* out);
template <> __device__ __forceinline__ void unpack_vec<float>(const uint4& r, float* out) {
  out[0] = __uint_as_float(r.x); out[1] = __uint_as_float(r.y); out[2] = __uint_as_float(r.z); out[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack_vec<f16_t>(const uint4& r, float* out) { unpack16<f16_t>(r, out); }
template <> __device__ __forceinline__ void unpack_vec<bf16_t>(const uint4& r, float* out) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(w[i] << 16);
    out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

template <typename T, int NV, int WPR, bool GELU, int WPB = LN_WPB>
__global__ __launch_bounds__(WPB * 64) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const T* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const T* __restrict__ dres,
                                                      T* __restrict__ dx, float* __restrict__ ws, int64_t rows, int cols,
                                                      int want_dbias) {
  constexpr int N = Vec<T>::N;
  constexpr int RPB = WPB / WPR;                     // rows per block iteration
  // (only where the registers are there: at the 128-VGPR cap of the 16-wave block the GELU / 2-vector variants would spill)
  constexpr int PF = WPR > 1 ? (WPB < 16 ? LN_PF_DEPTH : ((!GELU && NV == 1) ? 1 : 0)) : 0;   // rows fetched ahead
  static_assert(WPB % WPR == 0, "waves per block must be a multiple of waves per row");
  __shared__ float red[2][2][WPB];
  __shared__ float fold[WPB][64 * N];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rib = wib / WPR, part = wib % WPR;       // row-in-block, column part
  const int cpp = cols / WPR;                        // columns per part (launcher guarantees divisibility by N)
  const int c0 = part * cpp;
  float g[NV][N], dg[NV][N], db[NV][N], dbi[GELU ? NV : 1][N];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      dg[i][j] = db[i][j] = 0.f;
      if (GELU) dbi[i][j] = 0.f;
      g[i][j] = 0.f;
    }
    if (c < cpp) load_vec<T>(gamma + c0 + c, g[i]);
  }
  const int64_t stride = (int64_t)gridDim.x * RPB;
  const int64_t niter = (rows + stride - 1) / stride;
  constexpr int RING = PF > 0 ? PF : 1;
  uint4 rx[RING][NV], rd[RING][NV];                  // raw vectors of the rows in flight
  float nmu[RING], nrs[RING];
  auto fetch = [&](int64_t it, auto bc) {
    constexpr int b = decltype(bc)::value;
    const int64_t row = it * stride + (int64_t)blockIdx.x * RPB + rib;
    if (row < rows) {
      nmu[b] = mean[row];
      nrs[b] = rstd[row];
      const T* xr = x + row * cols + c0;
      const T* dyr = dy + row * cols + c0;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * N;
        if (c < cpp) {
          rx[b][i] = *reinterpret_cast<const uint4*>(xr + c);
          rd[b][i] = *reinterpret_cast<const uint4*>(dyr + c);
        }
      }
    }
  };
  if constexpr (PF > 0) static_for_ln<0, PF>([&](auto bc) { if (decltype(bc)::value < niter) fetch(decltype(bc)::value, bc); });
  // the loop is unrolled RING times so that the ring slot is a compile-time index (registers, no copies)
  for (int64_t it0 = 0; it0 < niter; it0 += RING)
  static_for_ln<0, RING>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const int64_t it = it0 + b;
    if (it >= niter) return;                         // block-uniform
    const int64_t row = it * stride + (int64_t)blockIdx.x * RPB + rib;
    const bool live = row < rows;
    float xv[NV][N], gp[GELU ? NV : 1][N], d[NV][N];  // xv: LN input (gelu(h) or x); gp: gelu'(h)
    float s1 = 0.f, s2 = 0.f, mu = 0.f, rs = 0.f;
    if constexpr (PF > 0) {
      if (live) {
        mu = nmu[b];
        rs = nrs[b];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int c = (i * 64 + lane) * N;
          if (c < cpp) {
            unpack_vec<T>(rx[b][i], xv[i]);
            unpack_vec<T>(rd[b][i], d[i]);
          }
        }
      }
      if (it + PF < niter) fetch(it + PF, bc);
    } else if (live) {
      mu = mean[row];
      rs = rstd[row];
      const T* xr = x + row * cols + c0;
      const T* dyr = dy + row * cols + c0;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * N;
        if (c < cpp) {
          load_vec<T>(xr + c, xv[i]);
          load_vec<T>(dyr + c, d[i]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * N;
        if (c < cpp) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            if (GELU) {
              const float h = xv[i][j];
              float cdf, e;
              gauss_cdf<!__is_same(T, bf16_t)>(h, cdf, e);
              gp[i][j] = cdf + h * 0.39894228040143267794f * e;
              xv[i][j] = h * cdf;
            }
            const float xh = (xv[i][j] - mu) * rs;
            const float gy = d[i][j] * g[i][j];
            s1 += gy;
            s2 += gy * xh;
            dg[i][j] += d[i][j] * xh;
            db[i][j] += d[i][j];
            xv[i][j] = xh;
            d[i][j] = gy;
          }
        }
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (WPR > 1) {                                   // one barrier per row: the exchange buffer alternates
      float (*rb)[WPB] = red[it & 1];
      if (lane == 0) { rb[0][wib] = s1; rb[1][wib] = s2; }
      __syncthreads();
      s1 = 0.f; s2 = 0.f;
#pragma unroll
      for (int p = 0; p < WPR; ++p) { s1 += rb[0][rib * WPR + p]; s2 += rb[1][rib * WPR + p]; }
    }
    s1 /= (float)cols;
    s2 /= (float)cols;
    if (live) {
      T* dxr = dx + row * cols + c0;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * N;
        if (c < cpp) {
          float o[N];
#pragma unroll
          for (int j = 0; j < N; ++j) {
            float t = rs * (d[i][j] - s1 - xv[i][j] * s2);
            if (GELU) {
              t *= gp[i][j];
              dbi[i][j] += t;
            }
            o[j] = t;
          }
          if constexpr (!GELU) {                          // (compiled out of the GELU variant: it sits at the 128-VGPR cap)
            if (dres) {                                   // gradient arriving through the residual branch
              float rsd[N];
              load_vec<T>(dres + row * cols + c0 + c, rsd);
#pragma unroll
              for (int j = 0; j < N; ++j) o[j] += rsd[j];
            }
          }
          store_vec<T>(dxr + c, o);
        }
      }
    }
  });
  // fold the block's waves: quantity q of vector i goes through fold[wave][lane*N + j]; the rib == 0 wave of each
  // column part sums its RPB peers in fixed order and writes the block's partial row
  const int nq = (GELU && want_dbias) ? 3 : 2;
  for (int q = 0; q < nq; ++q) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j)
        fold[wib][lane * N + j] = q == 0 ? dg[i][j] : (q == 1 ? db[i][j] : dbi[GELU ? i : 0][j]);
      __syncthreads();
      const int c = (i * 64 + lane) * N;
      if (rib == 0 && c < cpp) {
        float* w = ws + ((int64_t)q * gridDim.x + blockIdx.x) * cols + c0 + c;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          float a = 0.f;
#pragma unroll
          for (int r = 0; r < RPB; ++r) a += fold[r * WPR + part][lane * N + j];
          w[j] = a;
        }
      }
      __syncthreads();
    }
  }
}

// fold the per-block partial rows: 32 columns x 8 row-lanes per block, grid.y = quantity (dgamma, dbeta, dbias); writes
// the parameter dtype, optionally accumulating into an existing gradient (gradient arena of the train step).
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ ws, T* __restrict__ o0,
                                                            T* __restrict__ o1, T* __restrict__ o2, int cols, int nslots,
                                                            int accumulate) {
  __shared__ float sg[8][32];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  const int q = blockIdx.y;
  T* out = q == 0 ? o0 : (q == 1 ? o1 : o2);
  const float* w = ws + (int64_t)q * nslots * cols;
  float a = 0.f;
  if (c < cols)
    for (int s = ry; s < nslots; s += 8) a += w[(int64_t)s * cols + c];
  sg[ry][cx] = a;
  __syncthreads();
  if (ry == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += sg[r][cx];
    if (accumulate) t += ld1<T>(out + c);
    st1<T>(out + c, t);
  }
}

template <typename T, bool GELU>
static int ln_fwd_dispatch(const void* x, const void* g, const void* b, void* y, float* mean, float* rstd, int64_t rows,
                           int cols, float eps, hipStream_t st) {
  constexpr int N = Vec<T>::N;
  const int nv = cdiv(cols, 64 * N);
  dim3 grid(cdiv(rows, 4)), block(256);
#define LN_CASE(NV)                                                                                                \
  hipLaunchKernelGGL((ln_fwd_kernel<T, NV, GELU>), grid, block, 0, st, (const T*)x, (const T*)g, (const T*)b, (T*)y, \
                     mean, rstd, rows, cols, eps)
  if (nv <= 1) LN_CASE(1);
  else if (nv <= 2) LN_CASE(2);
  else if (nv <= 4) LN_CASE(4);
  else if (nv <= 8) LN_CASE(8);
  else if (nv <= 16) LN_CASE(16);
  else LN_CASE(32);
#undef LN_CASE
  return check_launch("layernorm_fwd");
}

static int ln_bwd_wpr(int cols, int n, bool gelu) {
  if (gelu && cols == 6 * 64 * n) return 6;          // 12-wave blocks, one full vector per lane
  int wpr = 1;
  while (wpr < 8 && cdiv(cols / wpr, 64 * n) > (gelu ? 1 : 2) && (cols % (wpr * 2 * n)) == 0) wpr *= 2;
  return wpr;
}

template <typename T, bool GELU>
static int ln_bwd_dispatch(const void* dy, const void* x, const void* g, const float* mean, const float* rstd,
                           const void* dres, void* dx, void* dgamma, void* dbeta, void* dbias, float* ws, int64_t rows,
                           int cols, int accumulate, hipStream_t st) {
  constexpr int N = Vec<T>::N;
  // waves per row: keep <= 2 vectors per lane (1 with GELU, whose extra gelu'/dbias registers would otherwise spill at
  // the 128-VGPR cap of a 1024-thread block: 169 us instead of ~70 for the 14336 x 3072 FFN LayerNorm) when the row
  // can be split evenly
  const int wpr = ln_bwd_wpr(cols, N, GELU);
  const int nv = cdiv(cols / wpr, 64 * N);
  const int wpb = wpr == 6 ? 12 : LN_WPB;
  const int rpb = wpb / wpr;
  int64_t nblk = (rows + rpb - 1) / rpb;
  nblk = nblk < 1 ? 1 : (nblk > LN_BWD_BLOCKS ? LN_BWD_BLOCKS : nblk);
  const int want_dbias = dbias != nullptr;
  dim3 grid((unsigned)nblk), block(64 * wpb);
  if (wpr == 6) {
    if constexpr (GELU)
      hipLaunchKernelGGL((ln_bwd_kernel<T, 1, 6, true, 12>), grid, block, 0, st, (const T*)dy, (const T*)x, (const T*)g,
                         mean, rstd, (const T*)dres, (T*)dx, ws, rows, cols, want_dbias);
  } else {
#define LN_LAUNCH(NV, WPR)                                                                                           \
  hipLaunchKernelGGL((ln_bwd_kernel<T, NV, WPR, GELU>), grid, block, 0, st, (const T*)dy, (const T*)x, (const T*)g,  \
                     mean, rstd, (const T*)dres, (T*)dx, ws, rows, cols, want_dbias)
#define LN_CASE(WPR)                      \
  do {                                    \
    if (nv <= 1) LN_LAUNCH(1, WPR);       \
    else if (nv <= 2) LN_LAUNCH(2, WPR);  \
    else if (nv <= 4) LN_LAUNCH(4, WPR);  \
    else LN_LAUNCH(8, WPR);               \
  } while (0)
  if (wpr == 1) LN_CASE(1);
  else if (wpr == 2) LN_CASE(2);
  else if (wpr == 4) LN_CASE(4);
  else LN_CASE(8);
#undef LN_CASE
#undef LN_LAUNCH
  }
  int rc = check_launch("layernorm_bwd");
  if (rc || accumulate == OFA_DEFER_FOLD) return rc;          // deferred: the caller folds ws (ofa_fold_batched)
  hipLaunchKernelGGL((ln_bwd_reduce_kernel<T>), dim3(cdiv(cols, 32), want_dbias ? 3 : 2), dim3(256), 0, st,
                     (const float*)ws, (T*)dgamma, (T*)dbeta, (T*)dbias, cols, (int)nblk, accumulate);
  return check_launch("layernorm_bwd_reduce");
}

static int ln_check(int64_t rows, int cols, int dtype, bool bwd) {
  OFA_REQUIRE(rows >= 0 && cols > 0, OFA_ERR_INVALID, "layernorm: bad shape rows=%lld cols=%d", (long long)rows, cols);
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "layernorm: bad dtype %d", dtype);
  const int n = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(cols % n == 0, OFA_ERR_UNSUPPORTED, "layernorm: cols=%d must be a multiple of %d", cols, n);
  if (bwd) {                                      // the same row split the dispatcher makes; <= 8 vectors per lane
    int wpr = 1;
    while (wpr < 8 && cdiv(cols / wpr, 64 * n) > 2 && (cols % (wpr * 2 * n)) == 0) wpr *= 2;
    OFA_REQUIRE(cdiv(cols / wpr, 64 * n) <= 8, OFA_ERR_UNSUPPORTED, "layernorm_bwd: cols=%d too wide", cols);
  } else {
    OFA_REQUIRE(cols <= 64 * n * 32, OFA_ERR_UNSUPPORTED, "layernorm: cols=%d exceeds %d", cols, 64 * n * 32);
  }
  return 0;
}

}  // namespace ofa

using namespace ofa;

extern "C" int ofa_layernorm_bwd_ws_rows(void) { return 3 * LN_BWD_BLOCKS; }

extern "C" int ofa_layernorm_bwd_slots(int64_t rows, int cols, int dtype, int gelu) {
  const int wpr = ln_bwd_wpr(cols, dtype == OFA_F32 ? 4 : 8, gelu != 0);
  const int rpb = (wpr == 6 ? 12 : LN_WPB) / wpr;
  int64_t nblk = (rows + rpb - 1) / rpb;
  return (int)(nblk < 1 ? 1 : (nblk > LN_BWD_BLOCKS ? LN_BWD_BLOCKS : nblk));
}

extern "C" int ofa_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                 int64_t rows, int cols, float eps, int dtype, void* stream) {
  if (int rc = ln_check(rows, cols, dtype, false)) return rc;
  OFA_REQUIRE(x && gamma && beta && y, OFA_ERR_INVALID, "layernorm_fwd: null pointer");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == OFA_F32    ? ln_fwd_dispatch<float, false>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st)
         : dtype == OFA_BF16 ? ln_fwd_dispatch<bf16_t, false>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st)
                             : ln_fwd_dispatch<f16_t, false>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st);
}

extern "C" int ofa_gelu_layernorm_fwd(const void* h, const void* gamma, const void* beta, void* y, float* mean,
                                      float* rstd, int64_t rows, int cols, float eps, int dtype, void* stream) {
  if (int rc = ln_check(rows, cols, dtype, false)) return rc;
  OFA_REQUIRE(h && gamma && beta && y, OFA_ERR_INVALID, "gelu_layernorm_fwd: null pointer");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == OFA_F32    ? ln_fwd_dispatch<float, true>(h, gamma, beta, y, mean, rstd, rows, cols, eps, st)
         : dtype == OFA_BF16 ? ln_fwd_dispatch<bf16_t, true>(h, gamma, beta, y, mean, rstd, rows, cols, eps, st)
                             : ln_fwd_dispatch<f16_t, true>(h, gamma, beta, y, mean, rstd, rows, cols, eps, st);
}

extern "C" int ofa_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, void* dgamma, void* dbeta, float* ws, int64_t rows, int cols,
                                 int accumulate, int dtype, void* stream) {
  if (int rc = ln_check(rows, cols, dtype, true)) return rc;
  OFA_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && ws, OFA_ERR_INVALID,
              "layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  return dtype == OFA_F32
             ? ln_bwd_dispatch<float, false>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, nullptr, ws, rows, cols, accumulate, st)
         : dtype == OFA_BF16
             ? ln_bwd_dispatch<bf16_t, false>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, nullptr, ws, rows, cols, accumulate, st)
             : ln_bwd_dispatch<f16_t, false>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, nullptr, ws, rows, cols, accumulate, st);
}

extern "C" int ofa_gelu_layernorm_bwd(const void* dy, const void* h, const void* gamma, const float* mean,
                                      const float* rstd, void* dh, void* dgamma, void* dbeta, void* dbias, float* ws,
                                      int64_t rows, int cols, int accumulate, int dtype, void* stream) {
  if (int rc = ln_check(rows, cols, dtype, true)) return rc;
  OFA_REQUIRE(dy && h && gamma && mean && rstd && dh && dgamma && dbeta && ws, OFA_ERR_INVALID,
              "gelu_layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  return dtype == OFA_F32
             ? ln_bwd_dispatch<float, true>(dy, h, gamma, mean, rstd, nullptr, dh, dgamma, dbeta, dbias, ws, rows, cols, accumulate, st)
         : dtype == OFA_BF16
             ? ln_bwd_dispatch<bf16_t, true>(dy, h, gamma, mean, rstd, nullptr, dh, dgamma, dbeta, dbias, ws, rows, cols, accumulate, st)
             : ln_bwd_dispatch<f16_t, true>(dy, h, gamma, mean, rstd, nullptr, dh, dgamma, dbeta, dbias, ws, rows, cols, accumulate, st);
}
