// LayerNorm forward/backward for gfx950: one 64-lane wavefront per row, the row lives in registers
// (16-byte vector loads, two-pass mean/variance in fp32), optional fused exact-erf GELU in front of it.
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
        if (GELU) v[i][j] = gelu_f(v[i][j]);
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

// Backward.  A 256-thread block walks rows with a grid stride.  A row is split over WPR (1, 2 or 4) of the block's
// waves -- wide rows (the 4D FFN LayerNorm) spread over four waves so that each lane keeps <= 2 vectors of x, dy, gamma
// AND its running dgamma/dbeta in registers (no LDS accumulators, no spills, full occupancy); the two row statistics
// cross the waves through a 2 x 4-float LDS exchange.  With GELU the activation and its derivative share one erf
// (gelu = x*Phi, gelu' = Phi + x*phi).  Partials go to ws[block_row_slot][cols]; a second kernel folds them
// (deterministic: fixed row->slot assignment and summation order).
template <typename T, int NV, int WPR, bool GELU>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const T* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx,
                                                     float* __restrict__ ws, int64_t rows, int cols, int nslots) {
  constexpr int N = Vec<T>::N;
  constexpr int RPB = 4 / WPR;                       // rows per block iteration
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rib = wib / WPR, part = wib % WPR;       // row-in-block, column part
  const int cpp = cols / WPR;                        // columns per part (launcher guarantees divisibility by N)
  const int c0 = part * cpp;
  const int slot = blockIdx.x * RPB + rib;
  float g[NV][N], dg[NV][N], db[NV][N];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
#pragma unroll
    for (int j = 0; j < N; ++j) dg[i][j] = db[i][j] = 0.f;
    if (c < cpp) load_vec<T>(gamma + c0 + c, g[i]);
  }
  const int64_t stride = (int64_t)gridDim.x * RPB;
  const int64_t niter = (rows + stride - 1) / stride;
  for (int64_t it = 0; it < niter; ++it) {
    const int64_t row = it * stride + slot;
    const bool live = row < rows;
    float xv[NV][N], gp[NV][N], d[NV][N];           // xv: LN input (gelu(h) or x); gp: gelu'(h)
    float s1 = 0.f, s2 = 0.f, mu = 0.f, rs = 0.f;
    if (live) {
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
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * N;
        if (c < cpp) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            if (GELU) {
              const float h = xv[i][j];
              const float cdf = 0.5f * (1.0f + erff(h * 0.70710678118654752440f));
              gp[i][j] = cdf + h * 0.39894228040143267794f * __expf(-0.5f * h * h);
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
    if (WPR > 1) {
      if (lane == 0) { red[0][wib] = s1; red[1][wib] = s2; }
      __syncthreads();
      s1 = 0.f; s2 = 0.f;
#pragma unroll
      for (int p = 0; p < WPR; ++p) { s1 += red[0][rib * WPR + p]; s2 += red[1][rib * WPR + p]; }
      __syncthreads();
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
            if (GELU) t *= gp[i][j];
            o[j] = t;
          }
          store_vec<T>(dxr + c, o);
        }
      }
    }
  }
  float* wg = ws + (int64_t)slot * cols + c0;
  float* wb = ws + (int64_t)nslots * cols + (int64_t)slot * cols + c0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
    if (c < cpp) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        wg[c + j] = dg[i][j];
        wb[c + j] = db[i][j];
      }
    }
  }
}

// fold the per-wave partials; 64 columns x 16 row-lanes per block; writes the parameter dtype, optionally accumulating
// into an existing gradient (gradient arena of the train step).
template <typename T>
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ ws, T* __restrict__ dgamma,
                                                             T* __restrict__ dbeta, int cols, int nwaves, int accumulate) {
  __shared__ float sg[16][64], sb[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float a = 0.f, b = 0.f;
  if (c < cols) {
    for (int w = ry; w < nwaves; w += 16) {
      a += ws[(int64_t)w * cols + c];
      b += ws[(int64_t)nwaves * cols + (int64_t)w * cols + c];
    }
  }
  sg[ry][cx] = a;
  sb[ry][cx] = b;
  __syncthreads();
  if (ry == 0 && c < cols) {
    float ga = 0.f, be = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ga += sg[r][cx]; be += sb[r][cx]; }
    if (accumulate) { ga += ld1<T>(dgamma + c); be += ld1<T>(dbeta + c); }
    st1<T>(dgamma + c, ga);
    st1<T>(dbeta + c, be);
  }
}

constexpr int LN_BWD_WAVES = 1024;  // upper bound on partial rows (workspace sizing)

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

template <typename T, bool GELU>
static int ln_bwd_dispatch(const void* dy, const void* x, const void* g, const float* mean, const float* rstd, void* dx,
                           void* dgamma, void* dbeta, float* ws, int64_t rows, int cols, int accumulate, hipStream_t st) {
  constexpr int N = Vec<T>::N;
  // waves per row: keep <= 2 vectors per lane when the row can be split evenly
  int wpr = 1;
  while (wpr < 4 && cdiv(cols / wpr, 64 * N) > 2 && (cols % (wpr * 2 * N)) == 0) wpr *= 2;
  const int nv = cdiv(cols / wpr, 64 * N);
  const int rpb = 4 / wpr;
  int64_t nblk = (rows + rpb * 4 - 1) / (rpb * 4);      // >= 4 rows per slot
  const int64_t maxblk = LN_BWD_WAVES / rpb;
  nblk = nblk < 1 ? 1 : (nblk > maxblk ? maxblk : nblk);
  const int nslots = (int)nblk * rpb;
  dim3 grid((unsigned)nblk), block(256);
#define LN_LAUNCH(NV, WPR)                                                                                           \
  hipLaunchKernelGGL((ln_bwd_kernel<T, NV, WPR, GELU>), grid, block, 0, st, (const T*)dy, (const T*)x, (const T*)g,  \
                     mean, rstd, (T*)dx, ws, rows, cols, nslots)
#define LN_CASE(WPR)                      \
  do {                                    \
    if (nv <= 1) LN_LAUNCH(1, WPR);       \
    else if (nv <= 2) LN_LAUNCH(2, WPR);  \
    else if (nv <= 4) LN_LAUNCH(4, WPR);  \
    else if (nv <= 8) LN_LAUNCH(8, WPR);  \
    else LN_LAUNCH(16, WPR);              \
  } while (0)
  if (wpr == 1) LN_CASE(1);
  else if (wpr == 2) LN_CASE(2);
  else LN_CASE(4);
#undef LN_CASE
#undef LN_LAUNCH
  int rc = check_launch("layernorm_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL((ln_bwd_reduce_kernel<T>), dim3(cdiv(cols, 64)), dim3(1024), 0, st, (const float*)ws, (T*)dgamma,
                     (T*)dbeta, cols, nslots, accumulate);
  return check_launch("layernorm_bwd_reduce");
}

static int ln_check(int64_t rows, int cols, int dtype, bool bwd) {
  OFA_REQUIRE(rows >= 0 && cols > 0, OFA_ERR_INVALID, "layernorm: bad shape rows=%lld cols=%d", (long long)rows, cols);
  OFA_REQUIRE(dtype == OFA_F32 || dtype == OFA_BF16, OFA_ERR_INVALID, "layernorm: bad dtype %d", dtype);
  const int n = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(cols % n == 0, OFA_ERR_UNSUPPORTED, "layernorm: cols=%d must be a multiple of %d", cols, n);
  const int maxc = 64 * n * (bwd ? 16 : 32);
  OFA_REQUIRE(cols <= maxc, OFA_ERR_UNSUPPORTED, "layernorm: cols=%d exceeds %d", cols, maxc);
  return 0;
}

}  // namespace ofa

using namespace ofa;

extern "C" int ofa_layernorm_bwd_ws_rows(void) { return LN_BWD_WAVES; }

extern "C" int ofa_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                 int64_t rows, int cols, float eps, int dtype, void* stream) {
  if (int rc = ln_check(rows, cols, dtype, false)) return rc;
  OFA_REQUIRE(x && gamma && beta && y, OFA_ERR_INVALID, "layernorm_fwd: null pointer");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == OFA_F32 ? ln_fwd_dispatch<float, false>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st)
                          : ln_fwd_dispatch<bf16_t, false>(x, gamma, beta, y, mean, rstd, rows, cols, eps, st);
}

extern "C" int ofa_gelu_layernorm_fwd(const void* h, const void* gamma, const void* beta, void* y, float* mean,
                                      float* rstd, int64_t rows, int cols, float eps, int dtype, void* stream) {
  if (int rc = ln_check(rows, cols, dtype, false)) return rc;
  OFA_REQUIRE(h && gamma && beta && y, OFA_ERR_INVALID, "gelu_layernorm_fwd: null pointer");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == OFA_F32 ? ln_fwd_dispatch<float, true>(h, gamma, beta, y, mean, rstd, rows, cols, eps, st)
                          : ln_fwd_dispatch<bf16_t, true>(h, gamma, beta, y, mean, rstd, rows, cols, eps, st);
}

extern "C" int ofa_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                                 void* dx, void* dgamma, void* dbeta, float* ws, int64_t rows, int cols, int accumulate,
                                 int dtype, void* stream) {
  if (int rc = ln_check(rows, cols, dtype, true)) return rc;
  OFA_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && ws, OFA_ERR_INVALID,
              "layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  return dtype == OFA_F32
             ? ln_bwd_dispatch<float, false>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, rows, cols, accumulate, st)
             : ln_bwd_dispatch<bf16_t, false>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, rows, cols, accumulate, st);
}

extern "C" int ofa_gelu_layernorm_bwd(const void* dy, const void* h, const void* gamma, const float* mean,
                                      const float* rstd, void* dh, void* dgamma, void* dbeta, float* ws, int64_t rows,
                                      int cols, int accumulate, int dtype, void* stream) {
  if (int rc = ln_check(rows, cols, dtype, true)) return rc;
  OFA_REQUIRE(dy && h && gamma && mean && rstd && dh && dgamma && dbeta && ws, OFA_ERR_INVALID,
              "gelu_layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  return dtype == OFA_F32
             ? ln_bwd_dispatch<float, true>(dy, h, gamma, mean, rstd, dh, dgamma, dbeta, ws, rows, cols, accumulate, st)
             : ln_bwd_dispatch<bf16_t, true>(dy, h, gamma, mean, rstd, dh, dgamma, dbeta, ws, rows, cols, accumulate, st);
}
