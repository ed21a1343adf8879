// Residual join of a pre-LN transformer sub-block, one kernel each way:
//     y = residual + dropout_p( LN_a(x) )            LN_a optional (attn_ln / self_attn_ln / cross_attn_ln, scale_attn)
//     z = LN_b(y)                                    LN_b optional (the NEXT block's pre-LayerNorm)
// (reference: module/transformer_layer.py:167-208 / :438-494 -- `x = attn_ln(x); x = dropout(x); x = residual + x;
//  residual = x; x = final_layer_norm(x)` and the same chain at the other sub-block boundaries.)  Op by op that is three
// kernels forward (LayerNorm, dropout+add, LayerNorm) and three backward plus their small reduce launches, every one of
// them a full pass over the [rows, C] activations; fused it is one pass each way.  Rounding points are kept where the
// unfused kernels put them (LN_a output, y, the gradient of y, the dropout gradient are rounded to the storage dtype),
// so the fused and the unfused compositions agree bit for bit.  HBM-bound: 4 (forward) / 6 (backward) row passes.
#include "common.h"

namespace ofa {

template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float rnd<f16_t>(float v) { return (float)(f16_t)v; }

// keep / drop decisions of the N (4 or 8) elements starting at element index e0 (e0 % N == 0): dropout_kernel's rule
// (element e <- Philox counter offset + e/8, halfword e%8)
template <int N>
__device__ __forceinline__ void keep_mask(const Philox& rng, uint64_t offset, int64_t e0, float p, bool (&keep)[N]) {
  const uint4 r = rng(offset + (uint64_t)(e0 >> 3));
  const uint32_t thresh = philox_thresh(p);
  if (N == 8) {
#pragma unroll
    for (int j = 0; j < N; ++j) keep[j] = philox_keep(r, j, thresh);
  } else {
    const bool hi = (e0 & 4) != 0;                  // the upper four halfwords
    const uint32_t w0 = hi ? r.z : r.x, w1 = hi ? r.w : r.y;
    keep[0] = (w0 & 0xffffu) >= thresh;
    keep[1] = (w0 >> 16) >= thresh;
    keep[2] = (w1 & 0xffffu) >= thresh;
    keep[3] = (w1 >> 16) >= thresh;
  }
}

template <typename T> __device__ __forceinline__ void jn_unpack(const uint4& r, float* out);
template <> __device__ __forceinline__ void jn_unpack<float>(const uint4& r, float* out) {
  out[0] = __uint_as_float(r.x); out[1] = __uint_as_float(r.y); out[2] = __uint_as_float(r.z); out[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void jn_unpack<f16_t>(const uint4& r, float* out) { unpack16<f16_t>(r, out); }
template <> __device__ __forceinline__ void jn_unpack<bf16_t>(const uint4& r, float* out) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(w[i] << 16);
    out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

struct JoinRng { float p; uint64_t seed, offset; const int64_t* base; };

// forward: one wave per row, the row in registers
template <typename T, int NV>
__global__ __launch_bounds__(256) void join_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                       const T* __restrict__ ga, const T* __restrict__ ba,
                                                       const T* __restrict__ gb, const T* __restrict__ bb, T* __restrict__ y,
                                                       T* __restrict__ z, float* __restrict__ stats, int64_t rows, int cols,
                                                       float eps, JoinRng rg) {
  constexpr int N = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NV][N];
  uint4 rraw[NV];                                       // the residual row is fetched together with x: one HBM latency, not two
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
#pragma unroll
    for (int j = 0; j < N; ++j) v[i][j] = 0.f;
    rraw[i] = make_uint4(0, 0, 0, 0);
    if (c < cols) {
      load_vec<T>(x + row * cols + c, v[i]);
      if (res) rraw[i] = *reinterpret_cast<const uint4*>(res + row * cols + c);     // (no residual: y = dropout(LN_a(x)), the adaptor post-hook)
    }
  }
  if (ga) {                                             // LN_a, two-pass statistics in registers
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) s += v[i][j];
    const float mu = wave_sum(s) / (float)cols;            // (divisions, not reciprocal multiplies: bit-equal to ln_fwd_kernel)
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((i * 64 + lane) * N < cols)
#pragma unroll
        for (int j = 0; j < N; ++j) q += (v[i][j] - mu) * (v[i][j] - mu);
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) { stats[row] = mu; stats[rows + row] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * N;
      if (c < cols) {
        float g[N], b[N];
        load_vec<T>(ga + c, g);
        load_vec<T>(ba + c, b);
#pragma unroll
        for (int j = 0; j < N; ++j) v[i][j] = rnd<T>((v[i][j] - mu) * rs * g[j] + b[j]);
      }
    }
  }
  const uint64_t off = rg.offset + (rg.base ? (uint64_t)rg.base[0] : 0);
  const Philox rng(rg.seed);
  const float scale = rg.p > 0.f ? 1.0f / (1.0f - rg.p) : 1.0f;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
    if (c < cols) {
      float r[N];
      jn_unpack<T>(rraw[i], r);
      if (rg.p > 0.f) {
        bool keep[N];
        keep_mask<N>(rng, off, row * cols + c, rg.p, keep);
#pragma unroll
        for (int j = 0; j < N; ++j) v[i][j] = keep[j] ? v[i][j] * scale : 0.f;
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        v[i][j] = rnd<T>(v[i][j] + r[j]);
        s += v[i][j];
      }
      store_vec<T>(y + row * cols + c, v[i]);
    }
  }
  if (!gb) return;
  const float mu = wave_sum(s) / (float)cols;            // LN_b of the (rounded) y
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((i * 64 + lane) * N < cols)
#pragma unroll
      for (int j = 0; j < N; ++j) q += (v[i][j] - mu) * (v[i][j] - mu);
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
  if (lane == 0) { stats[2 * rows + row] = mu; stats[3 * rows + row] = rs; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * N;
    if (c < cols) {
      float g[N], b[N], o[N];
      load_vec<T>(gb + c, g);
      load_vec<T>(bb + c, b);
#pragma unroll
      for (int j = 0; j < N; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
      store_vec<T>(z + row * cols + c, o);
    }
  }
}

// backward: 12-wave blocks (768 threads: 170 registers per lane, room for the next-row prefetch without spilling; 16
// waves for rows split 8 ways), a row split over WPR waves (one vector per lane), column partials folded through LDS
// into one row per block of ws[q][block][cols], q = dgamma_a, dbeta_a, dgamma_b, dbeta_b.
constexpr int JOIN_BLOCKS = 256;
constexpr int join_wpb(int wpr) { return wpr == 8 ? 16 : 12; }

template <typename T, int WPR>
__global__ __launch_bounds__(join_wpb(WPR) * 64) void join_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dz,
                                                        const T* __restrict__ x, const T* __restrict__ y,
                                                        const T* __restrict__ ga, const T* __restrict__ gb,
                                                        const float* __restrict__ stats, T* __restrict__ dres,
                                                        T* __restrict__ dx, float* __restrict__ ws, int64_t rows, int cols,
                                                        JoinRng rg, int want_xsum) {
  constexpr int N = Vec<T>::N;
  constexpr int JOIN_WPB = join_wpb(WPR);
  constexpr int RPB = JOIN_WPB / WPR;
  __shared__ float red[2][2][2][JOIN_WPB];          // [LN_b / LN_a exchange][row parity][s1, s2][wave]
  __shared__ float fold[JOIN_WPB][64 * N];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rib = wib / WPR, part = wib % WPR;
  const int cpp = cols / WPR, c0 = part * cpp;
  const int c = lane * N;
  const bool act = c < cpp;
  float gA[N], gB[N], acc[5][N];                          // dgamma_a, dbeta_a, dgamma_b, dbeta_b, column sums of dx
#pragma unroll
  for (int j = 0; j < N; ++j) {
    gA[j] = gB[j] = 0.f;
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q][j] = 0.f;
  }
  if (act && ga) load_vec<T>(ga + c0 + c, gA);
  if (act && gb) load_vec<T>(gb + c0 + c, gB);
  const uint64_t off = rg.offset + (rg.base ? (uint64_t)rg.base[0] : 0);
  const Philox rng(rg.seed);
  const float scale = rg.p > 0.f ? 1.0f / (1.0f - rg.p) : 1.0f;
  // sums over the whole row (all WPR parts): ONE barrier per exchange -- the buffer alternates with the row parity and
  // between the two exchanges of a row, so a fast wave's next write never lands on a buffer a slow wave still reads
  auto row_sums = [&](float& s1, float& s2, int which, int64_t it) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (WPR > 1) {
      float (*rb)[JOIN_WPB] = red[which][it & 1];
      if (lane == 0) { rb[0][wib] = s1; rb[1][wib] = s2; }
      __syncthreads();
      s1 = 0.f; s2 = 0.f;
#pragma unroll
      for (int p2 = 0; p2 < WPR; ++p2) { s1 += rb[0][rib * WPR + p2]; s2 += rb[1][rib * WPR + p2]; }
    }
  };
  const int64_t stride = (int64_t)gridDim.x * RPB;
  const int64_t niter = (rows + stride - 1) / stride;
  // The block synchronises on every row, so no wave runs ahead: the NEXT row's four vectors and statistics are fetched
  // (raw, 16 registers) before this row's arithmetic, otherwise every iteration pays the full HBM latency twice.
  uint4 r_dy, r_dz, r_y, r_x;
  float pst[4] = {0.f, 0.f, 0.f, 0.f};
  r_dy = r_dz = r_y = r_x = make_uint4(0, 0, 0, 0);
  auto fetch = [&](int64_t it) {
    const int64_t row = it * stride + (int64_t)blockIdx.x * RPB + rib;
    if (row < rows && act) {
      const int64_t e0 = row * cols + c0 + c;
      if (dy) r_dy = *reinterpret_cast<const uint4*>(dy + e0);
      if (gb) {
        r_dz = *reinterpret_cast<const uint4*>(dz + e0);
        r_y = *reinterpret_cast<const uint4*>(y + e0);
        pst[2] = stats[2 * rows + row];
        pst[3] = stats[3 * rows + row];
      }
      if (ga) {
        r_x = *reinterpret_cast<const uint4*>(x + e0);
        pst[0] = stats[row];
        pst[1] = stats[rows + row];
      }
    }
  };
  fetch(0);
  for (int64_t it = 0; it < niter; ++it) {
    const int64_t row = it * stride + (int64_t)blockIdx.x * RPB + rib;
    const bool live = row < rows && act;
    const int64_t e0 = row * cols + c0 + c;
    float g[N], dzv[N], yv[N];                            // g: running gradient of this lane's N columns
    jn_unpack<T>(r_dy, g);                                // (zeros when dy is absent)
    jn_unpack<T>(r_dz, dzv);
    jn_unpack<T>(r_y, yv);
    const uint4 xraw = r_x;
    const float mu_a = pst[0], rs_a = pst[1], mu_b = pst[2], rs_b = pst[3];
    if (it + 1 < niter) fetch(it + 1);
    if (!live) {
#pragma unroll
      for (int j = 0; j < N; ++j) g[j] = 0.f;
    }
    if (gb) {                                             // LN_b backward (block-uniform branch)
      float yh[N], s1 = 0.f, s2 = 0.f, rs = 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) yh[j] = 0.f;
      if (live) {
        const float mu = mu_b;
        rs = rs_b;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          yh[j] = (yv[j] - mu) * rs;
          const float gy = dzv[j] * gB[j];
          s1 += gy;
          s2 += gy * yh[j];
          acc[2][j] += dzv[j] * yh[j];
          acc[3][j] += dzv[j];
        }
      }
      row_sums(s1, s2, 0, it);
      s1 /= (float)cols;
      s2 /= (float)cols;
      if (live) {
#pragma unroll
        for (int j = 0; j < N; ++j) g[j] = rnd<T>(rs * (dzv[j] * gB[j] - s1 - yh[j] * s2) + g[j]);   // LN_b input gradient + dy, rounded once
      }
    }
    if (live && dres) store_vec<T>(dres + e0, g);         // gradient of the residual input == gradient of y
    if (rg.p > 0.f && live) {                             // dropout backward on the rounded gradient
      bool keep[N];
      keep_mask<N>(rng, off, e0, rg.p, keep);
#pragma unroll
      for (int j = 0; j < N; ++j) g[j] = rnd<T>(keep[j] ? g[j] * scale : 0.f);
    }
    if (ga) {                                             // LN_a backward
      float xh[N], s1 = 0.f, s2 = 0.f, rs = 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) xh[j] = 0.f;
      if (live) {
        const float mu = mu_a;
        rs = rs_a;
        float xv[N];
        jn_unpack<T>(xraw, xv);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          xh[j] = (xv[j] - mu) * rs;
          const float gy = g[j] * gA[j];
          s1 += gy;
          s2 += gy * xh[j];
          acc[0][j] += g[j] * xh[j];
          acc[1][j] += g[j];
        }
      }
      row_sums(s1, s2, 1, it);
      s1 /= (float)cols;
      s2 /= (float)cols;
      if (live) {
#pragma unroll
        for (int j = 0; j < N; ++j) g[j] = rs * (g[j] * gA[j] - s1 - xh[j] * s2);
      }
    }
    if (live) {
      store_vec<T>(dx + e0, g);
      if (want_xsum) {                                    // gradient of the bias of the Linear that produced x
#pragma unroll
        for (int j = 0; j < N; ++j) acc[4][j] += g[j];
      }
    }
  }
  // fold the block's waves (same scheme as ln_bwd_kernel)
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    if ((q < 2 && !ga) || (q >= 2 && q < 4 && !gb) || (q == 4 && !want_xsum)) continue;      // (block-uniform)
#pragma unroll
    for (int j = 0; j < N; ++j) fold[wib][lane * N + j] = acc[q][j];
    __syncthreads();
    if (rib == 0 && act) {
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

static int join_wpr(int cols, int n) {
  int wpr = 1;
  while (wpr < 8 && cols / wpr > 64 * n) wpr *= 2;
  return wpr;
}
static int join_check(int64_t rows, int cols, int dtype, const char* what) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "%s: bad dtype %d", what, dtype);
  const int n = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(rows >= 0 && cols > 0 && cols % n == 0 && cols <= 64 * n * 8, OFA_ERR_UNSUPPORTED,
              "%s: cols=%d must be a multiple of %d and <= %d", what, cols, n, 64 * n * 8);
  const int wpr = join_wpr(cols, n);
  OFA_REQUIRE(cols % (wpr * n) == 0, OFA_ERR_UNSUPPORTED, "%s: cols=%d does not split over %d waves", what, cols, wpr);
  return 0;
}

}  // namespace ofa
using namespace ofa;

extern "C" int ofa_join_fwd(const void* x, const void* residual, const void* gamma_a, const void* beta_a, const void* gamma_b,
                            const void* beta_b, void* y, void* z, float* stats, int64_t rows, int cols, float eps, float p,
                            uint64_t seed, uint64_t offset, const int64_t* offset_base, int dtype, void* stream) {
  if (int rc = join_check(rows, cols, dtype, "join_fwd")) return rc;
  OFA_REQUIRE(x && y && stats && (!gamma_a == !beta_a) && (!gamma_b == !beta_b) && (!gamma_b || z) && p >= 0.f && p < 1.f,
              OFA_ERR_INVALID, "join_fwd: bad argument");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const JoinRng rg{p, seed, offset, offset_base};
  const int n = dtype == OFA_F32 ? 4 : 8;
  const int nv = cdiv(cols, 64 * n);
  dim3 grid(cdiv(rows, 4)), block(256);
#define JOIN_FWD(T, NV)                                                                                                  \
  hipLaunchKernelGGL((join_fwd_kernel<T, NV>), grid, block, 0, st, (const T*)x, (const T*)residual, (const T*)gamma_a,   \
                     (const T*)beta_a, (const T*)gamma_b, (const T*)beta_b, (T*)y, (T*)z, stats, rows, cols, eps, rg)
#define JOIN_FWD_T(T)                 \
  do {                                \
    if (nv <= 1) JOIN_FWD(T, 1);      \
    else if (nv <= 2) JOIN_FWD(T, 2); \
    else if (nv <= 4) JOIN_FWD(T, 4); \
    else JOIN_FWD(T, 8);              \
  } while (0)
  if (dtype == OFA_F32) JOIN_FWD_T(float);
  else if (dtype == OFA_BF16) JOIN_FWD_T(bf16_t);
  else JOIN_FWD_T(f16_t);
#undef JOIN_FWD_T
#undef JOIN_FWD
  return check_launch("join_fwd");
}

extern "C" int ofa_join_bwd_slots(int64_t rows, int cols, int dtype) {
  const int wpr = join_wpr(cols, dtype == OFA_F32 ? 4 : 8);
  const int rpb = join_wpb(wpr) / wpr;
  int64_t nblk = (rows + rpb - 1) / rpb;
  return (int)(nblk < 1 ? 1 : (nblk > JOIN_BLOCKS ? JOIN_BLOCKS : nblk));
}

// dy / dz: gradients of y / z (either may be NULL = zero; dz must be NULL iff gamma_b is); dres: gradient of the residual
// input; dx: gradient of x; ws: fp32 [5][ofa_join_bwd_slots][cols] partial rows of dgamma_a, dbeta_a, dgamma_b, dbeta_b and,
// with want_dx_colsum, the column sums of dx (= the bias gradient of the Linear that produced x, so that Linear needs no
// separate column-sum pass); fold with ofa_fold_batched; quantities that do not apply are not written.
extern "C" int ofa_join_bwd(const void* dy, const void* dz, const void* x, const void* y, const void* gamma_a, const void* gamma_b,
                            const float* stats, void* dres, void* dx, float* ws, int64_t rows, int cols, float p, uint64_t seed,
                            uint64_t offset, const int64_t* offset_base, int want_dx_colsum, int dtype, void* stream) {
  if (int rc = join_check(rows, cols, dtype, "join_bwd")) return rc;
  OFA_REQUIRE(stats && dx && ws && (!gamma_a || x) && (!gamma_b || (y && dz)) && (dy || dz), OFA_ERR_INVALID,
              "join_bwd: bad argument");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const JoinRng rg{p, seed, offset, offset_base};
  const int wpr = join_wpr(cols, dtype == OFA_F32 ? 4 : 8);
  dim3 grid(ofa_join_bwd_slots(rows, cols, dtype)), block(64 * join_wpb(wpr));
#define JOIN_BWD(T, WPR)                                                                                              \
  hipLaunchKernelGGL((join_bwd_kernel<T, WPR>), grid, block, 0, st, (const T*)dy, (const T*)(gamma_b ? dz : nullptr),  \
                     (const T*)x, (const T*)y, (const T*)gamma_a, (const T*)gamma_b, stats, (T*)dres, (T*)dx, ws, rows, cols, rg, \
                     want_dx_colsum)
#define JOIN_BWD_T(T)                  \
  do {                                 \
    if (wpr == 1) JOIN_BWD(T, 1);      \
    else if (wpr == 2) JOIN_BWD(T, 2); \
    else if (wpr == 4) JOIN_BWD(T, 4); \
    else JOIN_BWD(T, 8);               \
  } while (0)
  if (dtype == OFA_F32) JOIN_BWD_T(float);
  else if (dtype == OFA_BF16) JOIN_BWD_T(bf16_t);
  else JOIN_BWD_T(f16_t);
#undef JOIN_BWD_T
#undef JOIN_BWD
  return check_launch("join_bwd");
}
