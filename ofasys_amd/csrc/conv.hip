// Convolution-stack pieces of the image_resnet / video / audio adaptors (reference: module/resnet.py:22-261,
// module/subsample.py:11-63): activations live as NHWC rows [B*H*W, C], so
//   * a 1x1 stride-1 convolution IS a GEMM on the rows; every other convolution is an im2col gather (taps ordered
//     (kh, kw, c): one contiguous C-vector per tap) + the MFMA GEMM (gemm_mfma.hip), its input gradient the matching
//     gather-formulated col2im (deterministic, no atomics);
//   * BatchNorm2d is a per-column statistic over the rows: one pass of column sums (x, x^2), a finalize kernel (mean,
//     rstd, running statistics with torch's momentum / unbiased-variance rule), and one fused normalise (+ residual)
//     (+ ReLU) pass; backward mirrors it (column sums of g and g*xhat, then dx [+ d_residual]);
//   * MaxPool2d(3, 2, 1) keeps the arg-max tap in a byte per output element.
// All HBM-bound; algorithmic bytes: im2col (1 + kh*kw) * rows*C*sizeof, BN forward 3 passes, backward 5.
#include "common.h"

namespace ofa {

static inline int grid_1d(int64_t work) {
  int64_t g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// col[(b, oh, ow)][(kh*KW + kw)*C + c] = x[b, oh*s - p + kh, ow*s - p + kw, c]  (0 outside), columns K..Kpad-1 zero.
// NCHW: x is [B, C, H, W] (the image itself, first convolution); else [B, H, W, C].
template <typename T, bool NCHW>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int Kpad) {
  constexpr int N = Vec<T>::N;
  const int K = KH * KW * C;
  if (!NCHW && (C % N) == 0 && (Kpad % N) == 0) {
    const int vpr = Kpad / N;
    const int64_t total = (int64_t)B * Ho * Wo * vpr;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
      const int k = (int)(v % vpr) * N;
      const int64_t r = v / vpr;
      float o[N];
#pragma unroll
      for (int j = 0; j < N; ++j) o[j] = 0.f;
      if (k < K) {
        const int tap = k / C, c = k % C;
        const int kh = tap / KW, kw = tap % KW;
        const int ow = (int)(r % Wo), oh = (int)((r / Wo) % Ho);
        const int64_t b = r / ((int64_t)Wo * Ho);
        const int h = oh * stride - pad + kh, w = ow * stride - pad + kw;
        if (h >= 0 && h < H && w >= 0 && w < W) load_vec<T>(x + ((b * H + h) * W + w) * C + c, o);
      }
      store_vec<T>(col + r * Kpad + k, o);
    }
    return;
  }
  const int64_t total = (int64_t)B * Ho * Wo * Kpad;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int k = (int)(e % Kpad);
    const int64_t r = e / Kpad;
    float v = 0.f;
    if (k < K) {
      const int tap = k / C, c = k % C;
      const int kh = tap / KW, kw = tap % KW;
      const int ow = (int)(r % Wo), oh = (int)((r / Wo) % Ho);
      const int64_t b = r / ((int64_t)Wo * Ho);
      const int h = oh * stride - pad + kh, w = ow * stride - pad + kw;
      if (h >= 0 && h < H && w >= 0 && w < W)
        v = NCHW ? ld1<T>(x + ((b * C + c) * H + h) * W + w) : ld1<T>(x + ((b * H + h) * W + w) * C + c);
    }
    st1<T>(col + e, v);
  }
}

// The first convolution reads the [B, C, H, W] image itself (C = 3, 7 x 7 taps, stride 2: Kpad = 152 columns per output position).
// One workgroup builds the rows of 64 consecutive output positions in LDS -- item (position, kh, c) gathers its KW taps, which
// are consecutive pixels of one image row (lanes run over positions: neighbouring lanes read neighbouring pixels) -- and then
// writes the 64 x Kpad block, contiguous in col, with 16-byte stores.  (The element-per-thread form above: 251 us for the
// 32 x 3 x 384 x 384 batch of cfg-2b, 358 MB at 1.4 TB/s, six integer divisions per element.)
template <typename T>
__global__ __launch_bounds__(256) void im2col_nchw_lds_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H, int W,
                                                              int C, int KH, int KW, int stride, int pad, int Ho, int Wo,
                                                              int Kpad, int64_t rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char im2col_smem[];
  T* tile = reinterpret_cast<T*>(im2col_smem);               // [64][Kpad]
  const int K = KH * KW * C;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  T zero;
  st1<T>(&zero, 0.f);
  for (int it = threadIdx.x; it < 64 * KH * C; it += 256) {
    const int pos = it & 63, kc = it >> 6;
    const int kh = kc / C, c = kc - kh * C;
    const int64_t r = r0 + pos;
    T* dst = tile + pos * Kpad + kh * KW * C + c;
    const int ow = (int)(r % Wo), oh = (int)((r / Wo) % Ho);
    const int64_t b = r / ((int64_t)Wo * Ho);
    const int h = oh * stride - pad + kh;
    const bool row_ok = r < rows && h >= 0 && h < H;
    const T* src = x + ((b * C + c) * H + (row_ok ? h : 0)) * W;
    const int w0 = ow * stride - pad;
    for (int kw = 0; kw < KW; ++kw) {
      const int w = w0 + kw;
      dst[kw * C] = (row_ok && w >= 0 && w < W) ? src[w] : zero;
    }
  }
  for (int it = threadIdx.x; it < 64 * (Kpad - K); it += 256) {    // the zero tail of each row
    const int pos = it & 63, k = K + (it >> 6);
    tile[pos * Kpad + k] = zero;
  }
  __syncthreads();
  const int64_t left = rows - r0;
  const int nrow = (int)(left < 64 ? left : 64);
  const int nv = nrow * Kpad * (int)sizeof(T) / 16;          // (Kpad * sizeof(T) is a multiple of 16)
  uint4* out = reinterpret_cast<uint4*>(col + r0 * Kpad);
  const uint4* in = reinterpret_cast<const uint4*>(tile);
  for (int i = threadIdx.x; i < nv; i += 256) out[i] = in[i];
}

// dx[b, h, w, c] = sum over taps (kh, kw) with oh = (h + p - kh)/s, ow = (w + p - kw)/s integral and in range of
//                  dcol[(b, oh, ow)][(kh*KW + kw)*C + c]                (NHWC only; C % vector width == 0)
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int B, int H, int W,
                                                     int C, int KH, int KW, int stride, int pad, int Ho, int Wo, int Kpad) {
  constexpr int N = Vec<T>::N;
  const int vpc = C / N;
  const int64_t total = (int64_t)B * H * W * vpc;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int c = (int)(v % vpc) * N;
    const int64_t p = v / vpc;
    const int w = (int)(p % W), h = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < KH; ++kh) {
      const int hh = h + pad - kh;
      if (hh < 0 || hh % stride) continue;
      const int oh = hh / stride;
      if (oh >= Ho) continue;
      for (int kw = 0; kw < KW; ++kw) {
        const int ww = w + pad - kw;
        if (ww < 0 || ww % stride) continue;
        const int ow = ww / stride;
        if (ow >= Wo) continue;
        float t[N];
        load_vec<T>(dcol + ((b * Ho + oh) * Wo + ow) * Kpad + (kh * KW + kw) * C + c, t);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] += t[j];
      }
    }
    store_vec<T>(dx + p * C + c, acc);
  }
}

// ---------------------------------------------------------------------------------------------- column statistics
// partial[g][0][c] = sum_r a[r][c],  partial[g][1][c] = sum_r a[r][c]*b[r][c]   over the rows of group g, where
//   MODE 0 (forward):  a = x,              b = x                      -> sum x, sum x^2
//   MODE 1 (backward): a = g = dy*[y>0],   b = xhat = (x-mean)*rstd   -> sum g, sum g*xhat
//   MODE 2: MODE 1 with the ReLU gate recomputed from x (no residual joins the sum there) instead of read from y
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_colstat_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const T* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, double* __restrict__ partial,
                                                         int64_t rows, int C, int relu, int cw_log2,
                                                         const T* __restrict__ gamma = nullptr, const T* __restrict__ beta = nullptr) {
  constexpr int N = Vec<T>::N;
  // a block is cw column vectors (32, or C/N when the layer is narrower: 64 channels are 8 bf16 vectors -- with a fixed
  // 32 x 8 shape three quarters of the threads of those layers had no column) x 256/cw row lanes
  __shared__ double red[2][256][N];
  const int cw = 1 << cw_log2, rl = 256 >> cw_log2;
  const int cx = threadIdx.x & (cw - 1), ry = threadIdx.x >> cw_log2;
  const int c = (blockIdx.x * cw + cx) * N;
  // double accumulators: BatchNorm over few values per channel (B*h*w = 32 on a 64x64 image) makes the backward chain
  // ill-conditioned; fp64 VALU is full rate on this part and the kernel is HBM-bound anyway
  double s0[N], s1[N];
  float mu[N], rs[N], gm[N], bt[N];
#pragma unroll
  for (int j = 0; j < N; ++j) { s0[j] = s1[j] = 0.0; mu[j] = 0.f; rs[j] = 1.f; gm[j] = 1.f; bt[j] = 0.f; }
  // relu == 2: the ReLU gate is recomputed from x -- [(x - mean) * rstd * gamma + beta > 0], bn_apply's own expression -- instead of
  // read from the layer's output y (no residual joins the sum there): one tensor less to stream in both backward passes
  constexpr bool regate = MODE == 2;
  if (c < C) {
    if (MODE >= 1) {
#pragma unroll
      for (int j = 0; j < N; ++j) { mu[j] = mean[c + j]; rs[j] = rstd[c + j]; }
      if (regate) {
        load_vec<T>(gamma + c, gm);
        load_vec<T>(beta + c, bt);
      }
    }
    const int64_t step = (int64_t)gridDim.y * rl;
    auto add = [&](const float (&a)[N], const float (&b)[N], const float (&yy)[N]) {
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) { s0[j] += (double)b[j]; s1[j] += (double)b[j] * (double)b[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const bool on = regate ? ((b[j] - mu[j]) * rs[j] * gm[j] + bt[j] > 0.f) : (!relu || yy[j] > 0.f);
          const float g = on ? a[j] : 0.f;
          s0[j] += (double)g;
          s1[j] += (double)g * (double)((b[j] - mu[j]) * rs[j]);
        }
      }
    };
    int64_t r = (int64_t)blockIdx.y * rl + ry;
    for (; r + step < rows; r += 2 * step) {            // two rows per trip: every load of both is in flight before the adds
      float a0[N], b0[N], y0[N], a1[N], b1[N], y1[N];
      load_vec<T>(x + r * C + c, b0);
      load_vec<T>(x + (r + step) * C + c, b1);
      if (MODE >= 1) {
        load_vec<T>(dy + r * C + c, a0);
        load_vec<T>(dy + (r + step) * C + c, a1);
        if (MODE == 1 && relu) {
          load_vec<T>(y + r * C + c, y0);
          load_vec<T>(y + (r + step) * C + c, y1);
        }
      }
      add(a0, b0, y0);
      add(a1, b1, y1);
    }
    if (r < rows) {
      float a0[N], b0[N], y0[N];
      load_vec<T>(x + r * C + c, b0);
      if (MODE >= 1) {
        load_vec<T>(dy + r * C + c, a0);
        if (MODE == 1 && relu) load_vec<T>(y + r * C + c, y0);
      }
      add(a0, b0, y0);
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) { red[0][threadIdx.x][j] = s0[j]; red[1][threadIdx.x][j] = s1[j]; }
  __syncthreads();
  // every thread folds whole (quantity, column) outputs over the row lanes in a fixed order (2 * cw * N outputs: 128 .. 512)
  constexpr int LN = N == 8 ? 3 : 2;
  static_assert((1 << LN) == N, "vector width");
  const int per_q = cw << LN;
  for (int o = threadIdx.x; o < 2 * per_q; o += 256) {
    const int q = o >> (cw_log2 + LN), rem = o & (per_q - 1);
    const int col = rem >> LN, j = rem & (N - 1);
    const int cc = (blockIdx.x * cw + col) * N;
    if (cc < C) {
      double t = 0.0;
      for (int l = 0; l < rl; ++l) t += red[q][l * cw + col][j];
      partial[((int64_t)blockIdx.y * 2 + q) * C + cc + j] = t;
    }
  }
}

// fold the per-group fp64 partials of one quantity pair: 16 columns x 16 group-lanes per 256-thread block; a lane's (up to 16)
// groups are all loaded before the first add -- one memory round trip instead of a serial chain of 32 (the kernel was 6.5 us)
__device__ __forceinline__ void bn_fold_groups(const double* __restrict__ partial, int groups, int C, int c, int gl, double& s,
                                               double& q) {
  __shared__ double red[2][16][16];
  double sa = 0.0, sb = 0.0;
  // 256 groups per trip (the statistics kernels write at most 256; the partial rows a convolution's GEMM epilogue leaves -- one per
  // wave block of 32 .. 128 rows -- can be more)
  for (int g0 = 0; g0 < groups; g0 += 256) {
    double a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int g = g0 + gl + 16 * i;
      const bool ok = c < C && g < groups;
      a[i] = ok ? partial[((int64_t)g * 2) * C + c] : 0.0;
      b[i] = ok ? partial[((int64_t)g * 2 + 1) * C + c] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { sa += a[i]; sb += b[i]; }
  }
  red[0][gl][threadIdx.x & 15] = sa;
  red[1][gl][threadIdx.x & 15] = sb;
  __syncthreads();
  s = q = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s += red[0][r][threadIdx.x & 15]; q += red[1][r][threadIdx.x & 15]; }
}

// forward finalize: mean / rstd of the batch + running statistics (torch: running = (1-m)*running + m*stat, unbiased var)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partial, int groups, int C, int64_t rows,
                                                          float eps, float momentum, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var,
                                                          const double* __restrict__ rows_dev = nullptr) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), gl = threadIdx.x >> 4;
  double s, q;
  bn_fold_groups(partial, groups, C, c, gl, s, q);
  if (gl != 0 || c >= C) return;
  const double R = rows_dev ? *rows_dev : (double)rows;      // (SyncBatchNorm: the all-reduced row count lives on the device)
  const double m = s / R;
  double var = q / R - m * m;
  var = var < 0.0 ? 0.0 : var;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = R > 1.0 ? var * R / (R - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// SyncBatchNorm forward: this rank's sums, [2][C] fp64 (sum x, sum x^2), for the all-reduce between ofa_batchnorm_fwd_stats and
// ofa_batchnorm_fwd_apply
__global__ __launch_bounds__(256) void bn_fold_sums_kernel(const double* __restrict__ partial, int groups, int C,
                                                           double* __restrict__ sums, int64_t rows) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), gl = threadIdx.x >> 4;
  double s, q;
  bn_fold_groups(partial, groups, C, c, gl, s, q);
  if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * C] = (double)rows;      // the row count travels with the sums
  if (gl != 0 || c >= C) return;
  sums[c] = s;
  sums[C + c] = q;
}

// eval mode: statistics come from the running buffers
__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* __restrict__ running_mean,
                                                            const float* __restrict__ running_var, float eps, int C,
                                                            float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// backward finalize: sums[0][c] = sum g, sums[1][c] = sum g*xhat; parameter gradients in the parameter dtype
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int groups, int C,
                                                              float* __restrict__ sums, T* __restrict__ dgamma,
                                                              T* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), gl = threadIdx.x >> 4;
  double s, q;
  bn_fold_groups(partial, groups, C, c, gl, s, q);
  if (gl != 0 || c >= C) return;
  sums[c] = (float)s;
  sums[C + c] = (float)q;
  if (dgamma) {
    st1<T>(dgamma + c, (accumulate ? ld1<T>(dgamma + c) : 0.f) + (float)q);
    st1<T>(dbeta + c, (accumulate ? ld1<T>(dbeta + c) : 0.f) + (float)s);
  }
}

// y = [relu]( (x - mean)*rstd*gamma + beta [+ residual] )
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                       const T* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const T* __restrict__ residual,
                                                       T* __restrict__ y, int64_t rows, int C, int relu) {
  constexpr int N = Vec<T>::N;
  const int vpr = C / N;
  const int64_t total = rows * vpr;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int c = (int)(v % vpr) * N;
    float a[N], g[N], b[N], r[N];
    load_vec<T>(x + v * N, a);
    load_vec<T>(gamma + c, g);
    load_vec<T>(beta + c, b);
    if (residual) load_vec<T>(residual + v * N, r);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float t = (a[j] - mean[c + j]) * rstd[c + j] * g[j] + b[j];
      if (residual) t += r[j];
      a[j] = relu ? fmaxf(t, 0.f) : t;
    }
    store_vec<T>(y + v * N, a);
  }
}

// N consecutive fp32 values (N = 4 or 8; p is 16-byte aligned: channel offsets are multiples of N)
template <int N> __device__ __forceinline__ void ldf(const float* __restrict__ p, float (&o)[N]) {
#pragma unroll
  for (int q = 0; q < N / 4; ++q) {
    const float4 v = reinterpret_cast<const float4*>(p)[q];
    o[4 * q] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
  }
}

// g = dy*[y>0];  dx = gamma*rstd*(g - [batch_stats] (sum_g + xhat*sum_gx)/rows);  dres = g (optional)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                        const T* __restrict__ x, const T* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ sums, T* __restrict__ dx,
                                                        T* __restrict__ dres, int64_t rows, int C, int relu,
                                                        int batch_stats, const T* __restrict__ beta = nullptr,
                                                        const double* __restrict__ stat_rows = nullptr) {
  constexpr int N = Vec<T>::N;
  const int vpr = C / N;
  const int64_t total = rows * vpr;
  const float inv = (float)(1.0 / (stat_rows ? *stat_rows : (double)rows));   // (SyncBatchNorm: the statistics cover every rank's rows)
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int c = (int)(v % vpr) * N;
    float g[N], xx[N], gm[N], o[N], mu[N], rs[N], sg[N], sx[N];
    load_vec<T>(dy + v * N, g);
    load_vec<T>(x + v * N, xx);
    load_vec<T>(gamma + c, gm);
    // the per-channel statistics as 16-byte loads, all issued here: read one by one inside the loop below (behind the
    // batch_stats branch) they were 4 dependent round trips per element -- 1.5 TB/s where bn_apply runs 6.8
    ldf<N>(mean + c, mu);
    ldf<N>(rstd + c, rs);
    ldf<N>(sums + c, sg);
    ldf<N>(sums + C + c, sx);
    if (relu == 2) {                                      // gate recomputed from x (see bn_colstat_kernel)
      float bt[N];
      load_vec<T>(beta + c, bt);
#pragma unroll
      for (int j = 0; j < N; ++j) g[j] = ((xx[j] - mu[j]) * rs[j] * gm[j] + bt[j] > 0.f) ? g[j] : 0.f;
    } else if (relu) {
      float yy[N];
      load_vec<T>(y + v * N, yy);
#pragma unroll
      for (int j = 0; j < N; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float xh = (xx[j] - mu[j]) * rs[j];
      const float t = batch_stats ? g[j] - (sg[j] + xh * sx[j]) * inv : g[j];
      o[j] = gm[j] * rs[j] * t;
    }
    store_vec<T>(dx + v * N, o);
    if (dres) store_vec<T>(dres + v * N, g);
  }
}

// ---------------------------------------------------------------------------------------------- MaxPool2d(k, s, p), NHWC
// One thread per (position, channel vector of N): 16-byte accesses on the activations, N bytes on the arg-max taps (the one
// element per thread form ran the backward at 1.2 TB/s: 233 us for the 32 x 192 x 192 x 64 stem of cfg-2b).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ arg,
                                                          int B, int H, int W, int C, int K, int stride, int pad, int Ho,
                                                          int Wo) {
  constexpr int N = Vec<T>::N;
  const int vpc = C / N;
  const int64_t total = (int64_t)B * Ho * Wo * vpc;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int c = (int)(v % vpc) * N;
    const int64_t r = v / vpc;
    const int ow = (int)(r % Wo), oh = (int)((r / Wo) % Ho);
    const int64_t b = r / ((int64_t)Wo * Ho);
    float best[N];
    uint8_t bi[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    for (int kh = 0; kh < K; ++kh) {
      const int h = oh * stride - pad + kh;
      if (h < 0 || h >= H) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int w = ow * stride - pad + kw;
        if (w < 0 || w >= W) continue;
        float t[N];
        load_vec<T>(x + ((b * H + h) * W + w) * C + c, t);
#pragma unroll
        for (int j = 0; j < N; ++j)
          if (t[j] > best[j] || (t[j] != t[j])) { best[j] = t[j]; bi[j] = (uint8_t)(kh * K + kw); }   // first maximum wins (torch), NaN propagates
      }
    }
    store_vec<T>(y + r * C + c, best);
#pragma unroll
    for (int j = 0; j < N; ++j) arg[r * C + c + j] = bi[j];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                          T* __restrict__ dx, int B, int H, int W, int C, int K, int stride,
                                                          int pad, int Ho, int Wo) {
  constexpr int N = Vec<T>::N;
  const int vpc = C / N;
  const int64_t total = (int64_t)B * H * W * vpc;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int c = (int)(v % vpc) * N;
    const int64_t p = v / vpc;
    const int w = (int)(p % W), h = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < K; ++kh) {
      const int hh = h + pad - kh;
      if (hh < 0 || hh % stride) continue;
      const int oh = hh / stride;
      if (oh >= Ho) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int ww = w + pad - kw;
        if (ww < 0 || ww % stride) continue;
        const int ow = ww / stride;
        if (ow >= Wo) continue;
        const int64_t o = ((b * Ho + oh) * Wo + ow) * C + c;
        float t[N];
        load_vec<T>(dy + o, t);
        uint8_t a8[N];
        __builtin_memcpy(a8, arg + o, N);                     // (o is a multiple of N: C % N == 0)
#pragma unroll
        for (int j = 0; j < N; ++j)
          if (a8[j] == kh * K + kw) acc[j] += t[j];
      }
    }
    store_vec<T>(dx + p * C + c, acc);
  }
}

// y = max(x, 0) [* mask of y for backward]
template <typename T>
__global__ __launch_bounds__(256) void relu_kernel(const T* __restrict__ x, const T* __restrict__ gate, T* __restrict__ y,
                                                   int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = ld1<T>(x + i);
    st1<T>(y + i, gate ? (ld1<T>(gate + i) > 0.f ? v : 0.f) : fmaxf(v, 0.f));
  }
}

}  // namespace ofa
using namespace ofa;

#define OFA_DT(name) OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, name ": bad dtype %d", dtype)
extern "C" int ofa_conv_out_size(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

extern "C" int ofa_im2col(const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int Kpad,
                          int x_nchw, int dtype, void* stream) {
  OFA_DT("im2col");
  OFA_REQUIRE(x && col && B > 0 && H > 0 && W > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && Kpad >= KH * KW * C,
              OFA_ERR_INVALID, "im2col: bad argument");
  const int Ho = ofa_conv_out_size(H, KH, stride, pad), Wo = ofa_conv_out_size(W, KW, stride, pad);
  OFA_REQUIRE(Ho > 0 && Wo > 0, OFA_ERR_INVALID, "im2col: empty output (H=%d W=%d k=%dx%d s=%d p=%d)", H, W, KH, KW, stride, pad);
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * Ho * Wo * Kpad;
  const int es = dtype == OFA_F32 ? 4 : 2;
  if (x_nchw && (Kpad * es) % 16 == 0 && (size_t)64 * Kpad * es <= 48 * 1024 && (int64_t)B * Ho * Wo <= ((int64_t)1 << 36)) {
    const int64_t rows = (int64_t)B * Ho * Wo;
    const dim3 g((unsigned)((rows + 63) / 64)), blk(256);
    const size_t lds = (size_t)64 * Kpad * es;
    if (dtype == OFA_F32) hipLaunchKernelGGL((im2col_nchw_lds_kernel<float>), g, blk, lds, st, (const float*)x, (float*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, rows);
    else if (dtype == OFA_BF16) hipLaunchKernelGGL((im2col_nchw_lds_kernel<bf16_t>), g, blk, lds, st, (const bf16_t*)x, (bf16_t*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, rows);
    else hipLaunchKernelGGL((im2col_nchw_lds_kernel<f16_t>), g, blk, lds, st, (const f16_t*)x, (f16_t*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, rows);
    return check_launch("im2col_nchw");
  }
  dim3 grid(grid_1d(total / 4)), block(256);
  if (dtype == OFA_F32) {
    if (x_nchw) hipLaunchKernelGGL((im2col_kernel<float, true>), grid, block, 0, st, (const float*)x, (float*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
    else hipLaunchKernelGGL((im2col_kernel<float, false>), grid, block, 0, st, (const float*)x, (float*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
  } else if (dtype == OFA_BF16) {
    if (x_nchw) hipLaunchKernelGGL((im2col_kernel<bf16_t, true>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
    else hipLaunchKernelGGL((im2col_kernel<bf16_t, false>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
  }
  else {
    if (x_nchw) hipLaunchKernelGGL((im2col_kernel<f16_t, true>), grid, block, 0, st, (const f16_t*)x, (f16_t*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
    else hipLaunchKernelGGL((im2col_kernel<f16_t, false>), grid, block, 0, st, (const f16_t*)x, (f16_t*)col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
  }
  return check_launch("im2col");
}

extern "C" int ofa_col2im(const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int Kpad,
                          int dtype, void* stream) {
  OFA_DT("col2im");
  OFA_REQUIRE(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && Kpad >= KH * KW * C, OFA_ERR_INVALID, "col2im: bad argument");
  const int n = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(C % n == 0 && Kpad % n == 0, OFA_ERR_UNSUPPORTED, "col2im: C=%d / Kpad=%d must be multiples of %d", C, Kpad, n);
  const int Ho = ofa_conv_out_size(H, KH, stride, pad), Wo = ofa_conv_out_size(W, KW, stride, pad);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(grid_1d((int64_t)B * H * W * (C / n))), block(256);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((col2im_kernel<float>), grid, block, 0, st, (const float*)dcol, (float*)dx, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((col2im_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dcol, (bf16_t*)dx, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
  else
    hipLaunchKernelGGL((col2im_kernel<f16_t>), grid, block, 0, st, (const f16_t*)dcol, (f16_t*)dx, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad);
  return check_launch("col2im");
}

// column vectors per statistics block, as a power of two: 32, or fewer when the layer has fewer than 32 vectors of channels
// (narrower blocks for launches of few blocks were tried: [25088 x 512] 30 -> 40 us, nothing gained elsewhere)
static int bn_cw_log2(int vec_cols) {
  int l = 5;
  while (l > 0 && (1 << l) > vec_cols) --l;
  return l;
}
static int bn_groups(int64_t rows) {
  int64_t g = (rows + 127) / 128;
  return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}
extern "C" int ofa_batchnorm_ws_floats(int C) { return 4 * 256 * C + 2 * C; }   // fp64 partials [256][2][C] + fp32 sums [2][C]

// column statistics of x into ws (fp64 partials [groups][2][C]); returns the group count
static int bn_fwd_colstat(const void* x, float* ws, int64_t rows, int C, int dtype, hipStream_t st) {
  const int n = dtype == OFA_F32 ? 4 : 8;
  const int groups = bn_groups(rows);
  const int cwl = bn_cw_log2(C / n);
  dim3 grid(cdiv(C / n, 1 << cwl), groups), block(256);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((bn_colstat_kernel<float, 0>), grid, block, 0, st, (const float*)x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (double*)ws, rows, C, 0, cwl);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((bn_colstat_kernel<bf16_t, 0>), grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (double*)ws, rows, C, 0, cwl);
  else
    hipLaunchKernelGGL((bn_colstat_kernel<f16_t, 0>), grid, block, 0, st, (const f16_t*)x, (const f16_t*)nullptr, (const f16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (double*)ws, rows, C, 0, cwl);
  return groups;
}

static int bn_apply_launch(const void* x, const void* gamma, const void* beta, const void* residual, void* y, const float* mean,
                           const float* rstd, int64_t rows, int C, int relu, int dtype, hipStream_t st) {
  const int n = dtype == OFA_F32 ? 4 : 8;
  dim3 grid(grid_1d(rows * (C / n))), block(256);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float>), grid, block, 0, st, (const float*)x, (const float*)gamma, (const float*)beta, (const float*)mean, (const float*)rstd, (const float*)residual, (float*)y, rows, C, relu);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((bn_apply_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)gamma, (const bf16_t*)beta, (const float*)mean, (const float*)rstd, (const bf16_t*)residual, (bf16_t*)y, rows, C, relu);
  else
    hipLaunchKernelGGL((bn_apply_kernel<f16_t>), grid, block, 0, st, (const f16_t*)x, (const f16_t*)gamma, (const f16_t*)beta, (const float*)mean, (const float*)rstd, (const f16_t*)residual, (f16_t*)y, rows, C, relu);
  return check_launch("batchnorm_apply");
}

// training: batch statistics (+ running update when running_mean != NULL); eval (use_running != 0): running statistics.
extern "C" int ofa_batchnorm_fwd(const void* x, const void* gamma, const void* beta, const void* residual, void* y, float* mean,
                                 float* rstd, float* running_mean, float* running_var, float* ws, int64_t rows, int C,
                                 float eps, float momentum, int use_running, int relu, int dtype, void* stream) {
  OFA_DT("batchnorm_fwd");
  OFA_REQUIRE(x && gamma && beta && y && mean && rstd && ws && rows > 0 && C > 0, OFA_ERR_INVALID, "batchnorm_fwd: bad argument");
  const int n = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(C % n == 0, OFA_ERR_UNSUPPORTED, "batchnorm: C=%d must be a multiple of %d", C, n);
  OFA_REQUIRE(!use_running || (running_mean && running_var), OFA_ERR_INVALID, "batchnorm_fwd: eval mode needs running statistics");
  hipStream_t st = (hipStream_t)stream;
  if (use_running) {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, running_mean, running_var, eps, C, mean, rstd);
  } else {
    const int groups = bn_fwd_colstat(x, ws, rows, C, dtype, st);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, (const double*)ws, groups, C, rows, eps, momentum, mean, rstd, running_mean, running_var);
  }
  int rc = check_launch("batchnorm_stats");
  if (rc) return rc;
  return bn_apply_launch(x, gamma, beta, residual, y, mean, rstd, rows, C, relu, dtype, st);
}

// SyncBatchNorm, forward in two phases around the caller's all-reduce (SUM) of `sums` over the ranks:
//   stats: sums [2*C + 1] fp64 = this rank's (sum x, sum x^2, row count);   apply: statistics from the reduced sums -- the row
//   count is read from sums[2*C] on the device: ranks may hold different numbers of rows and nobody syncs to learn the total --,
//   running buffers updated with them, y for this rank's rows.
extern "C" int ofa_batchnorm_fwd_stats(const void* x, double* sums, float* ws, int64_t rows, int C, int dtype, void* stream) {
  OFA_DT("batchnorm_fwd_stats");
  OFA_REQUIRE(x && sums && ws && rows > 0 && C > 0, OFA_ERR_INVALID, "batchnorm_fwd_stats: bad argument");
  OFA_REQUIRE(C % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "batchnorm: C=%d is not vectorizable", C);
  hipStream_t st = (hipStream_t)stream;
  const int groups = bn_fwd_colstat(x, ws, rows, C, dtype, st);
  hipLaunchKernelGGL(bn_fold_sums_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, (const double*)ws, groups, C, sums, rows);
  return check_launch("batchnorm_fwd_stats");
}

extern "C" int ofa_batchnorm_fwd_apply(const void* x, const void* gamma, const void* beta, const void* residual, void* y,
                                       float* mean, float* rstd, float* running_mean, float* running_var, const double* sums,
                                       int groups, int64_t rows, int C, float eps, float momentum, int relu, int dtype,
                                       void* stream) {
  OFA_DT("batchnorm_fwd_apply");
  OFA_REQUIRE(x && gamma && beta && y && mean && rstd && sums && groups >= 0 && rows > 0 && C > 0, OFA_ERR_INVALID,
              "batchnorm_fwd_apply: bad argument");
  OFA_REQUIRE(C % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "batchnorm: C=%d is not vectorizable", C);
  hipStream_t st = (hipStream_t)stream;
  if (groups == 0)   // SyncBatchNorm: the reduced sums are ONE group of partials over sums[2*C] rows (the count lives on the device)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, sums, 1, C, rows, eps, momentum, mean, rstd, running_mean, running_var, sums + 2 * (int64_t)C);
  else               // `groups` partial rows [groups][2][C] over this launch's `rows` rows, e.g. left by ofa_gemm_colstat's epilogue
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, sums, groups, C, rows, eps, momentum, mean, rstd, running_mean, running_var, (const double*)nullptr);
  int rc = check_launch("batchnorm_fwd_apply");
  if (rc) return rc;
  return bn_apply_launch(x, gamma, beta, residual, y, mean, rstd, rows, C, relu, dtype, st);
}

// backward statistics: sums [2][C] fp32 = (sum g, sum g*xhat) of this launch's rows, and the parameter gradients from them
static int bn_bwd_stats_launch(const void* dy, const void* y, const void* x, const void* gamma, const float* mean, const float* rstd,
                               float* sums, void* dgamma, void* dbeta, float* ws, int64_t rows, int C, int relu, int accumulate,
                               const void* beta, int dtype, hipStream_t st) {
  const int n = dtype == OFA_F32 ? 4 : 8;
  const int groups = bn_groups(rows);
  const int cwl = bn_cw_log2(C / n);
  dim3 grid(cdiv(C / n, 1 << cwl), groups), block(256);
  if (dtype == OFA_F32) {
    if (beta) hipLaunchKernelGGL((bn_colstat_kernel<float, 2>), grid, block, 0, st, (const float*)x, (const float*)dy, (const float*)y, mean, rstd, (double*)ws, rows, C, relu, cwl, (const float*)gamma, (const float*)beta);
    else hipLaunchKernelGGL((bn_colstat_kernel<float, 1>), grid, block, 0, st, (const float*)x, (const float*)dy, (const float*)y, mean, rstd, (double*)ws, rows, C, relu, cwl);
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<float>), dim3(cdiv(C, 16)), dim3(256), 0, st, (const double*)ws, groups, C, sums, (float*)dgamma, (float*)dbeta, accumulate);
  } else if (dtype == OFA_BF16) {
    if (beta) hipLaunchKernelGGL((bn_colstat_kernel<bf16_t, 2>), grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y, mean, rstd, (double*)ws, rows, C, relu, cwl, (const bf16_t*)gamma, (const bf16_t*)beta);
    else hipLaunchKernelGGL((bn_colstat_kernel<bf16_t, 1>), grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y, mean, rstd, (double*)ws, rows, C, relu, cwl);
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<bf16_t>), dim3(cdiv(C, 16)), dim3(256), 0, st, (const double*)ws, groups, C, sums, (bf16_t*)dgamma, (bf16_t*)dbeta, accumulate);
  }
  else {
    if (beta) hipLaunchKernelGGL((bn_colstat_kernel<f16_t, 2>), grid, block, 0, st, (const f16_t*)x, (const f16_t*)dy, (const f16_t*)y, mean, rstd, (double*)ws, rows, C, relu, cwl, (const f16_t*)gamma, (const f16_t*)beta);
    else hipLaunchKernelGGL((bn_colstat_kernel<f16_t, 1>), grid, block, 0, st, (const f16_t*)x, (const f16_t*)dy, (const f16_t*)y, mean, rstd, (double*)ws, rows, C, relu, cwl);
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<f16_t>), dim3(cdiv(C, 16)), dim3(256), 0, st, (const double*)ws, groups, C, sums, (f16_t*)dgamma, (f16_t*)dbeta, accumulate);
  }
  return check_launch("batchnorm_bwd_stats");
}

static int bn_bwd_dx_launch(const void* dy, const void* y, const void* x, const void* gamma, const float* mean, const float* rstd,
                            const float* sums, void* dx, void* dres, int64_t rows, const double* stat_rows, int C, int batch_stats,
                            int relu, const void* beta, int dtype, hipStream_t st) {
  const int n = dtype == OFA_F32 ? 4 : 8;
  dim3 g2(grid_1d(rows * (C / n))), block(256);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((bn_bwd_dx_kernel<float>), g2, block, 0, st, (const float*)dy, (const float*)y, (const float*)x, (const float*)gamma, mean, rstd, (const float*)sums, (float*)dx, (float*)dres, rows, C, relu, batch_stats, (const float*)beta, stat_rows);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((bn_bwd_dx_kernel<bf16_t>), g2, block, 0, st, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, (const bf16_t*)gamma, mean, rstd, (const float*)sums, (bf16_t*)dx, (bf16_t*)dres, rows, C, relu, batch_stats, (const bf16_t*)beta, stat_rows);
  else
    hipLaunchKernelGGL((bn_bwd_dx_kernel<f16_t>), g2, block, 0, st, (const f16_t*)dy, (const f16_t*)y, (const f16_t*)x, (const f16_t*)gamma, mean, rstd, (const float*)sums, (f16_t*)dx, (f16_t*)dres, rows, C, relu, batch_stats, (const f16_t*)beta, stat_rows);
  return check_launch("batchnorm_bwd_dx");
}

// dgamma/dbeta: [C] in `dtype` (accumulate != 0: added); dres (optional): gradient of the residual input (= gated dy).
extern "C" int ofa_batchnorm_bwd(const void* dy, const void* y, const void* x, const void* gamma, const float* mean,
                                 const float* rstd, void* dx, void* dres, void* dgamma, void* dbeta, float* ws, int64_t rows,
                                 int C, int batch_stats, int relu, int accumulate, const void* beta, int dtype, void* stream) {
  OFA_DT("batchnorm_bwd");
  OFA_REQUIRE(dy && x && gamma && mean && rstd && dx && ws && rows > 0 && C > 0 && (!relu || y || beta), OFA_ERR_INVALID, "batchnorm_bwd: bad argument");
  OFA_REQUIRE(!beta || (relu && !dres), OFA_ERR_INVALID, "batchnorm_bwd: the gate can be recomputed from x (beta != NULL) only for a ReLU layer without a residual input");
  if (beta) relu = 2;
  const int n = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(C % n == 0, OFA_ERR_UNSUPPORTED, "batchnorm: C=%d must be a multiple of %d", C, n);
  hipStream_t st = (hipStream_t)stream;
  float* sums = ws + (int64_t)4 * 256 * C;
  int rc = bn_bwd_stats_launch(dy, y, x, gamma, mean, rstd, sums, dgamma, dbeta, ws, rows, C, relu, accumulate, beta, dtype, st);
  if (rc) return rc;
  return bn_bwd_dx_launch(dy, y, x, gamma, mean, rstd, sums, dx, dres, rows, nullptr, C, batch_stats, relu, beta, dtype, st);
}

// SyncBatchNorm, backward in two phases around the caller's all-reduce (SUM) of `sums` [2][C] fp32 over the ranks: the parameter
// gradients are this rank's own sums (the data-parallel gradient exchange adds them up like every other gradient), the input
// gradient uses the reduced sums over total_rows rows (torch.nn.SyncBatchNorm's backward: batch_norm_backward_reduce -> all_reduce
// -> batch_norm_backward_elemt).  total_rows: DEVICE pointer to the reduced row count (element 2*C of the forward's sums).
extern "C" int ofa_batchnorm_bwd_stats(const void* dy, const void* y, const void* x, const void* gamma, const float* mean,
                                       const float* rstd, float* sums, void* dgamma, void* dbeta, float* ws, int64_t rows, int C,
                                       int relu, int accumulate, const void* beta, int dtype, void* stream) {
  OFA_DT("batchnorm_bwd_stats");
  OFA_REQUIRE(dy && x && gamma && mean && rstd && sums && ws && rows > 0 && C > 0 && (!relu || y || beta), OFA_ERR_INVALID, "batchnorm_bwd_stats: bad argument");
  if (beta) relu = 2;
  OFA_REQUIRE(C % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "batchnorm: C=%d is not vectorizable", C);
  return bn_bwd_stats_launch(dy, y, x, gamma, mean, rstd, sums, dgamma, dbeta, ws, rows, C, relu, accumulate, beta, dtype, (hipStream_t)stream);
}

extern "C" int ofa_batchnorm_bwd_dx(const void* dy, const void* y, const void* x, const void* gamma, const float* mean,
                                    const float* rstd, const float* sums, void* dx, void* dres, int64_t rows,
                                    const double* total_rows, int C, int relu, const void* beta, int dtype, void* stream) {
  OFA_DT("batchnorm_bwd_dx");
  OFA_REQUIRE(dy && x && gamma && mean && rstd && sums && dx && rows > 0 && total_rows && C > 0 && (!relu || y || beta), OFA_ERR_INVALID, "batchnorm_bwd_dx: bad argument");
  OFA_REQUIRE(!beta || (relu && !dres), OFA_ERR_INVALID, "batchnorm_bwd_dx: the gate can be recomputed from x (beta != NULL) only for a ReLU layer without a residual input");
  if (beta) relu = 2;
  OFA_REQUIRE(C % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "batchnorm: C=%d is not vectorizable", C);
  return bn_bwd_dx_launch(dy, y, x, gamma, mean, rstd, sums, dx, dres, rows, total_rows, C, 1, relu, beta, dtype, (hipStream_t)stream);
}

extern "C" int ofa_maxpool_fwd(const void* x, void* y, uint8_t* arg, int B, int H, int W, int C, int K, int stride, int pad,
                               int dtype, void* stream) {
  OFA_DT("maxpool_fwd");
  OFA_REQUIRE(x && y && arg && B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && K * K <= 255 && stride > 0 && pad >= 0, OFA_ERR_INVALID, "maxpool_fwd: bad argument");
  const int Ho = ofa_conv_out_size(H, K, stride, pad), Wo = ofa_conv_out_size(W, K, stride, pad);
  hipStream_t st = (hipStream_t)stream;
  const int nvec = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(C % nvec == 0, OFA_ERR_UNSUPPORTED, "maxpool_fwd: C=%d must be a multiple of %d", C, nvec);
  dim3 grid(grid_1d((int64_t)B * Ho * Wo * (C / nvec))), block(256);
  if (dtype == OFA_F32) hipLaunchKernelGGL((maxpool_fwd_kernel<float>), grid, block, 0, st, (const float*)x, (float*)y, arg, B, H, W, C, K, stride, pad, Ho, Wo);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, arg, B, H, W, C, K, stride, pad, Ho, Wo);
  else hipLaunchKernelGGL((maxpool_fwd_kernel<f16_t>), grid, block, 0, st, (const f16_t*)x, (f16_t*)y, arg, B, H, W, C, K, stride, pad, Ho, Wo);
  return check_launch("maxpool_fwd");
}

extern "C" int ofa_maxpool_bwd(const void* dy, const uint8_t* arg, void* dx, int B, int H, int W, int C, int K, int stride, int pad,
                               int dtype, void* stream) {
  OFA_DT("maxpool_bwd");
  OFA_REQUIRE(dy && dx && arg && B > 0 && H > 0 && W > 0 && C > 0 && K > 0, OFA_ERR_INVALID, "maxpool_bwd: bad argument");
  const int Ho = ofa_conv_out_size(H, K, stride, pad), Wo = ofa_conv_out_size(W, K, stride, pad);
  hipStream_t st = (hipStream_t)stream;
  const int nvec = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(C % nvec == 0, OFA_ERR_UNSUPPORTED, "maxpool_bwd: C=%d must be a multiple of %d", C, nvec);
  dim3 grid(grid_1d((int64_t)B * H * W * (C / nvec))), block(256);
  if (dtype == OFA_F32) hipLaunchKernelGGL((maxpool_bwd_kernel<float>), grid, block, 0, st, (const float*)dy, arg, (float*)dx, B, H, W, C, K, stride, pad, Ho, Wo);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dy, arg, (bf16_t*)dx, B, H, W, C, K, stride, pad, Ho, Wo);
  else hipLaunchKernelGGL((maxpool_bwd_kernel<f16_t>), grid, block, 0, st, (const f16_t*)dy, arg, (f16_t*)dx, B, H, W, C, K, stride, pad, Ho, Wo);
  return check_launch("maxpool_bwd");
}

// y = relu(x) (gate == NULL)   or   y = x * [gate > 0]  (backward: x = dy, gate = the forward output)
extern "C" int ofa_relu(const void* x, const void* gate, void* y, int64_t n, int dtype, void* stream) {
  OFA_DT("relu");
  OFA_REQUIRE(n >= 0 && (n == 0 || (x && y)), OFA_ERR_INVALID, "relu: bad argument");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32) hipLaunchKernelGGL((relu_kernel<float>), dim3(grid_1d(n)), dim3(256), 0, st, (const float*)x, (const float*)gate, (float*)y, n);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((relu_kernel<bf16_t>), dim3(grid_1d(n)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)gate, (bf16_t*)y, n);
  else hipLaunchKernelGGL((relu_kernel<f16_t>), dim3(grid_1d(n)), dim3(256), 0, st, (const f16_t*)x, (const f16_t*)gate, (f16_t*)y, n);
  return check_launch("relu");
}
