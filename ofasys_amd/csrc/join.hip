// Residual join of a pre-LN transformer sub-block, one kernel each way:
//     y = residual + dropout_p( LN_a(x) )            LN_a optional (attn_ln / self_attn_ln / cross_attn_ln, scale_attn)
//     z = LN_b(y)                                    LN_b optional (the NEXT block's pre-LayerNorm)
// (reference: module/transformer_layer.py:167-208 / :438-494 -- `x = attn_ln(x); x = dropout(x); x = residual + x;
//  residual = x; x = final_layer_norm(x)` and the same chain at the other sub-block boundaries.)  Op by op that is three
// kernels forward (LayerNorm, dropout+add, LayerNorm) and three backward plus their small reduce launches, every one of
// them a full pass over the [rows, C] activations; fused it is one pass each way.  Rounding points are kept where the
// unfused kernels put them (LN_a output, y, the gradient of y, the dropout gradient are rounded to the storage dtype),
// so the fused and the unfused compositions agree bit for bit (the split-row kernels below; the row-per-wave forms further down
// sum the row statistics in another lane grouping).  4 (forward) / 6 (backward) row passes of HBM traffic -- but at ~1000 wave64
// instructions per 768-column row these kernels are bound by the VALU before they are bound by HBM (round 5).
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Row-per-wave forms: 16-bit rows of 256 * k columns (every embed_dim of the model families: 256 ... 1536).  The kernels above spend their time
// in the VALU, not on HBM (a wave64 instruction holds a 16-lane SIMD for 4 cycles; per 768-column row they issue ~1000 (forward) / ~1200
// (backward) of them, a third of it the Philox generator, with a quarter of the lanes idle in the second vector: 22 / 28 us for passes HBM
// moves in 13 / 17 us).  Here a lane owns NF 8-element chunks plus, for an odd multiple of 256 columns, one 4-element tail chunk, so all 64
// lanes work on every chunk; a row never leaves its wave (no LDS exchange, no block barrier); the forward kernel leaves one byte of keep bits
// per (chunk, lane), so that the backward reads 1 / 16 of a pass instead of re-running the generator; the backward is persistent with the next
// two rows in flight (the raw vectors, ~12 KB per wave).  Rounding points are those of the kernels above; the row statistics are summed in a
// different lane grouping, means are sum * (1 / cols) and rstd is v_rsq_f32 (the split-row kernels divide and take 1 / sqrt: ~40 instructions
// per row): fp32 rounding noise against them, and against csrc/layernorm.hip.
template <int NF_, bool TAIL_> struct RowMap {
  static constexpr int NF = NF_;
  static constexpr bool TAIL = TAIL_;
  static constexpr int EPL = NF * 8 + (TAIL ? 4 : 0);          // elements per lane
  static constexpr int W = EPL / 2;                            // 32-bit words per lane
  static constexpr int NCH = NF + (TAIL ? 1 : 0);              // chunks per lane
  static constexpr int COLS = 64 * EPL;
  static constexpr int TAIL0 = NF * 512;                       // first column of the tail chunks
};

template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t w, float& lo, float& hi) {
  lo = __uint_as_float(w << 16);
  hi = __uint_as_float(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t w, float& lo, float& hi) {
  const f16x2_t h = __builtin_bit_cast(f16x2_t, w);
  lo = (float)h[0];
  hi = (float)h[1];
}

template <typename M, typename T> __device__ __forceinline__ void row_load(const T* __restrict__ p, int lane, uint32_t (&w)[M::W]) {
#pragma unroll
  for (int i = 0; i < M::NF; ++i) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + (i * 64 + lane) * 8);
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  if constexpr (M::TAIL) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + M::TAIL0 + lane * 4);
    w[4 * M::NF] = v.x; w[4 * M::NF + 1] = v.y;
  }
}
template <typename M, typename T> __device__ __forceinline__ void row_store(T* __restrict__ p, int lane, const uint32_t (&w)[M::W]) {
#pragma unroll
  for (int i = 0; i < M::NF; ++i)
    *reinterpret_cast<uint4*>(p + (i * 64 + lane) * 8) = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
  if constexpr (M::TAIL) *reinterpret_cast<uint2*>(p + M::TAIL0 + lane * 4) = make_uint2(w[4 * M::NF], w[4 * M::NF + 1]);
}
template <typename M, typename T> __device__ __forceinline__ void row_unpack(const uint32_t (&w)[M::W], float (&v)[M::EPL]) {
#pragma unroll
  for (int k = 0; k < M::W; ++k) unpack2<T>(w[k], v[2 * k], v[2 * k + 1]);
}
// rounds v to the storage type: w = the packed row, v = what a reader of w sees
template <typename M, typename T> __device__ __forceinline__ void row_round(float (&v)[M::EPL], uint32_t (&w)[M::W]) {
#pragma unroll
  for (int k = 0; k < M::W; ++k) {
    w[k] = pack2<T>(v[2 * k], v[2 * k + 1]);
    unpack2<T>(w[k], v[2 * k], v[2 * k + 1]);
  }
}
// keep bits of the lane's chunks: bit j of kb[i] <- element j of chunk i survives (keep_mask's rule: the same Philox positions as every
// other dropout kernel of the library)
template <typename M> __device__ __forceinline__ void row_keep_bits(const Philox& rng, uint64_t off, int64_t row0, int lane, float p,
                                                                    uint32_t (&kb)[M::NCH]) {
#pragma unroll
  for (int i = 0; i < M::NF; ++i) {
    bool keep[8];
    keep_mask<8>(rng, off, row0 + (i * 64 + lane) * 8, p, keep);
    uint32_t b = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) b |= (keep[j] ? 1u : 0u) << j;
    kb[i] = b;
  }
  if constexpr (M::TAIL) {
    bool keep[4];
    keep_mask<4>(rng, off, row0 + M::TAIL0 + lane * 4, p, keep);
    uint32_t b = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) b |= (keep[j] ? 1u : 0u) << j;
    kb[M::NF] = b;
  }
}
// v <- keep ? v * scale : 0 over the lane's chunks
template <typename M> __device__ __forceinline__ void row_drop(float (&v)[M::EPL], const uint32_t (&kb)[M::NCH], float scale) {
#pragma unroll
  for (int e = 0; e < M::EPL; ++e) v[e] = (kb[e >> 3] >> (e & 7)) & 1u ? v[e] * scale : 0.f;
}
template <typename M> __device__ __forceinline__ float row_sum(const float (&v)[M::EPL]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < M::EPL; ++e) s += v[e];
  return s;
}

template <typename T, typename M>
__global__ __launch_bounds__(256) void join_fwd_row_kernel(const T* __restrict__ x, const T* __restrict__ res, const T* __restrict__ ga,
                                                           const T* __restrict__ ba, const T* __restrict__ gb, const T* __restrict__ bb,
                                                           T* __restrict__ y, T* __restrict__ z, float* __restrict__ stats,
                                                           uint8_t* __restrict__ keep_bits, int64_t rows, float eps, JoinRng rg) {
  constexpr int EPL = M::EPL, W = M::W, cols = M::COLS;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t row0 = row * cols;
  uint32_t xw[W], rw[W];
  row_load<M, T>(x + row0, lane, xw);
  if (res) {
    row_load<M, T>(res + row0, lane, rw);
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) rw[k] = 0;
  }
  float v[EPL];
  row_unpack<M, T>(xw, v);
  if (ga) {                                             // LN_a, two-pass statistics in registers
    const float mu = wave_sum(row_sum<M>(v)) * (1.0f / cols);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) q += (v[e] - mu) * (v[e] - mu);
    const float rs = __builtin_amdgcn_rsqf(wave_sum(q) * (1.0f / cols) + eps);
    if (lane == 0) { stats[row] = mu; stats[rows + row] = rs; }
    uint32_t gw[W], bw[W];
    row_load<M, T>(ga, lane, gw);
    row_load<M, T>(ba, lane, bw);
    float g[EPL], b[EPL];
    row_unpack<M, T>(gw, g);
    row_unpack<M, T>(bw, b);
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = (v[e] - mu) * rs * g[e] + b[e];
    row_round<M, T>(v, xw);                             // (the unfused LayerNorm stores its output: one rounding)
  }
  if (rg.p > 0.f) {
    const uint64_t off = rg.offset + (rg.base ? (uint64_t)rg.base[0] : 0);
    uint32_t kb[M::NCH];
    row_keep_bits<M>(Philox(rg.seed), off, row0, lane, rg.p, kb);
    row_drop<M>(v, kb, 1.0f / (1.0f - rg.p));
    if (keep_bits) {
#pragma unroll
      for (int i = 0; i < M::NCH; ++i) keep_bits[(row * M::NCH + i) * 64 + lane] = (uint8_t)kb[i];
    }
  }
  float r[EPL];
  row_unpack<M, T>(rw, r);
#pragma unroll
  for (int e = 0; e < EPL; ++e) v[e] += r[e];
  row_round<M, T>(v, xw);
  row_store<M, T>(y + row0, lane, xw);
  if (!gb) return;
  const float mu = wave_sum(row_sum<M>(v)) * (1.0f / cols);            // LN_b of the (rounded) y
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) q += (v[e] - mu) * (v[e] - mu);
  const float rs = __builtin_amdgcn_rsqf(wave_sum(q) * (1.0f / cols) + eps);
  if (lane == 0) { stats[2 * rows + row] = mu; stats[3 * rows + row] = rs; }
  uint32_t gw[W], bw[W];
  row_load<M, T>(gb, lane, gw);
  row_load<M, T>(bb, lane, bw);
  float g[EPL], b[EPL];
  row_unpack<M, T>(gw, g);
  row_unpack<M, T>(bw, b);
#pragma unroll
  for (int e = 0; e < EPL; ++e) v[e] = (v[e] - mu) * rs * g[e] + b[e];
#pragma unroll
  for (int k = 0; k < W; ++k) xw[k] = pack2<T>(v[2 * k], v[2 * k + 1]);
  row_store<M, T>(z + row0, lane, xw);
}

template <typename M> struct JoinRowRaw {                // one row as fetched: raw vectors, keep bytes, the four statistics
  uint32_t dy[M::W], dz[M::W], y[M::W], x[M::W];
  uint32_t kb[M::NCH];
  float st[4];
};

// HAS_A / HAS_B / DROP: LN_a, LN_b, p > 0 as compile-time facts (as run-time flags every one of them costs register copies at the joins
// of the row loop: a third more instructions per row)
template <typename T, typename M, int WPB, int DIST, bool HAS_A, bool HAS_B, bool DROP>
__global__ __launch_bounds__(WPB * 64) void join_bwd_row_kernel(const T* __restrict__ dy, const T* __restrict__ dz,
                                                                        const T* __restrict__ x, const T* __restrict__ y,
                                                                        const T* __restrict__ ga, const T* __restrict__ gb,
                                                                        const float* __restrict__ stats,
                                                                        const uint8_t* __restrict__ keep_bits, T* __restrict__ dres,
                                                                        T* __restrict__ dx, float* __restrict__ ws, int64_t rows,
                                                                        JoinRng rg, int want_xsum) {
  constexpr int EPL = M::EPL, W = M::W, cols = M::COLS;
  __shared__ float fold[WPB][cols];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t gAw[W], gBw[W];                               // the gammas stay packed (registers: three raw rows are in flight)
  float acc[5][EPL];                                     // dgamma_a, dbeta_a, dgamma_b, dbeta_b, column sums of dx
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q][e] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < W; ++k) gAw[k] = gBw[k] = 0;
  if constexpr (HAS_A) row_load<M, T>(ga, lane, gAw);
  if constexpr (HAS_B) row_load<M, T>(gb, lane, gBw);
  const float scale = DROP ? 1.0f / (1.0f - rg.p) : 1.0f;
  const int64_t stride = (int64_t)gridDim.x * WPB;
  const int64_t first = (int64_t)blockIdx.x * WPB + wib;
  const int64_t niter = (rows + stride - 1) / stride;

  auto fetch = [&](JoinRowRaw<M>& b, int64_t it) {
    const int64_t row = it * stride + first;
    if (row < rows) {                                     // (wave-uniform)
      const int64_t row0 = row * cols;
      if (dy) row_load<M, T>(dy + row0, lane, b.dy);
      if constexpr (HAS_B) {
        row_load<M, T>(dz + row0, lane, b.dz);
        row_load<M, T>(y + row0, lane, b.y);
        b.st[2] = stats[2 * rows + row];
        b.st[3] = stats[3 * rows + row];
      }
      if constexpr (HAS_A) {
        row_load<M, T>(x + row0, lane, b.x);
        b.st[0] = stats[row];
        b.st[1] = stats[rows + row];
      }
      if constexpr (DROP) {
#pragma unroll
        for (int i = 0; i < M::NCH; ++i) b.kb[i] = keep_bits[(row * M::NCH + i) * 64 + lane];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                    // the loads stay HERE, two rows ahead of their use
  };
  auto process = [&](JoinRowRaw<M>& b, int64_t it) {
    const int64_t row = it * stride + first;
    if (row >= rows) return;
    const int64_t row0 = row * cols;
    float g[EPL];                                         // running gradient of this lane's columns
    uint32_t w[W];
    if (dy) {
      row_unpack<M, T>(b.dy, g);
    } else {
#pragma unroll
      for (int e = 0; e < EPL; ++e) g[e] = 0.f;
    }
    if constexpr (HAS_B) {                                // LN_b backward
      float dzv[EPL], yh[EPL], gB[EPL], s1 = 0.f, s2 = 0.f;
      row_unpack<M, T>(b.dz, dzv);
      row_unpack<M, T>(b.y, yh);
      row_unpack<M, T>(gBw, gB);
      const float mu = b.st[2], rs = b.st[3];
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        yh[e] = (yh[e] - mu) * rs;
        const float gy = dzv[e] * gB[e];
        s1 += gy;
        s2 += gy * yh[e];
        acc[2][e] += dzv[e] * yh[e];
        acc[3][e] += dzv[e];
      }
      s1 = wave_sum(s1) * (1.0f / cols);
      s2 = wave_sum(s2) * (1.0f / cols);
#pragma unroll
      for (int e = 0; e < EPL; ++e) g[e] = rs * (dzv[e] * gB[e] - s1 - yh[e] * s2) + g[e];   // LN_b input gradient + dy, rounded once
      row_round<M, T>(g, w);
      if (dres) row_store<M, T>(dres + row0, lane, w);
    } else if (dres) {                                    // gradient of the residual input == gradient of y
      row_store<M, T>(dres + row0, lane, b.dy);
    }
    if constexpr (DROP) {                                 // dropout backward on the rounded gradient
      row_drop<M>(g, b.kb, scale);
      row_round<M, T>(g, w);
    }
    if constexpr (HAS_A) {                                // LN_a backward
      float xh[EPL], gA[EPL], s1 = 0.f, s2 = 0.f;
      row_unpack<M, T>(b.x, xh);
      row_unpack<M, T>(gAw, gA);
      const float mu = b.st[0], rs = b.st[1];
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        xh[e] = (xh[e] - mu) * rs;
        const float gy = g[e] * gA[e];
        s1 += gy;
        s2 += gy * xh[e];
        acc[0][e] += g[e] * xh[e];
        acc[1][e] += g[e];
      }
      s1 = wave_sum(s1) * (1.0f / cols);
      s2 = wave_sum(s2) * (1.0f / cols);
#pragma unroll
      for (int e = 0; e < EPL; ++e) g[e] = rs * (g[e] * gA[e] - s1 - xh[e] * s2);
#pragma unroll
      for (int k = 0; k < W; ++k) w[k] = pack2<T>(g[2 * k], g[2 * k + 1]);
    } else if (!DROP && !HAS_B) {
#pragma unroll
      for (int k = 0; k < W; ++k) w[k] = b.dy[k];
    }
    row_store<M, T>(dx + row0, lane, w);
    if (want_xsum) {                                      // gradient of the bias of the Linear that produced x
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[4][e] += g[e];
    }
  };

  if constexpr (DIST == 2) {                                // two rows ahead
    JoinRowRaw<M> b0, b1, b2;
    fetch(b0, 0);
    fetch(b1, 1);
    for (int64_t it = 0; it < niter; it += 3) {
      fetch(b2, it + 2); process(b0, it);
      fetch(b0, it + 3); process(b1, it + 1);
      fetch(b1, it + 4); process(b2, it + 2);
    }
  } else {                                                  // one row ahead
    JoinRowRaw<M> b0, b1;
    fetch(b0, 0);
    for (int64_t it = 0; it < niter; it += 2) {
      fetch(b1, it + 1); process(b0, it);
      fetch(b0, it + 2); process(b1, it + 1);
    }
  }
  // the block's waves -> one partial row per quantity
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    if ((q < 2 && !HAS_A) || (q >= 2 && q < 4 && !HAS_B) || (q == 4 && !want_xsum)) continue;
#pragma unroll
    for (int i = 0; i < M::NF; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) fold[wib][(i * 64 + lane) * 8 + j] = acc[q][i * 8 + j];
    if constexpr (M::TAIL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) fold[wib][M::TAIL0 + lane * 4 + j] = acc[q][M::NF * 8 + j];
    }
    __syncthreads();
    float* wq = ws + ((int64_t)q * gridDim.x + blockIdx.x) * cols;
    for (int c = threadIdx.x; c < cols; c += WPB * 64) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < WPB; ++r) a += fold[r][c];
      wq[c] = a;
    }
    __syncthreads();
  }
}

// columns the row-per-wave kernels take (16-bit types); 0: the kernels above.  The backward holds three raw rows: up to 1024 columns.
// One launch of the row-per-wave backward.  Waves per block / rows in flight ahead of the one being worked on, by what fits the registers of
// the instantiation (measured, tools/join_bench.py): 8 waves (two per SIMD, 256 registers each) and two rows ahead; one row ahead where three
// raw rows do not fit (768 columns with both LayerNorms, 1024 columns); 1024 columns with both LayerNorms: 4 waves (512 registers), two ahead.
template <typename T, typename M, bool A, bool B, bool D, int WPB, int DIST, typename... Args>
static void join_bwd_row_go(int nblk, hipStream_t st, Args... args) {
  hipLaunchKernelGGL((join_bwd_row_kernel<T, M, WPB, DIST, A, B, D>), dim3(nblk), dim3(WPB * 64), 0, st, args...);
}
template <typename T, typename M, bool A, bool B, bool D, typename... Args>
static void join_bwd_row_launch(int variant, int nblk, hipStream_t st, Args... args) {
  constexpr bool AB = A && B;
  constexpr int WPB = (M::EPL == 16 && AB) ? 4 : 8;
  constexpr int DIST = M::EPL <= 8 ? 2 : (M::EPL == 12 ? (AB ? 1 : 2) : (AB ? 2 : 1));
#ifdef OFA_DEBUG_SWITCHES
  if (variant == 1) return join_bwd_row_go<T, M, A, B, D, 8, 1>(nblk, st, args...);
  if (variant == 2) return join_bwd_row_go<T, M, A, B, D, 4, 2>(nblk, st, args...);
  if (variant == 3) return join_bwd_row_go<T, M, A, B, D, 8, 2>(nblk, st, args...);
#endif
  join_bwd_row_go<T, M, A, B, D, WPB, DIST>(nblk, st, args...);
}
template <typename T, typename M, typename... Args>
static void join_bwd_row_flags(bool a, bool b, bool d, int variant, int nblk, hipStream_t st, Args... args) {
  switch ((a ? 4 : 0) | (b ? 2 : 0) | (d ? 1 : 0)) {
    case 0: return join_bwd_row_launch<T, M, false, false, false>(variant, nblk, st, args...);
    case 1: return join_bwd_row_launch<T, M, false, false, true>(variant, nblk, st, args...);
    case 2: return join_bwd_row_launch<T, M, false, true, false>(variant, nblk, st, args...);
    case 3: return join_bwd_row_launch<T, M, false, true, true>(variant, nblk, st, args...);
    case 4: return join_bwd_row_launch<T, M, true, false, false>(variant, nblk, st, args...);
    case 5: return join_bwd_row_launch<T, M, true, false, true>(variant, nblk, st, args...);
    case 6: return join_bwd_row_launch<T, M, true, true, false>(variant, nblk, st, args...);
    default: return join_bwd_row_launch<T, M, true, true, true>(variant, nblk, st, args...);
  }
}
template <typename T>
static void join_bwd_row_cols(int k, bool a, bool b, bool d, int variant, int nblk, hipStream_t st, const void* dy, const void* dz,
                              const void* x, const void* y, const void* ga, const void* gb, const float* stats, const uint8_t* keep_bits,
                              void* dres, void* dx, float* ws, int64_t rows, JoinRng rg, int want_xsum) {
#define GO(NF, TAIL)                                                                                                                        \
  join_bwd_row_flags<T, RowMap<NF, TAIL>>(a, b, d, variant, nblk, st, (const T*)dy, (const T*)dz, (const T*)x, (const T*)y, (const T*)ga, \
                                          (const T*)gb, stats, keep_bits, (T*)dres, (T*)dx, ws, rows, rg, want_xsum)
  switch (k) {
    case 1: GO(0, true); break;
    case 2: GO(1, false); break;
    case 3: GO(1, true); break;
    default: GO(2, false); break;
  }
#undef GO
}

// debug builds: OFA_JOIN_BWD = 0 (the split-row kernel) / 1, 2, 3 (row per wave: 8 waves one row ahead, 4 waves two ahead, 8 waves two ahead)
// / anything else (row per wave, the shipped choice per instantiation); OFA_JOIN_FWD = 0 (split row) / 1 (row per wave)
static int join_bwd_variant() {
#ifdef OFA_DEBUG_SWITCHES
  if (const char* e = getenv("OFA_JOIN_BWD")) return atoi(e);
#endif
  // Shipped since round 6.  (Round 5 held it back: the replayed cfg-2b graph died with a GPU memory access fault whenever this was the
  // backward kernel.  The fault was not this kernel's: its keep-bit tensors moved the decoder self-attention's lse buffer to the end of an
  // allocator segment, where an unclamped 32-row read of the dK/dV attention kernel -- csrc/attention.hip stat_dma, present since round 2 --
  // crossed into unmapped memory: profiles/round6_graph_fault_root_cause.txt.)
  return 9;
}
static int join_fwd_variant() {
#ifdef OFA_DEBUG_SWITCHES
  if (const char* e = getenv("OFA_JOIN_FWD")) return atoi(e);
#endif
  return 1;
}

static int join_row_k(int cols, int dtype, bool backward) {
  if (dtype == OFA_F32 || cols % 256 != 0 || !(backward ? join_bwd_variant() : join_fwd_variant())) return 0;
  const int k = cols / 256;
  return k >= 1 && k <= (backward ? 4 : 6) ? k : 0;
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

// bytes of the keep-bit buffer the forward may leave for the backward (0: this shape recomputes the mask in the backward)
extern "C" int64_t ofa_join_keep_bytes(int64_t rows, int cols, int dtype) {
  const int k = join_row_k(cols, dtype, true);
  return k ? rows * (int64_t)((k + 1) / 2) * 64 : 0;
}

#define JOIN_ROW_DISPATCH(k, CALL)              \
  do {                                          \
    switch (k) {                                \
      case 1: CALL(0, true); break;             \
      case 2: CALL(1, false); break;            \
      case 3: CALL(1, true); break;             \
      case 4: CALL(2, false); break;            \
      case 5: CALL(2, true); break;             \
      default: CALL(3, false); break;           \
    }                                           \
  } while (0)

extern "C" int ofa_join_fwd(const void* x, const void* residual, const void* gamma_a, const void* beta_a, const void* gamma_b,
                            const void* beta_b, void* y, void* z, float* stats, uint8_t* keep_bits, int64_t rows, int cols, float eps,
                            float p, uint64_t seed, uint64_t offset, const int64_t* offset_base, int dtype, void* stream) {
  if (int rc = join_check(rows, cols, dtype, "join_fwd")) return rc;
  OFA_REQUIRE(x && y && stats && (!gamma_a == !beta_a) && (!gamma_b == !beta_b) && (!gamma_b || z) && p >= 0.f && p < 1.f,
              OFA_ERR_INVALID, "join_fwd: bad argument");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
#ifdef OFA_DEBUG_SWITCHES
  if (getenv("OFA_JOIN_TRACE")) {
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "join_fwd rows %lld cols %d a %d b %d res %d p %g keep %p x %p y %p z %p stats %p\n", (long long)rows, cols, gamma_a != nullptr, gamma_b != nullptr,
            residual != nullptr, p, (void*)keep_bits, x, y, z, (void*)stats);
  }
#endif
  const JoinRng rg{p, seed, offset, offset_base};
  dim3 grid(cdiv(rows, 4)), block(256);
  if (const int k = join_row_k(cols, dtype, false)) {
    uint8_t* kb = ofa_join_keep_bytes(rows, cols, dtype) ? keep_bits : nullptr;
#define JOIN_FWD_ROW(NF, TAIL)                                                                                                      \
  do {                                                                                                                              \
    if (dtype == OFA_BF16)                                                                                                          \
      hipLaunchKernelGGL((join_fwd_row_kernel<bf16_t, RowMap<NF, TAIL>>), grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)residual, \
                         (const bf16_t*)gamma_a, (const bf16_t*)beta_a, (const bf16_t*)gamma_b, (const bf16_t*)beta_b, (bf16_t*)y,   \
                         (bf16_t*)z, stats, kb, rows, eps, rg);                                                                     \
    else                                                                                                                            \
      hipLaunchKernelGGL((join_fwd_row_kernel<f16_t, RowMap<NF, TAIL>>), grid, block, 0, st, (const f16_t*)x, (const f16_t*)residual,  \
                         (const f16_t*)gamma_a, (const f16_t*)beta_a, (const f16_t*)gamma_b, (const f16_t*)beta_b, (f16_t*)y,        \
                         (f16_t*)z, stats, kb, rows, eps, rg);                                                                      \
  } while (0)
    JOIN_ROW_DISPATCH(k, JOIN_FWD_ROW);
#undef JOIN_FWD_ROW
    return check_launch("join_fwd");
  }
  const int n = dtype == OFA_F32 ? 4 : 8;
  const int nv = cdiv(cols, 64 * n);
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
  int rpb;
  if (join_row_k(cols, dtype, true)) {
    rpb = 8;
  } else {
    const int wpr = join_wpr(cols, dtype == OFA_F32 ? 4 : 8);
    rpb = join_wpb(wpr) / wpr;
  }
  int64_t nblk = (rows + rpb - 1) / rpb;
  return (int)(nblk < 1 ? 1 : (nblk > JOIN_BLOCKS ? JOIN_BLOCKS : nblk));
}

// dy / dz: gradients of y / z (either may be NULL = zero; dz must be NULL iff gamma_b is); dres: gradient of the residual
// input; dx: gradient of x; ws: fp32 [5][ofa_join_bwd_slots][cols] partial rows of dgamma_a, dbeta_a, dgamma_b, dbeta_b and,
// with want_dx_colsum, the column sums of dx (= the bias gradient of the Linear that produced x, so that Linear needs no
// separate column-sum pass); fold with ofa_fold_batched; quantities that do not apply are not written.
// keep_bits: what ofa_join_fwd left (ofa_join_keep_bytes() > 0 and p > 0), or NULL: the mask is regenerated.
extern "C" int ofa_join_bwd(const void* dy, const void* dz, const void* x, const void* y, const void* gamma_a, const void* gamma_b,
                            const float* stats, const uint8_t* keep_bits, void* dres, void* dx, float* ws, int64_t rows, int cols,
                            float p, uint64_t seed, uint64_t offset, const int64_t* offset_base, int want_dx_colsum, int dtype,
                            void* stream) {
  if (int rc = join_check(rows, cols, dtype, "join_bwd")) return rc;
  OFA_REQUIRE(stats && dx && ws && (!gamma_a || x) && (!gamma_b || (y && dz)) && (dy || dz), OFA_ERR_INVALID,
              "join_bwd: bad argument");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
#ifdef OFA_DEBUG_SWITCHES
  if (getenv("OFA_JOIN_TRACE")) {
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "join_bwd rows %lld cols %d a %d b %d dy %d dres %d p %g keep %p xsum %d slots %d\n", (long long)rows, cols, gamma_a != nullptr, gamma_b != nullptr,
            dy != nullptr, dres != nullptr, p, (const void*)keep_bits, want_dx_colsum, ofa_join_bwd_slots(rows, cols, dtype));
  }
#endif
  const JoinRng rg{p, seed, offset, offset_base};
  if (const int k = (p > 0.f && !keep_bits) ? 0 : join_row_k(cols, dtype, true)) {
    const int nblk = ofa_join_bwd_slots(rows, cols, dtype);
    if (dtype == OFA_BF16)
      join_bwd_row_cols<bf16_t>(k, gamma_a, gamma_b, p > 0.f, join_bwd_variant(), nblk, st, dy, gamma_b ? dz : nullptr, x, y, gamma_a,
                                gamma_b, stats, keep_bits, dres, dx, ws, rows, rg, want_dx_colsum);
    else
      join_bwd_row_cols<f16_t>(k, gamma_a, gamma_b, p > 0.f, join_bwd_variant(), nblk, st, dy, gamma_b ? dz : nullptr, x, y, gamma_a,
                               gamma_b, stats, keep_bits, dres, dx, ws, rows, rg, want_dx_colsum);
    return check_launch("join_bwd");
  }
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
