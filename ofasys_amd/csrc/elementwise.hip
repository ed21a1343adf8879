// HBM-bound pieces of the layer for gfx950: 16-byte vector loads, grid-stride loops, wave64 ballots.
//   GELU fwd/bwd            module/gelu.py:18-19 (exact erf, fp32 math)
//   dropout(x) + residual   transformer_layer.py:181-182, 203-206 (Philox4x32-10 counter RNG, mask never stored)
//   embedding gather        adaptor/text.py:124-125, module/layer.py:8-15
//   embedding grad          dense scatter-add, deterministic: one wave per vocabulary row scans the ids with ballots
//   add + row-vector + row mask   adaptor/base.py:168-173, model/transformer.py:110-112
//   im2col of p x p patches  adaptor/image_patch_embed.py:59-73
#include "common.h"

namespace ofa {

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ y,
                                                   int64_t nvec, int64_t n) {
  constexpr int N = Vec<T>::N;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float a[N], g[N], o[N];
    load_vec<T>(x + v * N, a);
    if (BWD) load_vec<T>(dy + v * N, g);
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = BWD ? g[j] * gelu_grad_f(a[j]) : gelu_f(a[j]);
    store_vec<T>(y + v * N, o);
  }
  // tail (n not a multiple of the vector width)
  if (blockIdx.x == 0) {
    for (int64_t e = nvec * N + threadIdx.x; e < n; e += 256) {
      const float a = ld1<T>(x + e);
      st1<T>(y + e, BWD ? ld1<T>(dy + e) * gelu_grad_f(a) : gelu_f(a));
    }
  }
}

// element e uses Philox counter (offset + e/8), halfword e%8 (philox_keep in common.h).
template <typename T, bool ADD>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                      int64_t n, float p, uint64_t seed, uint64_t offset,
                                                      const int64_t* __restrict__ offset_base) {
  if (offset_base) offset += (uint64_t)offset_base[0];   // device-resident stream position (hipGraph replays advance it)
  const Philox rng(seed);
  const float keep_scale = 1.0f / (1.0f - p);
  const uint32_t thresh = philox_thresh(p);
  const int64_t nq = (n + 7) / 8;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
    const uint4 r = rng(offset + (uint64_t)q);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t e = q * 8 + j;
      if (e < n) {
        float v = philox_keep(r, j, thresh) ? ld1<T>(x + e) * keep_scale : 0.f;
        if (ADD) v += ld1<T>(res + e);
        st1<T>(y + e, v);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_rowvec_mask_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                              const T* __restrict__ vec, const uint8_t* __restrict__ rowmask,
                                                              T* __restrict__ y, int64_t rows, int cols, int64_t b_period) {
  constexpr int N = Vec<T>::N;
  const int vpr = cols / N;
  const int64_t total = rows * vpr;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * N;
    float x[N], t[N];
    load_vec<T>(a + r * cols + c, x);
    if (b) {
      load_vec<T>(b + (b_period ? r % b_period : r) * cols + c, t);     // b_period rows of b serve every sample (shared positions)
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] += t[j];
    }
    if (vec) {
      load_vec<T>(vec + c, t);
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] += t[j];
    }
    if (rowmask && rowmask[r]) {
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = 0.f;
    }
    store_vec<T>(y + r * cols + c, x);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const T* __restrict__ w, const int64_t* __restrict__ ids,
                                                            T* __restrict__ out, int64_t n, int D, int64_t V,
                                                            uint8_t* __restrict__ is_pad, int64_t pad_id) {
  constexpr int N = Vec<T>::N;
  const int vpr = D / N;
  const int64_t total = n * vpr;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * N;
    int64_t id = ids[r];
    if (is_pad && c == 0) is_pad[r] = id == pad_id;         // the padding mask of the same ids (adaptor/text.py:108-111), one byte per row
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    *reinterpret_cast<typename Vec<T>::type*>(out + r * D + c) =
        *reinterpret_cast<const typename Vec<T>::type*>(w + id * D + c);
  }
}

// out[r] = index[r] >= 0 ? src[index[r]] : 0   -- the pack / unpack step of the ragged row layout (ofasys_amd/packing.py): the
// forward gathers the non-pad rows of the padded [B*T, D] adaptor output, the backward is the same kernel with the inverse index
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ index,
                                                          T* __restrict__ out, int64_t n, int D, int64_t src_rows) {
  constexpr int N = Vec<T>::N;
  const int vpr = D / N;
  const int64_t total = n * vpr;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * N;
    const int64_t id = index[r];
    typename Vec<T>::type val = {};
    if (id >= 0 && id < src_rows) val = *reinterpret_cast<const typename Vec<T>::type*>(src + id * D + c);
    *reinterpret_cast<typename Vec<T>::type*>(out + r * D + c) = val;
  }
}

// The same gather over the VIRTUAL concatenation of up to 8 slot outputs along the sequence (adaptor/general.py:245-282 `torch.cat(.., dim=1)`):
// index[r] names a position b * T + t of the concatenated [B, T, D]; slot k holds the positions start[k] <= t < start[k+1] of every sample as
// its own contiguous [B, n_k, D].  The packed rows are gathered straight from the slots' outputs: the concatenated tensor is never built.
struct GatherParts { const void* src[8]; int start[9]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_parts_kernel(GatherParts gp, const int64_t* __restrict__ index, T* __restrict__ out,
                                                                int64_t n, int D, int Ttot, int64_t batch) {
  constexpr int N = Vec<T>::N;
  const int vpr = D / N;
  const int64_t total = n * vpr;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * N;
    const int64_t id = index[r];
    typename Vec<T>::type val = {};
    if (id >= 0 && id < batch * Ttot) {
      const int64_t b = id / Ttot;
      const int t = (int)(id - b * Ttot);
      int k = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q)
        if (q < gp.n && t >= gp.start[q]) k = q;
      const int nk = gp.start[k + 1] - gp.start[k];
      val = *reinterpret_cast<const typename Vec<T>::type*>((const T*)gp.src[k] + ((b * nk + (t - gp.start[k])) * D + c));
    }
    *reinterpret_cast<typename Vec<T>::type*>(out + r * D + c) = val;
  }
}

// ... and its backward for ONE slot: out[b * nk + j] = inverse[b * T + start + j] >= 0 ? src[inverse[..]] : 0  (src: the packed-row gradient)
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_part_kernel(const T* __restrict__ src, const int64_t* __restrict__ inverse,
                                                                T* __restrict__ out, int64_t batch, int nk, int Ttot, int start, int D,
                                                                int64_t src_rows) {
  constexpr int N = Vec<T>::N;
  const int vpr = D / N;
  const int64_t total = batch * nk * vpr;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * N;
    const int64_t b = r / nk;
    const int j = (int)(r - b * nk);
    const int64_t id = inverse[b * Ttot + start + j];
    typename Vec<T>::type val = {};
    if (id >= 0 && id < src_rows) val = *reinterpret_cast<const typename Vec<T>::type*>(src + id * D + c);
    *reinterpret_cast<typename Vec<T>::type*>(out + r * D + c) = val;
  }
}

// narrow tables (rel-pos bias tables are [n_rel, heads], heads = 4..16): element-wise gather
template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_scalar_kernel(const T* __restrict__ w, const int64_t* __restrict__ ids,
                                                                   T* __restrict__ out, int64_t n, int D, int64_t V) {
  const int64_t total = n * D;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / D;
    const int c = (int)(e % D);
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    out[e] = w[id * D + c];
  }
}

__global__ __launch_bounds__(256) void embedding_mark_kernel(const int64_t* __restrict__ ids, uint8_t* __restrict__ present,
                                                             int64_t n, int64_t V) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t id = ids[i];
    if (id >= 0 && id < V) present[id] = 1;     // benign race: every writer stores the same value
  }
}

// dweight[v] += sum_{i : ids[i]==v} dout[i], deterministic, no atomics.  Wave (v, s) sweeps slice s of the id list 64 ids
// at a time, ballots the matches and adds the matching rows (16-byte loads) in increasing position order.  Big tables
// (vocabulary: few hits per row) use one slice and add straight into dweight; small tables (positions, token types:
// thousands of hits per row, which one wave would walk serially) are cut into S slices whose fp32 partial rows are
// folded in slice order by embedding_fold_kernel.  Rows that do not occur are skipped through a presence byte.
template <typename T, int NV>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                            T* __restrict__ dw, float* __restrict__ part, int64_t n, int D,
                                                            int64_t V, int64_t padding_idx,
                                                            const uint8_t* __restrict__ present, int S) {
  constexpr int N = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int sl = blockIdx.y;
  if (v >= V || v == padding_idx) return;
  if (present && !present[v]) return;      // most vocabulary rows do not occur in a batch: skip their scan
  float acc[NV][N];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) acc[i][j] = 0.f;
  const int64_t per = ((n + S - 1) / S + 63) / 64 * 64;
  const int64_t lo = sl * per, hi = lo + per < n ? lo + per : n;
  bool any = false;
  for (int64_t base = lo; base < hi; base += 64) {
    const int64_t idx = base + lane;
    const bool hit = idx < hi && ids[idx] == v;
    unsigned long long m = __ballot(hit);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      any = true;
      const T* row = dout + (base + src) * D;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * N;
        if (c < D) {
          float x[N];
          load_vec<T>(row + c, x);
#pragma unroll
          for (int j = 0; j < N; ++j) acc[i][j] += x[j];
        }
      }
    }
  }
  if (S == 1) {
    if (!any) return;
    T* wr = dw + v * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * N;
      if (c < D) {
        float o[N];
        load_vec<T>(wr + c, o);
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] += acc[i][j];
        store_vec<T>(wr + c, o);
      }
    }
  } else {                                 // (zeros too: the fold reads every slice of a present row)
    float* pr = part + ((int64_t)sl * V + v) * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * N;
      if (c < D)
#pragma unroll
        for (int j = 0; j < N; ++j) pr[c + j] = acc[i][j];
    }
  }
}

// narrow / unaligned tables (rel-pos bias tables are [n_rel, heads]): element-wise variant of the same sweep
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_scalar_kernel(const T* __restrict__ dout, const int64_t* __restrict__ ids,
                                                                   T* __restrict__ dw, float* __restrict__ part, int64_t n,
                                                                   int D, int64_t V, int64_t padding_idx,
                                                                   const uint8_t* __restrict__ present, int S) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int sl = blockIdx.y;
  if (v >= V || v == padding_idx) return;
  if (present && !present[v]) return;
  constexpr int MAXC = 32;                 // columns per lane held in registers: D <= 2048
  float acc[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) acc[j] = 0.f;
  const int64_t per = ((n + S - 1) / S + 63) / 64 * 64;
  const int64_t lo = sl * per, hi = lo + per < n ? lo + per : n;
  bool any = false;
  for (int64_t base = lo; base < hi; base += 64) {
    const int64_t idx = base + lane;
    const bool hit = idx < hi && ids[idx] == v;
    unsigned long long m = __ballot(hit);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      any = true;
      const T* row = dout + (base + src) * D;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        const int c = j * 64 + lane;
        if (c < D) acc[j] += ld1<T>(row + c);
      }
    }
  }
  if (S == 1 && !any) return;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = j * 64 + lane;
    if (c < D) {
      if (S == 1) st1<T>(dw + v * D + c, ld1<T>(dw + v * D + c) + acc[j]);
      else part[((int64_t)sl * V + v) * D + c] = acc[j];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embedding_fold_kernel(const float* __restrict__ part, T* __restrict__ dw, int D,
                                                             int64_t V, int64_t padding_idx,
                                                             const uint8_t* __restrict__ present, int S) {
  const int64_t v = blockIdx.x;
  if (v == padding_idx || (present && !present[v])) return;
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[((int64_t)s * V + v) * D + c];
    st1<T>(dw + v * D + c, ld1<T>(dw + v * D + c) + a);
  }
}

// col[(b*(lead + nph*npw) + lead + ph*npw + pw)][c*p*p + i*p + j] = img[b][c][ph*p+i][pw*p+j]; columns K..Kpad-1 are zero, and so are the
// `lead` rows in front of every sample's patches (the class-token position of adaptor/image_patch_embed.py:71-73: the projection GEMM then
// runs over the [B, 1 + N, D] rows the adaptor returns -- no concatenation -- and the weight-gradient GEMM over the same rows sees zeros there)
template <typename T>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const T* __restrict__ img, T* __restrict__ col, int B, int C, int H,
                                                           int W, int p, int Kpad, int lead) {
  const int nph = H / p, npw = W / p, K = C * p * p;
  const int64_t per = (int64_t)lead + (int64_t)nph * npw;
  const int64_t total = (int64_t)B * per * Kpad;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int k = (int)(e % Kpad);
    const int64_t r = e / Kpad;
    const int64_t b = r / per, rs = r % per - lead;
    float v = 0.f;
    if (k < K && rs >= 0) {
      const int j = k % p, i = (k / p) % p, c = k / (p * p);
      const int pw = (int)(rs % npw), ph = (int)(rs / npw);
      v = ld1<T>(img + ((b * C + c) * H + ph * p + i) * W + pw * p + j);
    }
    st1<T>(col + e, v);
  }
}

// column sums of a [rows, cols] matrix (bias gradients): 32 column-vectors x 8 row-lanes per block, grid.y row groups,
// fp32 partials [grid.y][cols] folded by a second pass (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                             int64_t rows, int cols, int64_t ld) {
  constexpr int N = Vec<T>::N;
  __shared__ float red[8][32][N];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cx) * N;
  float acc[N];
#pragma unroll
  for (int j = 0; j < N; ++j) acc[j] = 0.f;
  if (c < cols) {
    for (int64_t r = (int64_t)blockIdx.y * 8 + ry; r < rows; r += (int64_t)gridDim.y * 8) {
      float v[N];
      load_vec<T>(x + r * ld + c, v);
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) red[ry][cx][j] = acc[j];
  __syncthreads();
  if (ry == 0 && c < cols) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += red[r][cx][j];
      partial[(int64_t)blockIdx.y * cols + c + j] = t;
    }
  }
}
template <typename TO>
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, TO* __restrict__ out,
                                                            int cols, int groups, float alpha, int accumulate) {
  __shared__ float sm[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (c < cols)
    for (int g = ry; g < groups; g += 16) s += partial[(int64_t)g * cols + c];
  sm[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += sm[r][cx];
    t *= alpha;
    if (accumulate) t += ld1<TO>(out + c);
    st1<TO>(out + c, t);
  }
}

static inline int grid_for(int64_t work) {
  int64_t g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace ofa
using namespace ofa;

#define OFA_DT_CHECK(name) OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, name ": bad dtype %d", dtype)

extern "C" int ofa_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  OFA_DT_CHECK("gelu_fwd");
  OFA_REQUIRE(n >= 0 && (n == 0 || (x && y)), OFA_ERR_INVALID, "gelu_fwd: bad argument");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((gelu_kernel<float, false>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const float*)nullptr,
                       (const float*)x, (float*)y, n / 4, n);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((gelu_kernel<bf16_t, false>), dim3(grid_for(n / 8)), dim3(256), 0, st, (const bf16_t*)nullptr,
                       (const bf16_t*)x, (bf16_t*)y, n / 8, n);
  else
    hipLaunchKernelGGL((gelu_kernel<f16_t, false>), dim3(grid_for(n / 8)), dim3(256), 0, st, (const f16_t*)nullptr,
                       (const f16_t*)x, (f16_t*)y, n / 8, n);
  return check_launch("gelu_fwd");
}

extern "C" int ofa_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream) {
  OFA_DT_CHECK("gelu_bwd");
  OFA_REQUIRE(n >= 0 && (n == 0 || (dy && x && dx)), OFA_ERR_INVALID, "gelu_bwd: bad argument");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((gelu_kernel<float, true>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const float*)dy,
                       (const float*)x, (float*)dx, n / 4, n);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((gelu_kernel<bf16_t, true>), dim3(grid_for(n / 8)), dim3(256), 0, st, (const bf16_t*)dy,
                       (const bf16_t*)x, (bf16_t*)dx, n / 8, n);
  else
    hipLaunchKernelGGL((gelu_kernel<f16_t, true>), dim3(grid_for(n / 8)), dim3(256), 0, st, (const f16_t*)dy,
                       (const f16_t*)x, (f16_t*)dx, n / 8, n);
  return check_launch("gelu_bwd");
}

extern "C" int ofa_dropout_add_fwd(const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed,
                                   uint64_t offset, const int64_t* offset_base, int dtype, void* stream) {
  OFA_DT_CHECK("dropout_add_fwd");
  OFA_REQUIRE(n >= 0 && p >= 0.f && p < 1.f && (n == 0 || (x && y)), OFA_ERR_INVALID, "dropout_add_fwd: bad argument");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(grid_for((n + 7) / 8)), block(256);
  if (dtype == OFA_F32) {
    if (residual) hipLaunchKernelGGL((dropout_kernel<float, true>), grid, block, 0, st, (const float*)x, (const float*)residual, (float*)y, n, p, seed, offset, offset_base);
    else hipLaunchKernelGGL((dropout_kernel<float, false>), grid, block, 0, st, (const float*)x, (const float*)nullptr, (float*)y, n, p, seed, offset, offset_base);
  } else if (dtype == OFA_BF16) {
    if (residual) hipLaunchKernelGGL((dropout_kernel<bf16_t, true>), grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, n, p, seed, offset, offset_base);
    else hipLaunchKernelGGL((dropout_kernel<bf16_t, false>), grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)nullptr, (bf16_t*)y, n, p, seed, offset, offset_base);
  }
  else {
    if (residual) hipLaunchKernelGGL((dropout_kernel<f16_t, true>), grid, block, 0, st, (const f16_t*)x, (const f16_t*)residual, (f16_t*)y, n, p, seed, offset, offset_base);
    else hipLaunchKernelGGL((dropout_kernel<f16_t, false>), grid, block, 0, st, (const f16_t*)x, (const f16_t*)nullptr, (f16_t*)y, n, p, seed, offset, offset_base);
  }
  return check_launch("dropout_add_fwd");
}

extern "C" int ofa_dropout_bwd(const void* dy, void* dx, int64_t n, float p, uint64_t seed, uint64_t offset,
                               const int64_t* offset_base, int dtype, void* stream) {
  return ofa_dropout_add_fwd(dy, nullptr, dx, n, p, seed, offset, offset_base, dtype, stream);
}

extern "C" int ofa_add_rowvec_mask(const void* a, const void* b, const void* vec, const uint8_t* rowmask, void* y,
                                   int64_t rows, int cols, int64_t b_period, int dtype, void* stream) {
  OFA_DT_CHECK("add_rowvec_mask");
  OFA_REQUIRE(rows >= 0 && cols > 0 && a && y && b_period >= 0, OFA_ERR_INVALID, "add_rowvec_mask: bad argument");
  OFA_REQUIRE(cols % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "add_rowvec_mask: cols=%d not vectorizable", cols);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((add_rowvec_mask_kernel<float>), dim3(grid_for(rows * cols / 4)), dim3(256), 0, st, (const float*)a,
                       (const float*)b, (const float*)vec, rowmask, (float*)y, rows, cols, b_period);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((add_rowvec_mask_kernel<bf16_t>), dim3(grid_for(rows * cols / 8)), dim3(256), 0, st,
                       (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)vec, rowmask, (bf16_t*)y, rows, cols, b_period);
  else
    hipLaunchKernelGGL((add_rowvec_mask_kernel<f16_t>), dim3(grid_for(rows * cols / 8)), dim3(256), 0, st,
                       (const f16_t*)a, (const f16_t*)b, (const f16_t*)vec, rowmask, (f16_t*)y, rows, cols, b_period);
  return check_launch("add_rowvec_mask");
}

extern "C" int ofa_embedding_fwd(const void* weight, const int64_t* ids, void* out, int64_t n, int D, int64_t V, uint8_t* is_pad,
                                 int64_t pad_id, int dtype, void* stream) {
  OFA_DT_CHECK("embedding_fwd");
  OFA_REQUIRE(!is_pad || D % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "embedding_fwd: the pad mask rides with the vector kernel (D = %d)", D);
  OFA_REQUIRE(n >= 0 && D > 0 && V > 0 && weight && ids && out, OFA_ERR_INVALID, "embedding_fwd: bad argument");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (D % (dtype == OFA_F32 ? 4 : 8) != 0) {
    if (dtype == OFA_F32)
      hipLaunchKernelGGL((embedding_fwd_scalar_kernel<float>), dim3(grid_for(n * D)), dim3(256), 0, st, (const float*)weight,
                         ids, (float*)out, n, D, V);
    else if (dtype == OFA_BF16)
      hipLaunchKernelGGL((embedding_fwd_scalar_kernel<bf16_t>), dim3(grid_for(n * D)), dim3(256), 0, st,
                         (const bf16_t*)weight, ids, (bf16_t*)out, n, D, V);
    else
      hipLaunchKernelGGL((embedding_fwd_scalar_kernel<f16_t>), dim3(grid_for(n * D)), dim3(256), 0, st,
                         (const f16_t*)weight, ids, (f16_t*)out, n, D, V);
    return check_launch("embedding_fwd_scalar");
  }
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((embedding_fwd_kernel<float>), dim3(grid_for(n * D / 4)), dim3(256), 0, st, (const float*)weight, ids,
                       (float*)out, n, D, V, is_pad, pad_id);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((embedding_fwd_kernel<bf16_t>), dim3(grid_for(n * D / 8)), dim3(256), 0, st, (const bf16_t*)weight,
                       ids, (bf16_t*)out, n, D, V, is_pad, pad_id);
  else
    hipLaunchKernelGGL((embedding_fwd_kernel<f16_t>), dim3(grid_for(n * D / 8)), dim3(256), 0, st, (const f16_t*)weight,
                       ids, (f16_t*)out, n, D, V, is_pad, pad_id);
  return check_launch("embedding_fwd");
}

extern "C" int ofa_gather_rows(const void* src, const int64_t* index, void* out, int64_t n, int D, int64_t src_rows, int dtype,
                               void* stream) {
  OFA_DT_CHECK("gather_rows");
  OFA_REQUIRE(n >= 0 && D > 0 && src_rows > 0 && src && index && out, OFA_ERR_INVALID, "gather_rows: bad argument");
  OFA_REQUIRE(D % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_INVALID, "gather_rows: D=%d must be a multiple of the 16-byte vector width", D);
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(grid_for(n * D / 4)), dim3(256), 0, st, (const float*)src, index,
                       (float*)out, n, D, src_rows);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(grid_for(n * D / 8)), dim3(256), 0, st, (const bf16_t*)src, index,
                       (bf16_t*)out, n, D, src_rows);
  else
    hipLaunchKernelGGL((gather_rows_kernel<f16_t>), dim3(grid_for(n * D / 8)), dim3(256), 0, st, (const f16_t*)src, index,
                       (f16_t*)out, n, D, src_rows);
  return check_launch("gather_rows");
}

extern "C" int ofa_gather_rows_parts(const void* const* srcs, const int* lens, int nparts, const int64_t* index, void* out, int64_t n, int D,
                                     int64_t batch, int dtype, void* stream) {
  OFA_DT_CHECK("gather_rows_parts");
  OFA_REQUIRE(srcs && lens && index && out && nparts >= 1 && nparts <= 8 && n >= 0 && D > 0 && batch > 0, OFA_ERR_INVALID, "gather_rows_parts: bad argument");
  OFA_REQUIRE(D % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "gather_rows_parts: D=%d not vectorizable", D);
  if (n == 0) return 0;
  GatherParts gp;
  gp.n = nparts;
  gp.start[0] = 0;
  for (int k = 0; k < 8; ++k) {
    gp.src[k] = srcs[k < nparts ? k : 0];
    if (k < nparts) {
      OFA_REQUIRE(srcs[k] && lens[k] > 0 && !((uintptr_t)srcs[k] & 15), OFA_ERR_INVALID, "gather_rows_parts: part %d is NULL, empty or not 16-byte aligned", k);
      gp.start[k + 1] = gp.start[k] + lens[k];
    } else {
      gp.start[k + 1] = gp.start[k];
    }
  }
  const int Ttot = gp.start[nparts];
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((gather_rows_parts_kernel<float>), dim3(grid_for(n * D / 4)), dim3(256), 0, st, gp, index, (float*)out, n, D, Ttot, batch);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((gather_rows_parts_kernel<bf16_t>), dim3(grid_for(n * D / 8)), dim3(256), 0, st, gp, index, (bf16_t*)out, n, D, Ttot, batch);
  else
    hipLaunchKernelGGL((gather_rows_parts_kernel<f16_t>), dim3(grid_for(n * D / 8)), dim3(256), 0, st, gp, index, (f16_t*)out, n, D, Ttot, batch);
  return check_launch("gather_rows_parts");
}

extern "C" int ofa_scatter_rows_part(const void* src, const int64_t* inverse, void* out, int64_t batch, int nk, int Ttot, int start, int D,
                                     int64_t src_rows, int dtype, void* stream) {
  OFA_DT_CHECK("scatter_rows_part");
  OFA_REQUIRE(src && inverse && out && batch > 0 && nk > 0 && start >= 0 && start + nk <= Ttot && D > 0 && src_rows > 0, OFA_ERR_INVALID,
              "scatter_rows_part: bad argument");
  OFA_REQUIRE(D % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "scatter_rows_part: D=%d not vectorizable", D);
  hipStream_t st = (hipStream_t)stream;
  const int64_t work = batch * nk * D / (dtype == OFA_F32 ? 4 : 8);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((scatter_rows_part_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, (const float*)src, inverse, (float*)out, batch, nk, Ttot, start, D, src_rows);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((scatter_rows_part_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)src, inverse, (bf16_t*)out, batch, nk, Ttot, start, D, src_rows);
  else
    hipLaunchKernelGGL((scatter_rows_part_kernel<f16_t>), dim3(grid_for(work)), dim3(256), 0, st, (const f16_t*)src, inverse, (f16_t*)out, batch, nk, Ttot, start, D, src_rows);
  return check_launch("scatter_rows_part");
}

// Narrow-table scatter-add with a PRECOMPUTED segment plan: the ids of a rel-pos bias lookup (bucket[i][j], adaptor/text.py:101-104,
// image_resnet.py:116-128) are the same every step, a few hundred distinct values over 10^4..10^6 positions, so the positions are
// sorted by id once on the host side of the op (order / seg_off / seg_row, cached per lookup) and one wave sums one segment:
// 64 / CPL positions per trip x CPL columns (CPL = D rounded up to a power of two), fixed order, no atomics, no scan of the id list
// (embedding_bwd_scalar_kernel lets every table row sweep ALL ids: 47 us per [252 x 252] lookup, 0.85 ms per cfg-2b step).
template <typename T, int CPL>
__global__ __launch_bounds__(256) void segment_rowsum_kernel(const T* __restrict__ dout, const int* __restrict__ order,
                                                             const int* __restrict__ seg_off, const int* __restrict__ seg_row,
                                                             T* __restrict__ dw, int nseg, int D, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int sg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sg >= nseg) return;
  constexpr int G = 64 / CPL;
  const int g = lane / CPL, c = lane % CPL;
  const int lo = seg_off[sg], hi = seg_off[sg + 1];
  float acc = 0.f;
  int p = lo + g;
  for (; p + 3 * G < hi; p += 4 * G) {                    // four dependent index -> row gathers in flight per lane
    const int64_t r0 = order[p], r1 = order[p + G], r2 = order[p + 2 * G], r3 = order[p + 3 * G];
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (c < D) {
      v0 = ld1<T>(dout + r0 * D + c);
      v1 = ld1<T>(dout + r1 * D + c);
      v2 = ld1<T>(dout + r2 * D + c);
      v3 = ld1<T>(dout + r3 * D + c);
    }
    acc += (v0 + v1) + (v2 + v3);
  }
  for (; p < hi; p += G) {
    const int64_t row = order[p];
    if (c < D) acc += ld1<T>(dout + row * D + c);
  }
#pragma unroll
  for (int off = CPL; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
  if (g == 0 && c < D) {
    T* o = dw + (int64_t)seg_row[sg] * D + c;
    st1<T>(o, (accumulate ? ld1<T>(o) : 0.f) + acc);
  }
}

// slices for a table of V rows of D columns (<= 64 slices)
extern "C" int ofa_embedding_bwd_slices(int64_t V, int D) {
  if (V <= 0 || D <= 0) return 1;
  const int64_t by_rows = 8192 / V;                 // wide tables: <= 8192 partial rows
  const int64_t by_bytes = ((int64_t)1 << 21) / (V * D);   // narrow ones (rel-pos tables [~7000, heads]): <= 8 MB of partials
  int64_t s = by_rows > by_bytes ? by_rows : by_bytes;
  return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

template <typename T>
static void launch_segment_rowsum(const void* dout, const int32_t* order, const int32_t* seg_off, const int32_t* seg_row, void* dw, int nseg,
                                  int D, int accumulate, hipStream_t st) {
  const dim3 grid(cdiv(nseg, 4)), block(256);
  if (D <= 8) hipLaunchKernelGGL((segment_rowsum_kernel<T, 8>), grid, block, 0, st, (const T*)dout, order, seg_off, seg_row, (T*)dw, nseg, D, accumulate);
  else if (D <= 16) hipLaunchKernelGGL((segment_rowsum_kernel<T, 16>), grid, block, 0, st, (const T*)dout, order, seg_off, seg_row, (T*)dw, nseg, D, accumulate);
  else if (D <= 32) hipLaunchKernelGGL((segment_rowsum_kernel<T, 32>), grid, block, 0, st, (const T*)dout, order, seg_off, seg_row, (T*)dw, nseg, D, accumulate);
  else hipLaunchKernelGGL((segment_rowsum_kernel<T, 64>), grid, block, 0, st, (const T*)dout, order, seg_off, seg_row, (T*)dw, nseg, D, accumulate);
}

extern "C" int ofa_segment_rowsum(const void* dout, const int32_t* order, const int32_t* seg_off, const int32_t* seg_row, void* dweight,
                                  int nseg, int D, int accumulate, int dtype, void* stream) {
  OFA_DT_CHECK("segment_rowsum");
  OFA_REQUIRE(dout && order && seg_off && seg_row && dweight && nseg >= 0 && D > 0 && D <= 64, OFA_ERR_INVALID,
              "segment_rowsum: bad argument (D = %d, at most 64 columns)", D);
  if (nseg == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32) launch_segment_rowsum<float>(dout, order, seg_off, seg_row, dweight, nseg, D, accumulate, st);
  else if (dtype == OFA_BF16) launch_segment_rowsum<bf16_t>(dout, order, seg_off, seg_row, dweight, nseg, D, accumulate, st);
  else launch_segment_rowsum<f16_t>(dout, order, seg_off, seg_row, dweight, nseg, D, accumulate, st);
  return check_launch("segment_rowsum");
}

extern "C" int ofa_embedding_bwd(const void* dout, const int64_t* ids, void* dweight, int64_t n, int D, int64_t V,
                                 int64_t padding_idx, uint8_t* present_ws, float* slice_ws, int dtype, void* stream) {
  OFA_DT_CHECK("embedding_bwd");
  OFA_REQUIRE(n >= 0 && D > 0 && V > 0 && dout && ids && dweight, OFA_ERR_INVALID, "embedding_bwd: bad argument");
  const int vecn = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(D <= 2048, OFA_ERR_UNSUPPORTED, "embedding_bwd: D=%d exceeds 2048", D);
  const bool vec_ok = D % vecn == 0 && D <= 64 * vecn * 4 && (((uintptr_t)dout | (uintptr_t)dweight) & 15) == 0;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (present_ws) {                                   // optional V-byte scratch: mark the rows that occur
    (void)hipMemsetAsync(present_ws, 0, (size_t)V, st);
    hipLaunchKernelGGL(embedding_mark_kernel, dim3(grid_for(n)), dim3(256), 0, st, ids, present_ws, n, V);
  }
  const int S = slice_ws ? ofa_embedding_bwd_slices(V, D) : 1;
  const int nv = cdiv(D, 64 * vecn);
  dim3 grid((unsigned)((V + 3) / 4), S), block(256);
#define EMB_LAUNCH(T, NV)                                                                                               \
  hipLaunchKernelGGL((embedding_bwd_kernel<T, NV>), grid, block, 0, st, (const T*)dout, ids, (T*)dweight, slice_ws, n, D, V, \
                     padding_idx, (const uint8_t*)present_ws, S)
#define EMB_CASE(T)                  \
  do {                               \
    if (nv <= 1) EMB_LAUNCH(T, 1);   \
    else if (nv <= 2) EMB_LAUNCH(T, 2); \
    else EMB_LAUNCH(T, 4);           \
  } while (0)
  if (!vec_ok) {
    if (dtype == OFA_F32)
      hipLaunchKernelGGL((embedding_bwd_scalar_kernel<float>), grid, block, 0, st, (const float*)dout, ids, (float*)dweight,
                         slice_ws, n, D, V, padding_idx, (const uint8_t*)present_ws, S);
    else if (dtype == OFA_BF16)
      hipLaunchKernelGGL((embedding_bwd_scalar_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dout, ids,
                         (bf16_t*)dweight, slice_ws, n, D, V, padding_idx, (const uint8_t*)present_ws, S);
    else
      hipLaunchKernelGGL((embedding_bwd_scalar_kernel<f16_t>), grid, block, 0, st, (const f16_t*)dout, ids,
                         (f16_t*)dweight, slice_ws, n, D, V, padding_idx, (const uint8_t*)present_ws, S);
  } else if (dtype == OFA_F32) EMB_CASE(float);
  else if (dtype == OFA_BF16) EMB_CASE(bf16_t);
         else EMB_CASE(f16_t);
#undef EMB_CASE
#undef EMB_LAUNCH
  int rc = check_launch("embedding_bwd");
  if (rc || S == 1) return rc;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((embedding_fold_kernel<float>), dim3((unsigned)V), dim3(256), 0, st, slice_ws, (float*)dweight, D, V,
                       padding_idx, (const uint8_t*)present_ws, S);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((embedding_fold_kernel<bf16_t>), dim3((unsigned)V), dim3(256), 0, st, slice_ws, (bf16_t*)dweight, D, V,
                       padding_idx, (const uint8_t*)present_ws, S);
  else
    hipLaunchKernelGGL((embedding_fold_kernel<f16_t>), dim3((unsigned)V), dim3(256), 0, st, slice_ws, (f16_t*)dweight, D, V,
                       padding_idx, (const uint8_t*)present_ws, S);
  return check_launch("embedding_fold");
}

extern "C" int ofa_im2col_patch(const void* img, void* col, int B, int C, int H, int W, int p, int Kpad, int lead, int dtype,
                                void* stream) {
  OFA_DT_CHECK("im2col_patch");
  OFA_REQUIRE(img && col && B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0 && Kpad >= C * p * p && lead >= 0, OFA_ERR_INVALID,
              "im2col_patch: bad argument (H=%d W=%d p=%d Kpad=%d)", H, W, p, Kpad);
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * (lead + (int64_t)(H / p) * (W / p)) * Kpad;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((im2col_patch_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)img, (float*)col,
                       B, C, H, W, p, Kpad, lead);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((im2col_patch_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)img,
                       (bf16_t*)col, B, C, H, W, p, Kpad, lead);
  else
    hipLaunchKernelGGL((im2col_patch_kernel<f16_t>), dim3(grid_for(total)), dim3(256), 0, st, (const f16_t*)img,
                       (f16_t*)col, B, C, H, W, p, Kpad, lead);
  return check_launch("im2col_patch");
}

extern "C" int ofa_colsum_ws_floats(int cols) { return 128 * cols; }

extern "C" int ofa_colsum_slots(int64_t rows) {
  int64_t groups = (rows + 63) / 64;
  return (int)(groups < 1 ? 1 : (groups > 128 ? 128 : groups));
}

extern "C" int ofa_colsum(const void* x, void* out, float* ws, int64_t rows, int cols, int64_t ld, float alpha,
                          int accumulate, int dtype, int out_dtype, void* stream) {
  OFA_DT_CHECK("colsum");
  OFA_REQUIRE(OFA_DT_OK(out_dtype), OFA_ERR_INVALID, "colsum: bad out dtype %d", out_dtype);
  OFA_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && x && (out || accumulate == OFA_DEFER_FOLD) && ws, OFA_ERR_INVALID,
              "colsum: bad argument");
  const int n = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(cols % n == 0 && ld % n == 0, OFA_ERR_UNSUPPORTED, "colsum: cols=%d / ld not vectorizable", cols);
  hipStream_t st = (hipStream_t)stream;
  const int groups = ofa_colsum_slots(rows);
  dim3 grid(cdiv(cols / n, 32), groups), block(256);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, block, 0, st, (const float*)x, ws, rows, cols, ld);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, ws, rows, cols, ld);
  else
    hipLaunchKernelGGL((colsum_partial_kernel<f16_t>), grid, block, 0, st, (const f16_t*)x, ws, rows, cols, ld);
  int rc = check_launch("colsum_partial");
  if (rc || accumulate == OFA_DEFER_FOLD) return rc;          // deferred: the caller folds ws (ofa_fold_batched)
  if (out_dtype == OFA_F32)
    hipLaunchKernelGGL((colsum_final_kernel<float>), dim3(cdiv(cols, 64)), dim3(1024), 0, st, (const float*)ws, (float*)out,
                       cols, groups, alpha, accumulate);
  else if (out_dtype == OFA_BF16)
    hipLaunchKernelGGL((colsum_final_kernel<bf16_t>), dim3(cdiv(cols, 64)), dim3(1024), 0, st, (const float*)ws,
                       (bf16_t*)out, cols, groups, alpha, accumulate);
  else
    hipLaunchKernelGGL((colsum_final_kernel<f16_t>), dim3(cdiv(cols, 64)), dim3(1024), 0, st, (const float*)ws,
                       (f16_t*)out, cols, groups, alpha, accumulate);
  return check_launch("colsum_final");
}

// ---- y = a * b  (b same shape, or a row vector broadcast over rows)
namespace ofa {
template <typename T>
__global__ __launch_bounds__(256) void mul_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                  int64_t rows, int cols, int b_rowvec) {
  constexpr int N = Vec<T>::N;
  const int vpr = cols / N;
  const int64_t total = rows * vpr;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * N;
    float x[N], t[N];
    load_vec<T>(a + r * cols + c, x);
    load_vec<T>(b_rowvec ? b + c : b + r * cols + c, t);
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] *= t[j];
    store_vec<T>(y + r * cols + c, x);
  }
}
// y[r, :] = x[r, :] * scale[r / group]   (DropPath: one keep/scale factor per sample = per group of T batch-major rows)
template <typename T>
__global__ __launch_bounds__(256) void scale_row_groups_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                               T* __restrict__ y, int64_t rows, int cols, int64_t group) {
  constexpr int N = Vec<T>::N;
  const int vpr = cols / N;
  const int64_t total = rows * vpr;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * N;
    const float s = scale[r / group];
    float a[N];
    load_vec<T>(x + r * cols + c, a);
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] *= s;
    store_vec<T>(y + r * cols + c, a);
  }
}
// out[0] = sum(x[0..n))   single block, deterministic
__global__ __launch_bounds__(256) void reduce_sum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  __shared__ float sw[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sw[0] + sw[1] + sw[2] + sw[3];
}
// out[h] = sum_b sum_t x[(b*heads + h)*ld + t], t < T     one block per head
__global__ __launch_bounds__(256) void head_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int heads,
                                                       int T, int ld) {
  __shared__ float sw[4];
  const int h = blockIdx.x;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (int t = threadIdx.x; t < T; t += 256) s += x[((int64_t)b * heads + h) * ld + t];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[h] = sw[0] + sw[1] + sw[2] + sw[3];
}
// out[i] = sum_{k < n} in_k[i]   (fp32 accumulation, one rounding): the gradient of a tensor that n consumers read (ops.fan_out: the
// abs-position bias every layer of a stack assembles its attention bias from) in ONE launch instead of autograd's n - 1 pairwise adds
struct AddNArgs { const void* in[16]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void add_n_kernel(AddNArgs a, T* __restrict__ out, int64_t numel) {
  constexpr int N = Vec<T>::N;
  const int64_t nvec = numel / N;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    for (int k = 0; k < a.n; ++k) {
      float t[N];
      load_vec<T>((const T*)a.in[k] + v * N, t);
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += t[j];
    }
    store_vec<T>(out + v * N, acc);
  }
  if (blockIdx.x == 0) {                                   // the tail that is not a whole 16-byte vector
    for (int64_t i = nvec * N + threadIdx.x; i < numel; i += 256) {
      float t = 0.f;
      for (int k = 0; k < a.n; ++k) t += ld1<T>((const T*)a.in[k] + i);
      st1<T>(out + i, t);
    }
  }
}
}  // namespace ofa

extern "C" int ofa_add_n(const void* const* inputs, int n, void* out, int64_t numel, int dtype, void* stream) {
  OFA_DT_CHECK("add_n");
  OFA_REQUIRE(inputs && out && n >= 1 && n <= 16 && numel >= 0, OFA_ERR_INVALID, "add_n: 1..16 inputs (got %d)", n);
  if (numel == 0) return 0;
  AddNArgs a;
  for (int k = 0; k < 16; ++k) a.in[k] = inputs[k < n ? k : 0];
  for (int k = 0; k < n; ++k) OFA_REQUIRE(inputs[k] && !((uintptr_t)inputs[k] & 15), OFA_ERR_INVALID, "add_n: input %d is NULL or not 16-byte aligned", k);
  OFA_REQUIRE(!((uintptr_t)out & 15), OFA_ERR_INVALID, "add_n: out is not 16-byte aligned");
  a.n = n;
  hipStream_t st = (hipStream_t)stream;
  const int64_t work = numel / (dtype == OFA_F32 ? 4 : 8) + 1;
  if (dtype == OFA_F32) hipLaunchKernelGGL((add_n_kernel<float>), dim3(grid_for(work)), dim3(256), 0, st, a, (float*)out, numel);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((add_n_kernel<bf16_t>), dim3(grid_for(work)), dim3(256), 0, st, a, (bf16_t*)out, numel);
  else hipLaunchKernelGGL((add_n_kernel<f16_t>), dim3(grid_for(work)), dim3(256), 0, st, a, (f16_t*)out, numel);
  return check_launch("add_n");
}

// out[i] (+)= sum_b x[b * n + i]: the gradient of a tensor every sample of the batch read (batch-invariant position embeddings)
template <typename T>
__global__ __launch_bounds__(256) void batch_sum_kernel(const T* __restrict__ x, T* __restrict__ out, int batch, int64_t n, int accumulate) {
  constexpr int N = Vec<T>::N;
  const int64_t total = n / N;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    float acc[N], t[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    if (accumulate) load_vec<T>(out + v * N, acc);
    for (int b = 0; b < batch; ++b) {
      load_vec<T>(x + (int64_t)b * n + v * N, t);
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] += t[j];
    }
    store_vec<T>(out + v * N, acc);
  }
}

extern "C" int ofa_batch_sum(const void* x, void* out, int batch, int64_t n, int accumulate, int dtype, void* stream) {
  OFA_DT_CHECK("batch_sum");
  OFA_REQUIRE(x && out && batch >= 1 && n >= 0, OFA_ERR_INVALID, "batch_sum: bad argument");
  OFA_REQUIRE(n % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "batch_sum: n=%lld not vectorizable", (long long)n);
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32) hipLaunchKernelGGL((batch_sum_kernel<float>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const float*)x, (float*)out, batch, n, accumulate);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((batch_sum_kernel<bf16_t>), dim3(grid_for(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, batch, n, accumulate);
  else hipLaunchKernelGGL((batch_sum_kernel<f16_t>), dim3(grid_for(n / 8)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)out, batch, n, accumulate);
  return check_launch("batch_sum");
}

extern "C" int ofa_mul(const void* a, const void* b, void* y, int64_t rows, int cols, int b_rowvec, int dtype, void* stream) {
  OFA_DT_CHECK("mul");
  OFA_REQUIRE(rows >= 0 && cols > 0 && a && b && y, OFA_ERR_INVALID, "mul: bad argument");
  OFA_REQUIRE(cols % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "mul: cols=%d not vectorizable", cols);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((mul_kernel<float>), dim3(grid_for(rows * cols / 4)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)y, rows, cols, b_rowvec);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((mul_kernel<bf16_t>), dim3(grid_for(rows * cols / 8)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, rows, cols, b_rowvec);
  else
    hipLaunchKernelGGL((mul_kernel<f16_t>), dim3(grid_for(rows * cols / 8)), dim3(256), 0, st, (const f16_t*)a, (const f16_t*)b, (f16_t*)y, rows, cols, b_rowvec);
  return check_launch("mul");
}

extern "C" int ofa_scale_row_groups(const void* x, const float* scale, void* y, int64_t rows, int cols, int64_t group, int dtype,
                                    void* stream) {
  OFA_DT_CHECK("scale_row_groups");
  OFA_REQUIRE(rows >= 0 && cols > 0 && group > 0 && x && scale && y, OFA_ERR_INVALID, "scale_row_groups: bad argument");
  OFA_REQUIRE(cols % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_UNSUPPORTED, "scale_row_groups: cols=%d not vectorizable", cols);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((scale_row_groups_kernel<float>), dim3(grid_for(rows * cols / 4)), dim3(256), 0, st, (const float*)x, scale, (float*)y, rows, cols, group);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((scale_row_groups_kernel<bf16_t>), dim3(grid_for(rows * cols / 8)), dim3(256), 0, st, (const bf16_t*)x, scale, (bf16_t*)y, rows, cols, group);
  else
    hipLaunchKernelGGL((scale_row_groups_kernel<f16_t>), dim3(grid_for(rows * cols / 8)), dim3(256), 0, st, (const f16_t*)x, scale, (f16_t*)y, rows, cols, group);
  return check_launch("scale_row_groups");
}

extern "C" int ofa_reduce_sum_f32(const float* x, float* out, int64_t n, void* stream) {
  OFA_REQUIRE(n >= 0 && out && (n == 0 || x), OFA_ERR_INVALID, "reduce_sum_f32: bad argument");
  hipLaunchKernelGGL(reduce_sum_f32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, out, n);
  return check_launch("reduce_sum_f32");
}

extern "C" int ofa_head_sum_f32(const float* x, float* out, int B, int heads, int T, int ld, void* stream) {
  OFA_REQUIRE(x && out && B > 0 && heads > 0 && T > 0 && ld >= T, OFA_ERR_INVALID, "head_sum_f32: bad argument");
  hipLaunchKernelGGL(head_sum_kernel, dim3(heads), dim3(256), 0, (hipStream_t)stream, x, out, B, heads, T, ld);
  return check_launch("head_sum_f32");
}
