// Criterion + train-step glue for gfx950 (HBM-bound, fp32 math):
//   cross entropy   engine/criterion/cross_entropy.py:27-67 (log_softmax in fp32, NLL sum, ignore_index = pad)
//   sum of squares  module/utils.py:342-384 (total grad norm for clip_grad_norm_)
//   Adam            engine/optim/adam.py:144-218 on the fp32 master copy of engine/optim/fp16_optimizer.py:32-71,
//                   fused with the grad multiply (engine/trainer.py:857-860) and the model-dtype copy-back.
#include "common.h"

namespace ofa {

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mm = fmaxf(m, m2);
  if (mm == -INFINITY) { m = mm; s = 0.f; return; }
  s = s * expf(m - mm) + s2 * expf(m2 - mm);
  m = mm;
}

// one 256-thread block per row
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                     float* __restrict__ lse, float* __restrict__ row_loss, int64_t V,
                                                     int64_t ld, int64_t ignore_index) {
  constexpr int N = Vec<T>::N;
  __shared__ float sm[4], ss[4];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ld;
  float m = -INFINITY, s = 0.f;
  const int64_t nvec = V / N;
  for (int64_t v = threadIdx.x; v < nvec; v += 256) {
    float a[N];
    load_vec<T>(x + v * N, a);
    float lm = a[0];
#pragma unroll
    for (int j = 1; j < N; ++j) lm = fmaxf(lm, a[j]);
    const float mm = fmaxf(m, lm);
    float acc = s * expf(m - mm);
#pragma unroll
    for (int j = 0; j < N; ++j) acc += expf(a[j] - mm);
    m = mm;
    s = acc;
  }
  for (int64_t e = nvec * N + threadIdx.x; e < V; e += 256) online_merge(m, s, ld1<T>(x + e), 1.f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    online_merge(m, s, m2, s2);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[wave] = m; ss[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = sm[0], S = ss[0];
    for (int w = 1; w < 4; ++w) online_merge(M, S, sm[w], ss[w]);
    const float l = M + logf(S);
    lse[row] = l;
    const int64_t t = target[row];
    row_loss[row] = (t == ignore_index) ? 0.f : l - ld1<T>(x + t);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ lse, const float* __restrict__ gscale,
                                                     T* __restrict__ dlogits, int64_t V, int64_t ld, int64_t ignore_index) {
  constexpr int N = Vec<T>::N;
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ld;
  T* dx = dlogits + row * ld;
  const int64_t t = target[row];
  const bool ignored = t == ignore_index;
  const float g = gscale ? gscale[0] : 1.0f;
  const float l = lse[row];
  const int64_t nvec = ld / N;   // ld is a multiple of the vector width (launch precondition)
  for (int64_t v = threadIdx.x; v < nvec; v += 256) {
    float a[N], o[N];
    load_vec<T>(x + v * N, a);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int64_t c = v * N + j;
      float d = 0.f;
      if (!ignored && c < V) d = (expf(a[j] - l) - (c == t ? 1.f : 0.f)) * g;
      o[j] = d;
    }
    store_vec<T>(dx + v * N, o);
  }
}

// Forward AND gradient in one pass over the logits (16-bit types, V <= 65536): the [rows, V] logits of the vocabulary projection
// (158 MB at cfg-2: 1536 decoder rows x 51265) were read by ce_fwd, read again and written as a gradient by ce_bwd -- 53 + 65 us.  The
// criterion is the LAST node of the forward graph and its upstream gradient is a scalar the caller knows before the forward runs (the
// backward seed 1, or the loss scale), so one kernel can do both: a 1024-thread block holds its whole row in registers (NV x 8 elements
// per thread), reduces max / sum-exp (waves by shuffles, the 16 waves through LDS), and writes lse, the row's loss and
// dlogits = (softmax - onehot) * grad_scale from the same registers: one read, one write.  Same arithmetic per element as
// dlogits = (softmax - onehot) * grad_scale from the same registers: one read, one write.  Exponentials by v_exp_f32 (1 ulp), a 1024-wide
// reduction tree: lse and the probabilities agree with ce_fwd_kernel / ce_bwd_kernel to fp32 rounding, not bit for bit.
template <typename T, int NV>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8))) void ce_fwd_grad_kernel(
    const T* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ gscale, float* __restrict__ lse,
    float* __restrict__ row_loss, T* __restrict__ dlogits, int V, int ld, int64_t ignore_index) {
  // (two blocks per CU -- 64 registers -- so that one block's loads and stores overlap the other's arithmetic: with one, 141 us at cfg-2
  // against 118 for the two kernels it replaces.  32-bit indices and scalar-base + 32-bit-offset addressing keep it there.)
  constexpr int N = Vec<T>::N;
  static_assert(N == 8, "16-bit element types");
  // exponentials as ONE v_exp_f32 behind an FMA (2^(x log2 e - m log2 e), 1 ulp): the library expf the two-kernel route uses is ~12
  // instructions, and with two exponentials per logit (one for the sum, one for the probability) that arithmetic, not HBM, set the time
  constexpr float L2E = 1.4426950408889634f;
  auto ex2 = [](float v) { return __builtin_amdgcn_exp2f(v); };
  __shared__ float sm[16], ss[16];
  const int64_t row = blockIdx.x;
  const char* xb = reinterpret_cast<const char*>(logits + row * ld);       // (uniform: scalar registers)
  char* db = reinterpret_cast<char*>(dlogits + row * ld);
  const int nvec = ld / N;                               // ld is a multiple of the vector width (launch precondition)
  const int tid = threadIdx.x;
  uint4 raw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * 1024;
    raw[i] = v < nvec ? *reinterpret_cast<const uint4*>(xb + (uint32_t)v * 16u) : make_uint4(0, 0, 0, 0);
  }
  float m = -INFINITY, s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = (tid + i * 1024) * N;
    if (c0 >= V) continue;
    float a[N];
    unpack16<T>(raw[i], a);
    if (c0 + N <= V) {
      float lm = a[0];
#pragma unroll
      for (int j = 1; j < N; ++j) lm = fmaxf(lm, a[j]);
      const float mm = fmaxf(m, lm), mk = -mm * L2E;
      float acc = s * ex2(fmaf(m, L2E, mk));             // (m == -inf on the first vector: 2^-inf = 0)
#pragma unroll
      for (int j = 0; j < N; ++j) acc += ex2(fmaf(a[j], L2E, mk));
      m = mm;
      s = acc;
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (c0 + j < V) online_merge(m, s, a[j], 1.f);
    }
  }
  // max first, then ONE rescale per partial sum: the pairwise online merges of ce_fwd_kernel (two expf each, six shuffle levels and
  // fifteen serial merges of the wave results) were this block's critical path -- nothing else runs on the CU while it reduces
  const float mw = wave_max(m);
  const float sw = wave_sum(mw == -INFINITY ? 0.f : s * ex2((m - mw) * L2E));
  const int wave = tid >> 6;
  if ((tid & 63) == 0) { sm[wave] = mw; ss[wave] = sw; }
  __syncthreads();
  float M = sm[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) M = fmaxf(M, sm[w]);
  float S = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) S += ss[w] * ex2((sm[w] - M) * L2E);   // (every thread: the same 16 values in the same order)
  const float l = M + logf(S);
  const int64_t t64 = target[row];
  const bool ignored = t64 == ignore_index;
  const int t = (t64 >= 0 && t64 < V) ? (int)t64 : -1;
  if (tid == 0) {
    lse[row] = l;
    row_loss[row] = ignored ? 0.f : l - ld1<T>(reinterpret_cast<const T*>(xb) + t64);
  }
  const float g = ignored ? 0.f : (gscale ? gscale[0] : 1.0f);
  const float lk = -l * L2E;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * 1024;
    if (v >= nvec) continue;
    float a[N], o[N];
    unpack16<T>(raw[i], a);
    const int c0 = v * N;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float d = 0.f;
      if (!ignored && c0 + j < V) d = (ex2(fmaf(a[j], L2E, lk)) - (c0 + j == t ? 1.f : 0.f)) * g;
      o[j] = d;
    }
    uint4 w;
    w.x = pack2<T>(o[0], o[1]); w.y = pack2<T>(o[2], o[3]); w.z = pack2<T>(o[4], o[5]); w.w = pack2<T>(o[6], o[7]);
    *reinterpret_cast<uint4*>(db + (uint32_t)v * 16u) = w;
  }
}

// ---- label-smoothed cross entropy (engine/criterion/label_smoothed_cross_entropy.py:62-191), fused with the
// log-softmax: per non-ignored row
//     nll = lse - x[t];   smooth = sum_{v allowed} (lse - x[v]);   eps_i = eps / (Vc - 1 [+1e-6 with constraints])
//     loss = (1 - eps - eps_i) * nll + eps_i * smooth
// where the allowed set is the whole vocabulary, or [0,4) U [cstart, cend) (constraint_range; logits outside are -inf in
// the reference, :140-151) intersected with an optional per-row byte mask (sample["constraint_masks"]).
struct LsCfg { float eps; int64_t cstart, cend; const uint8_t* cmask; };
__device__ __forceinline__ bool ls_allowed(const LsCfg& c, int64_t row, int64_t v, int64_t V) {
  bool ok = c.cstart < 0 || v < 4 || (v >= c.cstart && v < c.cend);
  if (ok && c.cmask) ok = c.cmask[row * V + v] != 0;
  return ok;
}

template <typename T>
__global__ __launch_bounds__(256) void lsce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                       float* __restrict__ lse, float* __restrict__ row_loss,
                                                       float* __restrict__ row_nll, float* __restrict__ row_cnt, int64_t V,
                                                       int64_t ld, int64_t ignore_index, LsCfg cfg) {
  __shared__ float sm[4], ss[4], sx[4], sc[4];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ld;
  float m = -INFINITY, s = 0.f, sumx = 0.f, cnt = 0.f;
  for (int64_t v = threadIdx.x; v < V; v += 256) {
    if (!ls_allowed(cfg, row, v, V)) continue;
    const float a = ld1<T>(x + v);
    online_merge(m, s, a, 1.f);
    sumx += a;
    cnt += 1.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    online_merge(m, s, m2, s2);
  }
  sumx = wave_sum(sumx);
  cnt = wave_sum(cnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[wave] = m; ss[wave] = s; sx[wave] = sumx; sc[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = sm[0], S = ss[0];
    for (int w = 1; w < 4; ++w) online_merge(M, S, sm[w], ss[w]);
    const float l = M + logf(S);
    const float X = sx[0] + sx[1] + sx[2] + sx[3], C = sc[0] + sc[1] + sc[2] + sc[3];
    lse[row] = l;
    row_cnt[row] = C;
    const int64_t t = target[row];
    if (t == ignore_index) {
      row_loss[row] = 0.f;
      row_nll[row] = 0.f;
    } else {
      const float nll = l - ld1<T>(x + t);
      const float smooth = C * l - X;
      const float eps_i = cfg.eps / (cfg.cstart < 0 && !cfg.cmask ? (C - 1.f) : (C - 1.f + 1e-6f));
      row_nll[row] = nll;
      row_loss[row] = (1.f - cfg.eps - eps_i) * nll + eps_i * smooth;
    }
  }
}

// d loss_row / d x[v] = (1 - eps - eps_i)(p_v - [v == t]) + eps_i (C p_v - 1)   for allowed v, 0 elsewhere; times
// row_w[row] (drop_worst / ignored rows: 0) and the scalar grad_scale.
template <typename T>
__global__ __launch_bounds__(256) void lsce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                       const float* __restrict__ lse, const float* __restrict__ row_cnt,
                                                       const float* __restrict__ row_w, const float* __restrict__ gscale,
                                                       T* __restrict__ dlogits, int64_t V, int64_t ld, int64_t ignore_index,
                                                       LsCfg cfg) {
  const int64_t row = blockIdx.x;
  const T* x = logits + row * ld;
  T* dx = dlogits + row * ld;
  const int64_t t = target[row];
  float g = gscale ? gscale[0] : 1.0f;
  if (row_w) g *= row_w[row];
  if (t == ignore_index) g = 0.f;
  const float l = lse[row], C = row_cnt[row];
  const float eps_i = cfg.eps / (cfg.cstart < 0 && !cfg.cmask ? (C - 1.f) : (C - 1.f + 1e-6f));
  const float w_nll = 1.f - cfg.eps - eps_i;
  for (int64_t v = threadIdx.x; v < ld; v += 256) {
    float d = 0.f;
    if (g != 0.f && v < V && ls_allowed(cfg, row, v, V)) {
      const float p = expf(ld1<T>(x + v) - l);
      d = (w_nll * (p - (v == t ? 1.f : 0.f)) + eps_i * (C * p - 1.f)) * g;
    }
    st1<T>(dx + v, d);
  }
}

// get_normalized_probs (model/ofa.py:287-299): fp32 softmax / log-softmax of the logits, one block per row.
__device__ __forceinline__ float block_sum(float v, float* sw) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  return sw[0] + sw[1] + sw[2] + sw[3];
}
template <typename T>
__global__ __launch_bounds__(256) void probs_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, int64_t V, int64_t ld,
                                                        int log_probs) {
  __shared__ float sw[4];
  const T* xr = x + (int64_t)blockIdx.x * ld;
  float* yr = y + (int64_t)blockIdx.x * V;
  float m = -INFINITY;
  for (int64_t c = threadIdx.x; c < V; c += 256) m = fmaxf(m, ld1<T>(xr + c));
  m = wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sw[0], sw[1]), fmaxf(sw[2], sw[3]));
  float s = 0.f;
  for (int64_t c = threadIdx.x; c < V; c += 256) s += expf(ld1<T>(xr + c) - m);
  s = block_sum(s, sw);
  const float l = m + logf(s);
  for (int64_t c = threadIdx.x; c < V; c += 256) {
    const float t = ld1<T>(xr + c) - l;
    yr[c] = log_probs ? t : expf(t);
  }
}
// log: dx = dy - exp(y)*sum(dy);   prob: dx = y*(dy - sum(dy*y))
template <typename T>
__global__ __launch_bounds__(256) void probs_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        T* __restrict__ dx, int64_t V, int64_t ld, int log_probs) {
  __shared__ float sw[4];
  const float* dyr = dy + (int64_t)blockIdx.x * V;
  const float* yr = y + (int64_t)blockIdx.x * V;
  T* dxr = dx + (int64_t)blockIdx.x * ld;
  float s = 0.f;
  for (int64_t c = threadIdx.x; c < V; c += 256) s += log_probs ? dyr[c] : dyr[c] * yr[c];
  s = block_sum(s, sw);
  for (int64_t c = threadIdx.x; c < ld; c += 256) {
    float g = 0.f;
    if (c < V) g = log_probs ? dyr[c] - expf(yr[c]) * s : yr[c] * (dyr[c] - s);
    st1<T>(dxr + c, g);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* __restrict__ x, float* __restrict__ partial, int64_t n) {
  constexpr int N = Vec<T>::N;
  __shared__ float sw[4];
  float s = 0.f;
  const int64_t nvec = n / N;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float a[N];
    load_vec<T>(x + v * N, a);
#pragma unroll
    for (int j = 0; j < N; ++j) s += a[j] * a[j];
  }
  if (blockIdx.x == 0)
    for (int64_t e = nvec * N + threadIdx.x; e < n; e += 256) { const float a = ld1<T>(x + e); s += a * a; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nb) {
  __shared__ float sw[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += sw[0] + sw[1] + sw[2] + sw[3];
}

template <typename T>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                   const T* __restrict__ grad, T* __restrict__ model,
                                                   const float* __restrict__ coef, int64_t n, float lr, float beta1,
                                                   float beta2, float eps, float wd, float step_size, int dev_sched) {
  const float gmul = coef ? coef[0] : 1.0f;
  if (dev_sched) {                       // schedule state lives on the device: coef = [grad multiplier, step size, lr, skip]
    if (coef[3] != 0.f) return;          // non-finite gradient norm / empty batch: parameters and both moments stay untouched
    step_size = coef[1];
    lr = coef[2];
  }
  auto upd = [&](float g, float& p, float& mi, float& vi) {
    g *= gmul;
    mi = mi * beta1 + (1.f - beta1) * g;
    vi = vi * beta2 + (1.f - beta2) * g * g;
    if (wd != 0.f) p -= wd * lr * p;
    p -= step_size * mi / (sqrtf(vi) + eps);
  };
  // Two quads (4 parameters each) per thread and trip, every load of the trip issued before the first update.  The fp32 state and
  // the gradient are streamed with NONTEMPORAL loads / stores (each byte is touched once per step and the arrays are many times
  // the size of L2 / MALL): tools/experiments/adam_stream_bench.hip, 141.6 M parameters = 3.96 GB per pass: plain accesses on a
  // 4096-block grid 778 us (5.1 TB/s), nontemporal + 65536 blocks 620 us (6.4 TB/s).  The model-dtype copy is read by the next
  // forward: plain store.
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const int64_t nq = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x; q0 < nq; q0 += 2 * stride) {
    f4v p4[2], m4[2], v4[2], g4[2];
    u2v g2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t q = q0 + k * stride;
      if (q < nq) {
        const int64_t i = q << 2;
        p4[k] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(master + i));
        m4[k] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(m + i));
        v4[k] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(v + i));
        if constexpr (sizeof(T) == 4) g4[k] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(grad + i));
        else g2[k] = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(grad + i));
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t q = q0 + k * stride;
      if (q < nq) {
        const int64_t i = q << 2;
        float g[4];
        if constexpr (sizeof(T) == 4) {
          g[0] = g4[k].x; g[1] = g4[k].y; g[2] = g4[k].z; g[3] = g4[k].w;
        } else {
          // (the two words as scalars first: hipcc 7.2 folds __builtin_bit_cast(f16x2_t, v.y) of an ext-vector ELEMENT to element 0 --
          // the fp16 kernel updated elements 2, 3 of every quad with the gradients of elements 0, 1; tests/test_fp16_gpu.py)
          const uint32_t gx = g2[k][0], gy = g2[k][1];
          if constexpr (!__is_same(T, bf16_t)) {                                  // fp16
            const f16x2_t a = __builtin_bit_cast(f16x2_t, gx), b = __builtin_bit_cast(f16x2_t, gy);
            g[0] = (float)a[0]; g[1] = (float)a[1]; g[2] = (float)b[0]; g[3] = (float)b[1];
          } else {
            g[0] = __uint_as_float(gx << 16); g[1] = __uint_as_float(gx & 0xffff0000u);
            g[2] = __uint_as_float(gy << 16); g[3] = __uint_as_float(gy & 0xffff0000u);
          }
        }
        float px = p4[k].x, py = p4[k].y, pz = p4[k].z, pw = p4[k].w;
        float mx = m4[k].x, my = m4[k].y, mz = m4[k].z, mw = m4[k].w;
        float vx = v4[k].x, vy = v4[k].y, vz = v4[k].z, vw = v4[k].w;
        upd(g[0], px, mx, vx);
        upd(g[1], py, my, vy);
        upd(g[2], pz, mz, vz);
        upd(g[3], pw, mw, vw);
        const f4v mo = {mx, my, mz, mw}, vo = {vx, vy, vz, vw}, po = {px, py, pz, pw};
        __builtin_nontemporal_store(mo, reinterpret_cast<f4v*>(m + i));
        __builtin_nontemporal_store(vo, reinterpret_cast<f4v*>(v + i));
        __builtin_nontemporal_store(po, reinterpret_cast<f4v*>(master + i));
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f4v*>(model + i) = po;
        } else {
          u2v o;
          o.x = pack2<T>(px, py);
          o.y = pack2<T>(pz, pw);
          *reinterpret_cast<u2v*>(model + i) = o;
        }
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {       // tail
    const int64_t i = (nq << 2) + threadIdx.x;
    float p = master[i], mi = m[i], vi = v[i];
    upd(ld1<T>(grad + i), p, mi, vi);
    m[i] = mi;
    v[i] = vi;
    master[i] = p;
    st1<T>(model + i, p);
  }
}

// One thread: the scalar arithmetic between the gradient norm and the Adam update, kept on the device so that a captured
// train step replays it (trainer.py:857-884 clip + 1/sample_size, adam.py:205-207 bias corrections).  ~25 tiny torch
// kernels otherwise, ~4.5 us each inside a hipGraph.
// Guard (engine/trainer.py:866-876 raises FloatingPointError("gradients are Nan/Inf") and never reaches optimizer.step): when the
// gradient norm is not finite or the step saw no target token (sample_size <= 0), sched = [0, 0, lr, 1] -- ofa_adam_step(step = 0)
// then returns without touching master weights or moments -- the update counter does not advance, and sched[4] (a running count
// of skipped updates) is incremented for the host to poll.
__global__ void step_schedule_kernel(const float* __restrict__ gsq, const double* __restrict__ sample_size,
                                     double* __restrict__ step, const double* __restrict__ lr, float* __restrict__ sched,
                                     float* __restrict__ gnorm, float clip_norm, double beta1, double beta2) {
  if (threadIdx.x || blockIdx.x) return;
  const double n = sample_size[0];
  const float inv_n = n > 0.0 ? (float)(1.0 / n) : 0.f;
  const float gn = sqrtf(gsq[0]) * inv_n;
  gnorm[0] = n > 0.0 ? gn : __builtin_nanf("");
  if (!(n > 0.0) || !isfinite(gn)) {
    sched[0] = 0.f;
    sched[1] = 0.f;
    sched[2] = (float)lr[0];
    sched[3] = 1.f;
    sched[4] += 1.f;
    return;
  }
  sched[3] = 0.f;
  float coef = inv_n;
  if (clip_norm > 0.f) coef *= fminf(clip_norm / (gn + 1e-6f), 1.0f);
  const double t = step[0] + 1.0;
  step[0] = t;
  const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
  sched[0] = coef;
  sched[1] = (float)(lr[0] * sqrt(bc2) / bc1);
  sched[2] = (float)lr[0];
  gnorm[0] = gn;
}

// The scaled-loss variant of step_schedule_kernel: engine/optim/fp16_optimizer.py:170-204 + dynamic_loss_scaler.py:9-70 in one
// device thread.  `ls` (fp64[8]) = [loss_scale, iter, last_overflow_iter, last_rescale_iter, overflows_since_rescale, fatal,
// 0, 0]; the gradients in the arena are loss_scale * (sum of per-sample gradients).
//   multiply_factor = 1 / (loss_scale * sample_size)                        (zero_grad :228-229, trainer.py:857-860)
//   grad_norm       = multiply_factor * ||g||                               (:178)
//   if grad_norm > max_norm > 0: multiply_factor *= max_norm / grad_norm    (:181-182; no "+1e-6", no clamp: the fp16 form)
//   overflow (inf / nan norm): check_overflow (:44-70) -- last_overflow_iter = iter, ++overflows_since_rescale, and when
//       overflows / iters_since_rescale >= tolerance: loss_scale = max(loss_scale / factor, threshold), last_rescale_iter = iter;
//       loss_scale <= min_loss_scale restores the previous scale and sets `fatal` (the reference raises FloatingPointError);
//       ++iter; the update is SKIPPED (trainer.py catches OverflowError and zeroes the grads)
//   otherwise update() (:33-37): if (iter - last_overflow_iter) % scale_window == 0: loss_scale *= factor,
//       last_rescale_iter = iter; ++iter.
__global__ void step_schedule_scaled_kernel(const float* __restrict__ gsq, const double* __restrict__ sample_size,
                                            double* __restrict__ step, const double* __restrict__ lr, float* __restrict__ sched,
                                            float* __restrict__ gnorm, double* __restrict__ ls, float clip_norm, double beta1,
                                            double beta2, double scale_factor, double scale_window, double tolerance,
                                            double threshold, double min_loss_scale) {
  if (threadIdx.x || blockIdx.x) return;
  const double n = sample_size[0];
  const double scale = ls[0];
  double mf = n > 0.0 ? 1.0 / (scale * n) : 0.0;
  const float gn = (float)(mf * sqrt((double)gsq[0]));
  gnorm[0] = n > 0.0 ? gn : __builtin_nanf("");
  if (!(n > 0.0)) {                              // an EMPTY batch is not an overflow: skip the update, leave the scaler's state alone
    sched[0] = 0.f;                              // (the reference's scaler only reacts to an inf / nan norm, dynamic_loss_scaler.py:44-70)
    sched[1] = 0.f;
    sched[2] = (float)lr[0];
    sched[3] = 1.f;
    sched[4] += 1.f;
    return;
  }
  if (!isfinite(gn)) {                           // overflow: the loss scaler's check_overflow, the update is skipped
    const double it = ls[1];
    const double since = it - ls[3];
    ls[2] = it;
    ls[4] += 1.0;
    const double pct = ls[4] / since;             // (since == 0 -> inf / nan >= tolerance is true / false as in Python's float division
    if (since <= 0.0 || pct >= tolerance) {      //  ... which raises ZeroDivisionError there; treated as "decrease" here)
      double dec = scale / scale_factor;
      if (threshold > 0.0 && dec < threshold) dec = threshold;
      ls[0] = dec;
      ls[3] = it;
      ls[4] = 0.0;
    }
    if (ls[0] <= min_loss_scale) {                // FloatingPointError in the reference: raised BEFORE its `_iter += 1`
      ls[0] = scale;
      ls[5] = 1.0;
    } else {
      ls[1] = it + 1.0;
    }
    sched[0] = 0.f;
    sched[1] = 0.f;
    sched[2] = (float)lr[0];
    sched[3] = 1.f;
    sched[4] += 1.f;
    return;
  }
  sched[3] = 0.f;
  if (clip_norm > 0.f && gn > clip_norm) mf *= (double)clip_norm / (double)gn;
  const double t = step[0] + 1.0;
  step[0] = t;
  const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
  sched[0] = (float)mf;
  sched[1] = (float)(lr[0] * sqrt(bc2) / bc1);
  sched[2] = (float)lr[0];
  const double it = ls[1];
  if (fmod(it - ls[2], scale_window) == 0.0) {
    ls[0] = scale * scale_factor;
    ls[3] = it;
  }
  ls[1] = it + 1.0;
}
}  // namespace ofa
using namespace ofa;

extern "C" int ofa_cross_entropy_fwd(const void* logits, const int64_t* target, float* lse, float* row_loss, int64_t rows,
                                     int64_t V, int64_t ld, int64_t ignore_index, int dtype, void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "cross_entropy_fwd: bad dtype %d", dtype);
  OFA_REQUIRE(rows >= 0 && V > 0 && ld >= V && logits && target && lse && row_loss, OFA_ERR_INVALID, "cross_entropy_fwd: bad argument");
  OFA_REQUIRE(ld % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_INVALID, "cross_entropy_fwd: ld=%lld must be a multiple of the 16-byte vector width", (long long)ld);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((ce_fwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, target, lse, row_loss, V, ld, ignore_index);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((ce_fwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, target, lse, row_loss, V, ld, ignore_index);
  else
    hipLaunchKernelGGL((ce_fwd_kernel<f16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const f16_t*)logits, target, lse, row_loss, V, ld, ignore_index);
  return check_launch("cross_entropy_fwd");
}

// the whole row lives in the block's registers: NV vectors of 8 per thread, 1024 threads
template <typename T>
static bool ce_fwd_grad_launch(const T* logits, const int64_t* target, const float* gs, float* lse, float* row_loss, T* dlogits,
                               int64_t rows, int64_t V, int64_t ld, int64_t ignore_index, hipStream_t st) {
  const int64_t per_thread = (ld / 8 + 1023) / 1024;
#define CE_FG(NV) hipLaunchKernelGGL((ce_fwd_grad_kernel<T, NV>), dim3((unsigned)rows), dim3(1024), 0, st, logits, target, gs, lse, row_loss, dlogits, (int)V, (int)ld, ignore_index)
  if (per_thread <= 1) CE_FG(1);
  else if (per_thread <= 2) CE_FG(2);
  else if (per_thread <= 4) CE_FG(4);
  else if (per_thread <= 7) CE_FG(7);
  else if (per_thread <= 8) CE_FG(8);
  else return false;
#undef CE_FG
  return true;
}

extern "C" int ofa_cross_entropy_fwd_grad_ok(int64_t V, int64_t ld, int dtype) {
  return (dtype == OFA_BF16 || dtype == OFA_F16) && V > 0 && ld >= V && ld % 8 == 0 && ld <= 8 * 1024 * 8;
}

extern "C" int ofa_cross_entropy_fwd_grad(const void* logits, const int64_t* target, const float* grad_scale, float* lse,
                                          float* row_loss, void* dlogits, int64_t rows, int64_t V, int64_t ld, int64_t ignore_index,
                                          int dtype, void* stream) {
  OFA_REQUIRE(ofa_cross_entropy_fwd_grad_ok(V, ld, dtype), OFA_ERR_UNSUPPORTED,
              "cross_entropy_fwd_grad: 16-bit logits of at most 65536 (padded) columns, ld a multiple of 8 (V=%lld ld=%lld dtype=%d)", (long long)V, (long long)ld, dtype);
  OFA_REQUIRE(rows >= 0 && logits && target && lse && row_loss && dlogits, OFA_ERR_INVALID, "cross_entropy_fwd_grad: bad argument");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool ok = dtype == OFA_BF16 ? ce_fwd_grad_launch<bf16_t>((const bf16_t*)logits, target, grad_scale, lse, row_loss, (bf16_t*)dlogits, rows, V, ld, ignore_index, st)
                                    : ce_fwd_grad_launch<f16_t>((const f16_t*)logits, target, grad_scale, lse, row_loss, (f16_t*)dlogits, rows, V, ld, ignore_index, st);
  OFA_REQUIRE(ok, OFA_ERR_UNSUPPORTED, "cross_entropy_fwd_grad: row too long");
  return check_launch("cross_entropy_fwd_grad");
}

extern "C" int ofa_cross_entropy_bwd(const void* logits, const int64_t* target, const float* lse, const float* grad_scale,
                                     void* dlogits, int64_t rows, int64_t V, int64_t ld, int64_t ignore_index, int dtype,
                                     void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "cross_entropy_bwd: bad dtype %d", dtype);
  OFA_REQUIRE(rows >= 0 && V > 0 && ld >= V && logits && target && lse && dlogits, OFA_ERR_INVALID, "cross_entropy_bwd: bad argument");
  OFA_REQUIRE(ld % (dtype == OFA_F32 ? 4 : 8) == 0, OFA_ERR_INVALID, "cross_entropy_bwd: ld must be a multiple of the 16-byte vector width");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, target, lse, grad_scale, (float*)dlogits, V, ld, ignore_index);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((ce_bwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, target, lse, grad_scale, (bf16_t*)dlogits, V, ld, ignore_index);
  else
    hipLaunchKernelGGL((ce_bwd_kernel<f16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const f16_t*)logits, target, lse, grad_scale, (f16_t*)dlogits, V, ld, ignore_index);
  return check_launch("cross_entropy_bwd");
}

extern "C" int ofa_ls_cross_entropy_fwd(const void* logits, const int64_t* target, float* lse, float* row_loss, float* row_nll,
                                        float* row_cnt, int64_t rows, int64_t V, int64_t ld, int64_t ignore_index, float eps,
                                        int64_t cstart, int64_t cend, const uint8_t* cmask, int dtype, void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "ls_cross_entropy_fwd: bad dtype %d", dtype);
  OFA_REQUIRE(rows >= 0 && V > 1 && ld >= V && logits && target && lse && row_loss && row_nll && row_cnt, OFA_ERR_INVALID,
              "ls_cross_entropy_fwd: bad argument");
  OFA_REQUIRE(eps >= 0.f && eps < 1.f && (cstart < 0 || (cstart >= 4 && cend > cstart && cend <= V)), OFA_ERR_INVALID,
              "ls_cross_entropy_fwd: bad eps / constraint range [%lld, %lld)", (long long)cstart, (long long)cend);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const LsCfg cfg{eps, cstart, cend, cmask};
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((lsce_fwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, target, lse, row_loss, row_nll, row_cnt, V, ld, ignore_index, cfg);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((lsce_fwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, target, lse, row_loss, row_nll, row_cnt, V, ld, ignore_index, cfg);
  else
    hipLaunchKernelGGL((lsce_fwd_kernel<f16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const f16_t*)logits, target, lse, row_loss, row_nll, row_cnt, V, ld, ignore_index, cfg);
  return check_launch("ls_cross_entropy_fwd");
}

extern "C" int ofa_ls_cross_entropy_bwd(const void* logits, const int64_t* target, const float* lse, const float* row_cnt,
                                        const float* row_w, const float* grad_scale, void* dlogits, int64_t rows, int64_t V,
                                        int64_t ld, int64_t ignore_index, float eps, int64_t cstart, int64_t cend,
                                        const uint8_t* cmask, int dtype, void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "ls_cross_entropy_bwd: bad dtype %d", dtype);
  OFA_REQUIRE(rows >= 0 && V > 1 && ld >= V && logits && target && lse && row_cnt && dlogits, OFA_ERR_INVALID,
              "ls_cross_entropy_bwd: bad argument");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const LsCfg cfg{eps, cstart, cend, cmask};
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((lsce_bwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, target, lse, row_cnt, row_w, grad_scale, (float*)dlogits, V, ld, ignore_index, cfg);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((lsce_bwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, target, lse, row_cnt, row_w, grad_scale, (bf16_t*)dlogits, V, ld, ignore_index, cfg);
  else
    hipLaunchKernelGGL((lsce_bwd_kernel<f16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const f16_t*)logits, target, lse, row_cnt, row_w, grad_scale, (f16_t*)dlogits, V, ld, ignore_index, cfg);
  return check_launch("ls_cross_entropy_bwd");
}

extern "C" int ofa_probs_fwd(const void* logits, float* out, int64_t rows, int64_t V, int64_t ld, int log_probs, int dtype,
                             void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "probs_fwd: bad dtype %d", dtype);
  OFA_REQUIRE(rows >= 0 && V > 0 && ld >= V && logits && out, OFA_ERR_INVALID, "probs_fwd: bad argument");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32) hipLaunchKernelGGL((probs_fwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, out, V, ld, log_probs);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((probs_fwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, out, V, ld, log_probs);
  else hipLaunchKernelGGL((probs_fwd_kernel<f16_t>), dim3((unsigned)rows), dim3(256), 0, st, (const f16_t*)logits, out, V, ld, log_probs);
  return check_launch("probs_fwd");
}

extern "C" int ofa_probs_bwd(const float* dy, const float* y, void* dlogits, int64_t rows, int64_t V, int64_t ld,
                             int log_probs, int dtype, void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "probs_bwd: bad dtype %d", dtype);
  OFA_REQUIRE(rows >= 0 && V > 0 && ld >= V && dy && y && dlogits, OFA_ERR_INVALID, "probs_bwd: bad argument");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32) hipLaunchKernelGGL((probs_bwd_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, dy, y, (float*)dlogits, V, ld, log_probs);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((probs_bwd_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, dy, y, (bf16_t*)dlogits, V, ld, log_probs);
  else hipLaunchKernelGGL((probs_bwd_kernel<f16_t>), dim3((unsigned)rows), dim3(256), 0, st, dy, y, (f16_t*)dlogits, V, ld, log_probs);
  return check_launch("probs_bwd");
}

extern "C" int ofa_sumsq_ws_floats(void) { return 1024; }

extern "C" int ofa_sumsq(const void* x, float* out, float* ws, int64_t n, int dtype, void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "sumsq: bad dtype %d", dtype);
  OFA_REQUIRE(n >= 0 && out && ws && (n == 0 || x), OFA_ERR_INVALID, "sumsq: bad argument");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int vecw = dtype == OFA_F32 ? 4 : 8;
  int64_t nbl = (n / vecw + 255) / 256;
  const int nb = (int)(nbl < 1 ? 1 : (nbl > 1024 ? 1024 : nbl));
  if (dtype == OFA_F32) hipLaunchKernelGGL((sumsq_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)x, ws, n);
  else if (dtype == OFA_BF16) hipLaunchKernelGGL((sumsq_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, ws, n);
  else hipLaunchKernelGGL((sumsq_kernel<f16_t>), dim3(nb), dim3(256), 0, st, (const f16_t*)x, ws, n);
  int rc = check_launch("sumsq");
  if (rc) return rc;
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, out, nb);
  return check_launch("sumsq_final");
}

extern "C" int ofa_adam_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, void* model_param,
                             const float* coef, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, int dtype, void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "adam_step: bad dtype %d", dtype);
  OFA_REQUIRE(n >= 0 && step >= 0 && master && exp_avg && exp_avg_sq && grad && model_param, OFA_ERR_INVALID, "adam_step: bad argument");
  OFA_REQUIRE(step >= 1 || coef, OFA_ERR_INVALID, "adam_step: step == 0 takes the step size and lr from coef[1], coef[2]");
  if (n == 0) return 0;
  // adam.py:205-207
  const int dev_sched = step == 0;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = dev_sched ? 0.f : (float)(lr * sqrt(bc2) / bc1);
  hipStream_t st = (hipStream_t)stream;
  OFA_REQUIRE(!(((uintptr_t)master | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) && !(((uintptr_t)grad | (uintptr_t)model_param) & 7),
              OFA_ERR_INVALID, "adam_step: arenas must be 16-byte (fp32 state) / 8-byte (grad, model copy) aligned");
  int64_t nbl = (n + 255) / 256;
  const int nb = (int)(nbl > 65536 ? 65536 : nbl);      // (small blocks of work: the tail of a 4096-block grid cost 15 %)
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((adam_kernel<float>), dim3(nb), dim3(256), 0, st, master, exp_avg, exp_avg_sq, (const float*)grad, (float*)model_param, coef, n, lr, beta1, beta2, eps, weight_decay, step_size, dev_sched);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((adam_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, master, exp_avg, exp_avg_sq, (const bf16_t*)grad, (bf16_t*)model_param, coef, n, lr, beta1, beta2, eps, weight_decay, step_size, dev_sched);
  else
    hipLaunchKernelGGL((adam_kernel<f16_t>), dim3(nb), dim3(256), 0, st, master, exp_avg, exp_avg_sq, (const f16_t*)grad, (f16_t*)model_param, coef, n, lr, beta1, beta2, eps, weight_decay, step_size, dev_sched);
  return check_launch("adam_step");
}

// One micro-batch's contribution to the step statistics [sample_size, loss_sum, ntokens] (engine/trainer.py:842-860: the criterion's
// logging outputs summed over the micro-batches of an update): the count of non-pad targets, the loss sum (a device scalar) and the
// count again, in ONE single-block launch -- was target.ne(pad).sum() plus three read-modify-writes of a fp64 element, nine launches.
__global__ __launch_bounds__(256) void step_stats_add_kernel(double* __restrict__ stats, const float* __restrict__ loss,
                                                             const int64_t* __restrict__ target, int64_t n, int64_t pad) {
  __shared__ int red[4];
  int cnt = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) cnt += target[i] != pad;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double c = (double)(red[0] + red[1] + red[2] + red[3]);
    stats[0] += c;
    stats[1] += (double)loss[0];
    stats[2] += c;
  }
}

extern "C" int ofa_step_stats_add(double* stats, const float* loss, const int64_t* target, int64_t n, int64_t pad, void* stream) {
  OFA_REQUIRE(stats && loss && (target || n == 0) && n >= 0, OFA_ERR_INVALID, "step_stats_add: bad argument");
  hipLaunchKernelGGL(step_stats_add_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, loss, target, n, pad);
  return check_launch("step_stats_add");
}

extern "C" int ofa_step_schedule_scaled(const float* gsq, const double* sample_size, double* step, const double* lr, float* sched,
                                        float* gnorm, double* loss_scaler, float clip_norm, double beta1, double beta2,
                                        double scale_factor, double scale_window, double tolerance, double threshold,
                                        double min_loss_scale, void* stream) {
  OFA_REQUIRE(gsq && sample_size && step && lr && sched && gnorm && loss_scaler, OFA_ERR_INVALID, "step_schedule_scaled: null pointer");
  OFA_REQUIRE(scale_factor > 1.0 && scale_window >= 1.0 && tolerance >= 0.0 && min_loss_scale >= 0.0, OFA_ERR_INVALID,
              "step_schedule_scaled: bad scaler configuration");
  hipLaunchKernelGGL(step_schedule_scaled_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gsq, sample_size, step, lr, sched, gnorm,
                     loss_scaler, clip_norm, beta1, beta2, scale_factor, scale_window, tolerance, threshold, min_loss_scale);
  return check_launch("step_schedule_scaled");
}

extern "C" int ofa_step_schedule(const float* gsq, const double* sample_size, double* step, const double* lr, float* sched,
                                 float* gnorm, float clip_norm, double beta1, double beta2, void* stream) {
  OFA_REQUIRE(gsq && sample_size && step && lr && sched && gnorm, OFA_ERR_INVALID, "step_schedule: null pointer");
  hipLaunchKernelGGL(step_schedule_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gsq, sample_size, step, lr, sched, gnorm,
                     clip_norm, beta1, beta2);
  return check_launch("step_schedule");
}
