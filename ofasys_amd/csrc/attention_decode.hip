// Decode-time attention for gfx950: ONE query row per (batch, head) against a key/value cache --
// the incremental branch of the reference's MultiheadAttention (module/multihead_attention.py:188-353 with
// incremental_state: prev_key/prev_value concatenated with the current step, :241-279; softmax in fp32, :333; per-head
// c_attn scale, :342-345).
//
// HBM-bound: each (b, h) block streams its S x 64 keys and values exactly once (2 * S * 64 * sizeof(T) bytes); the
// scores live in LDS (S floats), so the probabilities can also be written out (need_weights) without a second pass over
// K.  A key is covered by 64/N lanes (N = elements per 16-byte load: 8 lanes for bf16, 16 for fp32), a 256-thread block
// handles 32 (bf16) / 16 (fp32) keys per iteration with fully coalesced 16-byte loads along the cache rows.
// The cache is addressed as k + b * batch_stride + s * ld + h * 64: rows of a [B, capacity, heads*64] buffer of which
// only the first S rows are valid -- appending a step never moves the cache.
#include "common.h"

namespace ofa {

struct DecodeArgs {
  const void* q; const void* k; const void* v; const void* bias; const uint8_t* kpm; const void* c_attn; int c_dt;
  void* out; void* probs;
  int B, heads, S;
  int64_t ldk, bsk, kpm_ld;
  float scale;
};

template <typename T> __device__ __forceinline__ float exp_t(float x);
template <> __device__ __forceinline__ float exp_t<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float exp_t<bf16_t>(float x) { return __expf(x); }
template <> __device__ __forceinline__ float exp_t<f16_t>(float x) { return __expf(x); }

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(DecodeArgs a) {
  constexpr int N = Vec<T>::N, HD = 64;
  constexpr int LPK = HD / N;          // lanes per key
  constexpr int KPI = 256 / LPK;       // keys per block iteration
  extern __shared__ float smem[];
  float* sc = smem;                    // [S] scores, then exp(score - max)
  float* red = smem + ((a.S + 3) & ~3);   // [KPI][HD] partial outputs; its first 8 floats double as the scalar exchange
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tid / LPK, sub = tid % LPK;
  const int bh = blockIdx.x, b = bh / a.heads, h = bh % a.heads;
  const T* kb = (const T*)a.k + (int64_t)b * a.bsk + h * HD + sub * N;
  const T* vb = (const T*)a.v + (int64_t)b * a.bsk + h * HD + sub * N;
  float qv[N];
  load_vec<T>((const T*)a.q + ((int64_t)b * a.heads + h) * HD + sub * N, qv);
  const T* brow = a.bias ? (const T*)a.bias + (int64_t)bh * a.S : nullptr;
  const uint8_t* mrow = a.kpm ? a.kpm + (int64_t)b * a.kpm_ld : nullptr;

  // ---- scores
  float mx = -INFINITY;
  for (int s0 = 0; s0 < a.S; s0 += KPI) {
    const int s = s0 + g;
    float d = 0.f;
    if (s < a.S) {
      float kv[N];
      load_vec<T>(kb + (int64_t)s * a.ldk, kv);
#pragma unroll
      for (int j = 0; j < N; ++j) d += qv[j] * kv[j];
    }
#pragma unroll
    for (int o = 1; o < LPK; o <<= 1) d += __shfl_xor(d, o, 64);
    if (s < a.S && sub == 0) {
      float x = d * a.scale;
      if (brow) x += ld1<T>(brow + s);
      if (mrow && mrow[s]) x = -INFINITY;
      sc[s] = x;
      mx = fmaxf(mx, x);
    }
  }
  mx = wave_max(mx);
  __syncthreads();                                   // sc complete; red is free
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  // ---- exp + sum (a fully masked row gives probabilities 0, like the fused kernels)
  float l = 0.f;
  for (int s = tid; s < a.S; s += 256) {
    const float e = mx == -INFINITY ? 0.f : exp_t<T>(sc[s] - mx);
    sc[s] = e;
    l += e;
  }
  l = wave_sum(l);
  if (lane == 0) red[4 + wave] = l;
  __syncthreads();
  l = red[4] + red[5] + red[6] + red[7];
  const float inv_l = l > 0.f ? 1.0f / l : 0.f;
  __syncthreads();
  // ---- out = sum_s p[s] * v[s]
  float acc[N];
#pragma unroll
  for (int j = 0; j < N; ++j) acc[j] = 0.f;
  for (int s = g; s < a.S; s += KPI) {
    float vv[N];
    load_vec<T>(vb + (int64_t)s * a.ldk, vv);
    const float p = sc[s];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] += p * vv[j];
  }
#pragma unroll
  for (int j = 0; j < N; ++j) red[g * HD + sub * N + j] = acc[j];
  __syncthreads();
  if (tid < HD) {
    float o = 0.f;
#pragma unroll 4
    for (int r = 0; r < KPI; ++r) o += red[r * HD + tid];
    float c = 1.0f;
    if (a.c_attn) c = a.c_dt == OFA_BF16 ? bf2f(((const bf16_t*)a.c_attn)[h]) : a.c_dt == OFA_F16 ? (float)((const f16_t*)a.c_attn)[h] : ((const float*)a.c_attn)[h];
    st1<T>((T*)a.out + ((int64_t)b * a.heads + h) * HD + tid, o * inv_l * c);
  }
  if (a.probs) {
    T* prow = (T*)a.probs + (int64_t)bh * a.S;
    for (int s = tid; s < a.S; s += 256) st1<T>(prow + s, sc[s] * inv_l);
  }
}

}  // namespace ofa
using namespace ofa;

extern "C" int ofa_attn_decode(const void* q, const void* k, const void* v, const void* bias, const uint8_t* kpm,
                               const void* c_attn, int c_attn_dtype, void* out, void* probs, int B, int heads, int head_dim,
                               int S, int64_t ldk, int64_t k_batch_stride, int64_t kpm_ld, float scale, int dtype,
                               void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "attn_decode: bad dtype %d", dtype);
  OFA_REQUIRE(OFA_DT_OK(c_attn_dtype), OFA_ERR_INVALID, "attn_decode: bad c_attn dtype %d", c_attn_dtype);
  OFA_REQUIRE(head_dim == 64, OFA_ERR_UNSUPPORTED, "attn_decode: head_dim=%d (only 64: every OFA size)", head_dim);
  OFA_REQUIRE(B > 0 && heads > 0 && S > 0, OFA_ERR_INVALID, "attn_decode: bad shape B=%d heads=%d S=%d", B, heads, S);
  OFA_REQUIRE(S <= 32768, OFA_ERR_UNSUPPORTED, "attn_decode: S=%d exceeds the LDS score buffer (32768)", S);
  OFA_REQUIRE(q && k && v && out, OFA_ERR_INVALID, "attn_decode: null pointer");
  const int n = dtype == OFA_F32 ? 4 : 8;
  OFA_REQUIRE(ldk % n == 0 && k_batch_stride % n == 0 && ldk >= (int64_t)heads * 64, OFA_ERR_INVALID,
              "attn_decode: cache strides must keep 16-byte alignment (ld=%lld, batch stride=%lld)", (long long)ldk,
              (long long)k_batch_stride);
  OFA_REQUIRE(!kpm || kpm_ld >= S, OFA_ERR_INVALID, "attn_decode: key padding mask row shorter than S");
  DecodeArgs a{};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.kpm = kpm; a.c_attn = c_attn; a.c_dt = c_attn_dtype;
  a.out = out; a.probs = probs; a.B = B; a.heads = heads; a.S = S; a.ldk = ldk; a.bsk = k_batch_stride; a.kpm_ld = kpm_ld;
  a.scale = scale;
  const int kpi = 256 / (64 / n);
  const size_t lds = ((size_t)((S + 3) & ~3) + (size_t)kpi * 64) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OFA_F32) {
    auto kern = attn_decode_kernel<float>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), lds, st, a);
  } else if (dtype == OFA_BF16) {
    auto kern = attn_decode_kernel<bf16_t>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), lds, st, a);
  }
  else {
    auto kern = attn_decode_kernel<f16_t>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), lds, st, a);
  }
  return check_launch("attn_decode");
}
