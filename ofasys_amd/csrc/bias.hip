// Attention-bias assembly for gfx950 (HBM-bound): the per-layer, per-slot relative-position bias is added onto the
// diagonal block of the absolute-position bias (reference: adaptor/general.py:265-280,
//   self_attn_bias[:, :, s:e, s:e] += slot_bias   with slot_bias = values[T,T,A] expanded over batch, base.py:242-256).
// Forward adds in place on the caller's clone; backward reduces the block over the batch.
#include "common.h"

namespace ofa {

// bias[b][a][s+i][s+j] += values[i][j][a]        one thread per (b,a,i,j), j fastest (coalesced on bias)
template <typename T>
__global__ __launch_bounds__(256) void bias_block_add_kernel(T* __restrict__ bias, const T* __restrict__ values, int B, int A,
                                                             int Tt, int s, int n) {
  const int64_t total = (int64_t)B * A * n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % n);
    const int i = (int)((e / n) % n);
    const int a = (int)((e / ((int64_t)n * n)) % A);
    const int64_t b = e / ((int64_t)n * n * A);
    T* p = bias + (((b * A + a) * Tt) + s + i) * Tt + s + j;
    st1<T>(p, ld1<T>(p) + ld1<T>(values + ((int64_t)i * n + j) * A + a));
  }
}

// PER-SAMPLE block (a custom adaptor's own self_attn_bias, adaptor/base.py:183-189; adaptor/general.py:276):
//   ADD:  bias[b][a][s+i][s+j] += values[b][a][i][j]        !ADD (gradient):  values[b][a][i][j] = bias[b][a][s+i][s+j]
// one thread per (b,a,i,j), j fastest: coalesced on both sides
template <typename T, bool ADD>
__global__ __launch_bounds__(256) void bias_block_batch_kernel(T* __restrict__ bias, T* __restrict__ values, int B, int A, int Tt,
                                                               int s, int n) {
  const int64_t total = (int64_t)B * A * n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % n);
    const int i = (int)((e / n) % n);
    const int64_t ba = e / ((int64_t)n * n);
    T* p = bias + ((ba * Tt) + s + i) * Tt + s + j;
    if (ADD) st1<T>(p, ld1<T>(p) + ld1<T>(values + e));
    else values[e] = *p;
  }
}

// dvalues[i][j][a] = sum_b dbias[b][a][s+i][s+j]; deterministic (b ascending).  One block per (row i, 64-column chunk): the A head
// rows are read along j (coalesced on dbias), turned through LDS and written as the contiguous [64][A] run of dvalues -- the one
// thread per (i, j, a) form wrote 2-byte elements A * 2 bytes apart (55 us for the 1568^2 video block of cfg-4, now HBM speed)
template <typename T>
__global__ __launch_bounds__(256) void bias_block_grad_kernel(const T* __restrict__ dbias, T* __restrict__ dvalues, int B,
                                                              int A, int Tt, int s, int n) {
  __shared__ float tile[64 * 33];                            // [j][a], a < 32 (pitch 33)
  const int i = blockIdx.y, j0 = blockIdx.x * 64;
  const int jj = threadIdx.x & 63, al = threadIdx.x >> 6;
  for (int a0 = 0; a0 < A; a0 += 32) {                       // (A <= 32 in every model of the reference: one trip)
    const int an = A - a0 < 32 ? A - a0 : 32;
    for (int a = al; a < an; a += 4) {
      float acc = 0.f;
      if (j0 + jj < n)
        for (int b = 0; b < B; ++b) acc += ld1<T>(dbias + ((((int64_t)b * A + a0 + a) * Tt) + s + i) * Tt + s + j0 + jj);
      tile[jj * 33 + a] = acc;
    }
    __syncthreads();
    const int jn = n - j0 < 64 ? n - j0 : 64;
    for (int e = threadIdx.x; e < jn * an; e += 256) {
      const int j = e / an, a = e - j * an;
      st1<T>(dvalues + ((int64_t)i * n + j0 + j) * A + a0 + a, tile[j * 33 + a]);
    }
    __syncthreads();
  }
}

// ---- the batch-shared position bias of one layer: assembly + the two swizzled images (include/ofasys_amd.h, ofa_bias_build)
// One workgroup per 32 x 32 block (qt, kt), all heads: phase 1 gathers abs[h][q][k] (+ the slot's values[q - s][k - s][h] on a
// diagonal block) into an LDS tile per head -- one rounding to the 16-bit type, as torch's in-place add on the bias tensor -- and
// writes the row-major tensor; phase 2 writes each head's tile in the row image (lane (i, hi): row i, columns crowl(r, hi)) and the
// column image (lane (i, hi): column i, rows crowl(r, hi)), 8 bytes per thread and store.
struct BiasSlotsDev {
  const void* values[8];
  const void* values2[8];   // != NULL: an OUTER slot (video): value(i, j) = values[i / inner][j / inner] + values2[i % inner][j % inner]
  int start[8];
  int n[8];
  int inner[8];
  int count;
};
constexpr int BIAS_TP = 40;                                  // tile row pitch in elements: 80 bytes (16-byte aligned chunks)

template <typename T>
__global__ __launch_bounds__(256) void bias_build_kernel(const T* __restrict__ abs_bias, BiasSlotsDev slots, T* __restrict__ out,
                                                         T* __restrict__ swz_row, T* __restrict__ swz_col, int A, int Tb, int Sb,
                                                         int At, int h0) {   // heads h0 .. h0 + A - 1 of At (the slot values are [.., At])
  extern __shared__ __attribute__((aligned(16))) unsigned char bias_smem[];
  T* tile = reinterpret_cast<T*>(bias_smem);                // [A][32][BIAS_TP]
  const int kt = blockIdx.x, qt = blockIdx.y;
  const int nqt = gridDim.y, nkt = gridDim.x;
  const int q0 = qt * 32, k0 = kt * 32;
  const bool vec = (Sb & 7) == 0 && k0 + 32 <= Sb;           // whole 16-byte chunks of the row-major rows
  // phase 1a: the abs-pos block of every head -> LDS, 8 columns per thread and trip (zeros outside [Tb, Sb] or without abs_bias)
  for (int c = threadIdx.x; c < A * 128; c += 256) {
    const int kc = (c & 3) * 8, qr = (c >> 2) & 31, h = c >> 7;
    const int q = q0 + qr;
    uint4 w = make_uint4(0, 0, 0, 0);
    if (abs_bias && q < Tb) {
      const T* src = abs_bias + ((int64_t)h * Tb + q) * Sb + k0 + kc;
      if (vec) {
        w = *reinterpret_cast<const uint4*>(src);
      } else {
        T e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          T z;
          st1<T>(&z, 0.f);
          e[j] = k0 + kc + j < Sb ? src[j] : z;
        }
        __builtin_memcpy(&w, e, 16);
      }
    }
    *reinterpret_cast<uint4*>(tile + (h * 32 + qr) * BIAS_TP + kc) = w;
  }
  // phase 1b: a slot's values [n][n][A] on its diagonal block: a thread reads the A heads of one position (contiguous) and
  // adds them onto the tiles -- one rounding to the 16-bit type, as torch's in-place add on the bias tensor
  if (slots.count > 0) {
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const int kc = e & 31, qr = e >> 5;
      const int q = q0 + qr, k = k0 + kc;
      if (q >= Tb || k >= Sb) continue;
      for (int s = 0; s < slots.count; ++s) {
        const int i = q - slots.start[s], j = k - slots.start[s], n = slots.n[s];
        if (i < 0 || j < 0 || i >= n || j >= n) continue;
        if (slots.values2[s]) {              // frame-level table + patch-level table (video_image_sequence.py:187-204), summed in
          const int P = slots.inner[s], F = n / P;                      // the 16-bit type first, as the reference's broadcast add
          const T* vf = reinterpret_cast<const T*>(slots.values[s]) + ((int64_t)(i / P) * F + j / P) * At + h0;
          const T* vi = reinterpret_cast<const T*>(slots.values2[s]) + ((int64_t)(i % P) * P + j % P) * At + h0;
          for (int h = 0; h < A; ++h) {
            T v;
            st1<T>(&v, ld1<T>(vf + h) + ld1<T>(vi + h));
            T* t = tile + (h * 32 + qr) * BIAS_TP + kc;
            st1<T>(t, ld1<T>(t) + ld1<T>(&v));
          }
          continue;
        }
        const T* vp = reinterpret_cast<const T*>(slots.values[s]) + ((int64_t)i * n + j) * At + h0;
        for (int h = 0; h < A; ++h) {
          T* t = tile + (h * 32 + qr) * BIAS_TP + kc;
          st1<T>(t, ld1<T>(t) + ld1<T>(vp + h));
        }
      }
    }
  }
  __syncthreads();
  // phase 1c: the row-major tensor
  if (out) {
    for (int c = threadIdx.x; c < A * 128; c += 256) {
      const int kc = (c & 3) * 8, qr = (c >> 2) & 31, h = c >> 7;
      const int q = q0 + qr;
      if (q >= Tb) continue;
      const T* src = tile + (h * 32 + qr) * BIAS_TP + kc;
      T* dst = out + ((int64_t)h * Tb + q) * Sb + k0 + kc;
      if (vec) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
      } else {
        for (int j = 0; j < 8; ++j)
          if (k0 + kc + j < Sb) dst[j] = src[j];
      }
    }
  }
  // phase 2: the two images, 8 bytes per thread and store
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6;    // lane of the MFMA layout; g: register quad r = 4g .. 4g + 3
  const int i = l & 31, hi = l >> 5;
  const int c0 = 8 * g + 4 * hi;                            // crowl(4g + e, hi) = e + 8g + 4hi
  for (int h = 0; h < A; ++h) {
    const T* th = tile + h * 32 * BIAS_TP;
    T cv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) cv[e] = th[(c0 + e) * BIAS_TP + i];
    T* pr = swz_row + ((((int64_t)h * nqt + qt) * nkt + kt) * 64 + l) * 16 + 4 * g;
    T* pc = swz_col + ((((int64_t)h * nkt + kt) * nqt + qt) * 64 + l) * 16 + 4 * g;
    uint2 uc;
    __builtin_memcpy(&uc, cv, 8);
    *reinterpret_cast<uint2*>(pr) = *reinterpret_cast<const uint2*>(th + i * BIAS_TP + c0);
    *reinterpret_cast<uint2*>(pc) = uc;
  }
}

// Gradient of an OUTER slot from the bias gradient G [A, T, T] (one matrix: the batch sum):
//   d_vf[f][f'][a] = sum_{p, p'} G[a][s + f P + p][s + f' P + p'],   d_vi[p][p'][a] = sum_{f, f'} G[a][s + f P + p][s + f' P + p']
// (the reference materialises the [F P, F P, A] values and lets autograd reduce the broadcast twice).  Fixed summation order.
// V = 4: rows of the block are read as 8-byte (16-bit types) / 16-byte (fp32) pieces (start, T and P multiples of 4), V = 1: element-wise.
template <typename T, int V>
__device__ __forceinline__ void ld_piece(const T* p, float (&o)[V]) {
  if constexpr (V == 1) {
    o[0] = ld1<T>(p);
  } else if constexpr (sizeof(T) == 4) {
    const float4 u = *reinterpret_cast<const float4*>(p);
    o[0] = u.x; o[1] = u.y; o[2] = u.z; o[3] = u.w;
  } else {
    T e[4];
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    __builtin_memcpy(e, &u, 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = ld1<T>(e + j);
  }
}

// frames: one block per (f, f', head): 4 row lanes x 64 column lanes sweep the P x P block
template <typename T, int V>
__global__ __launch_bounds__(256) void bias_outer_grad_frames_kernel(const T* __restrict__ G, T* __restrict__ dvf, int A, int Tt, int s,
                                                                     int F, int P) {
  __shared__ float red[4];
  const int f = blockIdx.x / F, f2 = blockIdx.x - f * F, a = blockIdx.y;
  const T* base = G + ((int64_t)a * Tt + s + f * P) * Tt + s + f2 * P;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  float acc = 0.f;
  for (int c = cl * V; c < P; c += 64 * V)
    for (int p = rl; p < P; p += 4) {
      float o[V];
      ld_piece<T, V>(base + (int64_t)p * Tt + c, o);
#pragma unroll
      for (int j = 0; j < V; ++j) acc += o[j];
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) st1<T>(dvf + ((int64_t)f * F + f2) * A + a, red[0] + red[1] + red[2] + red[3]);
}

// patches: one block per (4 rows p, head): thread (row lane, column piece) sums its piece over the F x F frame pairs
template <typename T, int V>
__global__ __launch_bounds__(256) void bias_outer_grad_patches_kernel(const T* __restrict__ G, T* __restrict__ dvi, int A, int Tt, int s,
                                                                      int F, int P) {
  const int a = blockIdx.y;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + rl;
  if (p >= P) return;
  for (int c = cl * V; c < P; c += 64 * V) {
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int f = 0; f < F; ++f) {
      const T* row = G + ((int64_t)a * Tt + s + f * P + p) * Tt + s + c;
      for (int f2 = 0; f2 < F; ++f2) {
        float o[V];
        ld_piece<T, V>(row + f2 * P, o);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += o[j];
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) st1<T>(dvi + ((int64_t)p * P + c + j) * A + a, acc[j]);
  }
}

}  // namespace ofa
using namespace ofa;

extern "C" int ofa_bias_outer_grad(const void* dbias, void* d_frames, void* d_patches, int A, int T, int start, int F, int P, int dtype,
                                   void* stream) {
  OFA_REQUIRE(dtype == OFA_BF16 || dtype == OFA_F16 || dtype == OFA_F32, OFA_ERR_INVALID, "bias_outer_grad: bad dtype %d", dtype);
  OFA_REQUIRE(dbias && d_frames && d_patches && A > 0 && F > 0 && P > 0 && start >= 0 && start + F * P <= T, OFA_ERR_INVALID,
              "bias_outer_grad: bad argument (T=%d start=%d F=%d P=%d)", T, start, F, P);
  hipStream_t st = (hipStream_t)stream;
  const dim3 gf(F * F, A), gp((P + 3) / 4, A), blk(256);
  const bool v4 = (start & 3) == 0 && (T & 3) == 0 && (P & 3) == 0 && !((uintptr_t)dbias & 15);
#define OFA_OUTER(TT)                                                                                                              \
  do {                                                                                                                             \
    if (v4) {                                                                                                                      \
      hipLaunchKernelGGL((bias_outer_grad_frames_kernel<TT, 4>), gf, blk, 0, st, (const TT*)dbias, (TT*)d_frames, A, T, start, F, P);   \
      hipLaunchKernelGGL((bias_outer_grad_patches_kernel<TT, 4>), gp, blk, 0, st, (const TT*)dbias, (TT*)d_patches, A, T, start, F, P); \
    } else {                                                                                                                       \
      hipLaunchKernelGGL((bias_outer_grad_frames_kernel<TT, 1>), gf, blk, 0, st, (const TT*)dbias, (TT*)d_frames, A, T, start, F, P);   \
      hipLaunchKernelGGL((bias_outer_grad_patches_kernel<TT, 1>), gp, blk, 0, st, (const TT*)dbias, (TT*)d_patches, A, T, start, F, P); \
    }                                                                                                                              \
  } while (0)
  if (dtype == OFA_F32) OFA_OUTER(float);
  else if (dtype == OFA_BF16) OFA_OUTER(bf16_t);
  else OFA_OUTER(f16_t);
#undef OFA_OUTER
  return check_launch("bias_outer_grad");
}

extern "C" int64_t ofa_bias_swz_elems(int heads, int Tb, int Sb) {
  if (heads <= 0 || Tb <= 0 || Sb <= 0) return 0;
  return (int64_t)heads * ((Tb + 31) / 32) * ((Sb + 31) / 32) * 1024;
}

extern "C" int ofa_bias_build(const void* abs_bias, const ofa_bias_slots* slots, void* out, void* swz_row, void* swz_col, int heads,
                              int Tb, int Sb, int dtype, void* stream) {
  OFA_REQUIRE(dtype == OFA_BF16 || dtype == OFA_F16, OFA_ERR_UNSUPPORTED, "bias_build: 16-bit dtypes only (got %d)", dtype);
  OFA_REQUIRE(swz_row && swz_col && heads > 0 && Tb > 0 && Sb > 0, OFA_ERR_INVALID, "bias_build: bad argument");
  OFA_REQUIRE(!(((uintptr_t)swz_row | (uintptr_t)swz_col) & 15), OFA_ERR_INVALID, "bias_build: the swizzled images must be 16-byte aligned");
  BiasSlotsDev d{};
  if (slots) {
    OFA_REQUIRE(slots->count >= 0 && slots->count <= 8, OFA_ERR_INVALID, "bias_build: %d slots (at most 8)", slots->count);
    OFA_REQUIRE(slots->count == 0 || Tb == Sb, OFA_ERR_INVALID, "bias_build: slot blocks sit on the diagonal of a square bias (Tb=%d Sb=%d)", Tb, Sb);
    d.count = slots->count;
    for (int s = 0; s < d.count; ++s) {
      OFA_REQUIRE(slots->values[s] && slots->start[s] >= 0 && slots->n[s] > 0 && slots->start[s] + slots->n[s] <= Tb, OFA_ERR_INVALID,
                  "bias_build: slot %d (start %d, n %d) outside the %d positions", s, slots->start[s], slots->n[s], Tb);
      OFA_REQUIRE(!slots->values2[s] || (slots->inner[s] > 0 && slots->n[s] % slots->inner[s] == 0), OFA_ERR_INVALID,
                  "bias_build: outer slot %d: n=%d is not a multiple of the inner size %d", s, slots->n[s], slots->inner[s]);
      d.values[s] = slots->values[s];
      d.values2[s] = slots->values2[s];
      d.start[s] = slots->start[s];
      d.n[s] = slots->n[s];
      d.inner[s] = slots->inner[s];
    }
  }
  const dim3 grid((Sb + 31) / 32, (Tb + 31) / 32), block(256);
  const int64_t per_head = (int64_t)Tb * Sb, swz_head = (int64_t)grid.x * grid.y * 1024;
  for (int h0 = 0; h0 < heads; h0 += 24) {                       // (a workgroup holds the block of up to 24 heads in LDS)
    const int A = heads - h0 < 24 ? heads - h0 : 24;
    const size_t lds = (size_t)A * 32 * BIAS_TP * 2;
    if (dtype == OFA_BF16)
      hipLaunchKernelGGL((bias_build_kernel<bf16_t>), grid, block, lds, (hipStream_t)stream,
                         abs_bias ? (const bf16_t*)abs_bias + h0 * per_head : nullptr, d, out ? (bf16_t*)out + h0 * per_head : nullptr,
                         (bf16_t*)swz_row + h0 * swz_head, (bf16_t*)swz_col + h0 * swz_head, A, Tb, Sb, heads, h0);
    else
      hipLaunchKernelGGL((bias_build_kernel<f16_t>), grid, block, lds, (hipStream_t)stream,
                         abs_bias ? (const f16_t*)abs_bias + h0 * per_head : nullptr, d, out ? (f16_t*)out + h0 * per_head : nullptr,
                         (f16_t*)swz_row + h0 * swz_head, (f16_t*)swz_col + h0 * swz_head, A, Tb, Sb, heads, h0);
  }
  return check_launch("bias_build");
}

static inline int bias_grid(int64_t work) {
  int64_t g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int ofa_bias_block_add(void* bias, const void* values, int B, int A, int T, int start, int n, int dtype,
                                  void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "bias_block_add: bad dtype %d", dtype);
  OFA_REQUIRE(bias && values && B > 0 && A > 0 && n > 0 && start >= 0 && start + n <= T, OFA_ERR_INVALID,
              "bias_block_add: bad argument (T=%d start=%d n=%d)", T, start, n);
  const int64_t total = (int64_t)B * A * n * n;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((bias_block_add_kernel<float>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (float*)bias, (const float*)values, B, A, T, start, n);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((bias_block_add_kernel<bf16_t>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)bias, (const bf16_t*)values, B, A, T, start, n);
  else
    hipLaunchKernelGGL((bias_block_add_kernel<f16_t>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (f16_t*)bias, (const f16_t*)values, B, A, T, start, n);
  return check_launch("bias_block_add");
}

template <bool ADD>
static int bias_block_batch_launch(void* bias, void* values, int B, int A, int T, int start, int n, int dtype, void* stream,
                                   const char* what) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "%s: bad dtype %d", what, dtype);
  OFA_REQUIRE(bias && values && B > 0 && A > 0 && n > 0 && start >= 0 && start + n <= T, OFA_ERR_INVALID,
              "%s: bad argument (T=%d start=%d n=%d)", what, T, start, n);
  const int64_t total = (int64_t)B * A * n * n;
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((bias_block_batch_kernel<float, ADD>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (float*)bias, (float*)values, B, A, T, start, n);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((bias_block_batch_kernel<bf16_t, ADD>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)bias, (bf16_t*)values, B, A, T, start, n);
  else
    hipLaunchKernelGGL((bias_block_batch_kernel<f16_t, ADD>), dim3(bias_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (f16_t*)bias, (f16_t*)values, B, A, T, start, n);
  return check_launch(what);
}

extern "C" int ofa_bias_block_add_batch(void* bias, const void* values, int B, int A, int T, int start, int n, int dtype,
                                        void* stream) {
  return bias_block_batch_launch<true>(bias, const_cast<void*>(values), B, A, T, start, n, dtype, stream, "bias_block_add_batch");
}

extern "C" int ofa_bias_block_slice(const void* dbias, void* dvalues, int B, int A, int T, int start, int n, int dtype,
                                    void* stream) {
  return bias_block_batch_launch<false>(const_cast<void*>(dbias), dvalues, B, A, T, start, n, dtype, stream, "bias_block_slice");
}

extern "C" int ofa_bias_block_grad(const void* dbias, void* dvalues, int B, int A, int T, int start, int n, int dtype,
                                   void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "bias_block_grad: bad dtype %d", dtype);
  OFA_REQUIRE(dbias && dvalues && B > 0 && A > 0 && n > 0 && start >= 0 && start + n <= T, OFA_ERR_INVALID,
              "bias_block_grad: bad argument (T=%d start=%d n=%d)", T, start, n);
  const dim3 grid((n + 63) / 64, n);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((bias_block_grad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)dbias, (float*)dvalues, B, A, T, start, n);
  else if (dtype == OFA_BF16)
    hipLaunchKernelGGL((bias_block_grad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dbias, (bf16_t*)dvalues, B, A, T, start, n);
  else
    hipLaunchKernelGGL((bias_block_grad_kernel<f16_t>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const f16_t*)dbias, (f16_t*)dvalues, B, A, T, start, n);
  return check_launch("bias_block_grad");
}
