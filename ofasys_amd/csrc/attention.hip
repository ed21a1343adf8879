// Fused multi-head attention for gfx950 (bf16 in/out, fp32 softmax, head_dim 64), forward + backward.
//
// Reference arithmetic: the slow path of MultiheadAttention.forward, module/multihead_attention.py:218-346
// (q*scaling, bmm(q,k^T), += attn_bias, causal -inf mask, key-padding -inf, softmax(dtype=fp32), bmm(p,v),
// per-head c_attn scale) and, with bias == NULL and scale = head_dim^-0.5, the F.multi_head_attention_forward
// fast path (:155-186).  The [B*A,T,S] score/probability tensors are never written to HBM.
//
// CDNA4 mapping.  One 64-lane wavefront owns 32 query rows (or 32 key rows in the dK/dV kernel) and walks the other
// sequence in blocks of 32.  Every product is a v_mfma_f32_32x32x16_bf16 whose operands are fetched with 16-byte
// (or paired 8-byte) loads that are contiguous in memory -- no LDS transposes:
//     S^T = K Q^T        (contract d)    A <- K rows,          B <- Q rows
//     O^T = V^T P^T      (contract key)  A <- V^T rows (key-contiguous copy),  B <- P^T straight from the S^T
//                                         accumulator registers (C-layout rows = keys = MFMA k-slots)
//   backward:
//     dP^T = V dO^T, dQ^T = K^T dS^T     (dQ kernel, lanes <-> queries)
//     S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS   (dK/dV kernel, lanes <-> keys)
// The "swapped" orientation keeps each softmax row inside one lane (+ one cross-half exchange), so the online
// max/sum needs no 32-lane butterfly.  K/V for one (batch, head) are <= ~200 KB and stay L2/L1 resident, so they
// are read straight from global memory by each wave (cdna_hip_programming.md, common mistake 7).
// The k-slot permutation used for the key contraction (slot (hi,e) <-> key 16j + 8*(e>>2) + 4*hi + (e&3)) is applied
// identically to both operands, so the accumulator registers feed the next MFMA without any cross-lane traffic.
#include "common.h"

namespace ofa {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ bf16x8 ld_frag16(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
// two 8-byte pieces: elements 0..3 from p, 4..7 from p + 8
__device__ __forceinline__ bf16x8 ld_frag8x2(const bf16_t* p) {
  const uint2 a = *reinterpret_cast<const uint2*>(p);
  const uint2 b = *reinterpret_cast<const uint2*>(p + 8);
  union { uint4 u; bf16x8 v; } r;
  r.u = make_uint4(a.x, a.y, b.x, b.y);
  return r.v;
}
__device__ __forceinline__ bf16x8 pack8(const float* f) {
  union { uint4 u; bf16x8 v; } r;
  r.u.x = pack_bf16x2(f[0], f[1]);
  r.u.y = pack_bf16x2(f[2], f[3]);
  r.u.z = pack_bf16x2(f[4], f[5]);
  r.u.w = pack_bf16x2(f[6], f[7]);
  return r.v;
}
__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// row index inside a 32x32 MFMA accumulator for register r of a lane in half hi
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ void st4bf(bf16_t* p, float a, float b, float c, float d) {
  uint2 o;
  o.x = pack_bf16x2(a, b);
  o.y = pack_bf16x2(c, d);
  *reinterpret_cast<uint2*>(p) = o;
}

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* vt; const bf16_t* qt; const bf16_t* kt;
  const bf16_t* dot; const bf16_t* dout; const bf16_t* bias; const uint8_t* kpm; const float* c_attn;
  bf16_t* out; float* lse; const float* delta;
  bf16_t* dq; bf16_t* dk; bf16_t* dv; bf16_t* dbias;
  int B, heads, T, S, Tpad, Spad;
  int64_t ldq, ldk, ldo;
  float scale; int causal;
};

// 32-bit "dead key" mask of a 32-key block (bit j: key0+j is out of range or padded), wave-uniform, from ONE byte load
// per lane + a ballot instead of 16 byte loads per lane.
__device__ __forceinline__ uint32_t key_dead_mask(const uint8_t* kp, int key0, int S, int i) {
  const int key = key0 + i;
  bool dead = key >= S;
  if (kp && !dead) dead = kp[key] != 0;
  return (uint32_t)__ballot(dead);        // lanes 0..31 and 32..63 carry the same 32 keys; keep the low half
}

// ------------------------------------------------------------------------------------------------ forward
// Per 32-key block: S^T = K Q^T (4 MFMAs), online softmax in registers, O^T += V^T P^T (4 MFMAs).  The K rows and V^T
// rows of block kb+1 are fetched (global -> VGPR, 16 + 16 registers) while block kb is being computed; masks cost
// nothing on blocks that need none (wave-uniform branch); the O rescale runs only when some row max actually grew.
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.heads, h = bh % a.heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
  if (q0 >= a.T) return;
  const int qi = q0 + i;
  const int qrow = qi < a.T ? qi : a.T - 1;
  const bf16_t* qp = a.q + ((int64_t)b * a.T + qrow) * a.ldq + h * HD + hi * 8;
  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = ld_frag16(qp + kk * 16);
  const bf16_t* kbase = a.k + (int64_t)b * a.S * a.ldk + h * HD + hi * 8;
  const bf16_t* vtbase = a.vt + ((int64_t)bh * HD + i) * a.Spad + 4 * hi;
  const bf16_t* brow = a.bias ? a.bias + ((int64_t)bh * a.T + qrow) * a.S : nullptr;
  const uint8_t* kp = a.kpm ? a.kpm + (int64_t)b * a.S : nullptr;
  const float sc = a.scale * LOG2E;

  f32x16 ot[2];
  zero16(ot[0]);
  zero16(ot[1]);
  float m_run = -INFINITY, l_run = 0.f;
  int nkb = (a.S + 31) / 32;
  if (a.causal) {
    const int lim = (q0 + 31 < a.S - 1 ? q0 + 31 : a.S - 1) / 32 + 1;
    nkb = lim < nkb ? lim : nkb;
  }
  bf16x8 kf[4], vf[2][2];
  uint32_t kdead;
  {
    const int krow = i < a.S ? i : a.S - 1;
    const bf16_t* kr = kbase + (int64_t)krow * a.ldk;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kf[kk] = ld_frag16(kr + kk * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) vf[j][dt] = ld_frag8x2(vtbase + (int64_t)dt * 32 * a.Spad + 16 * j);
    kdead = key_dead_mask(kp, 0, a.S, i);
  }
  for (int kb = 0; kb < nkb; ++kb) {
    const int key0 = kb * 32;
    f32x16 st;
    zero16(st);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], st, 0, 0, 0);
    const uint32_t dead_now = kdead;
    bf16x8 vcur[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) vcur[j][dt] = vf[j][dt];
    if (kb + 1 < nkb) {                                  // prefetch block kb+1
      const int nk0 = key0 + 32;
      const int krow = nk0 + i < a.S ? nk0 + i : a.S - 1;
      const bf16_t* kr = kbase + (int64_t)krow * a.ldk;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[kk] = ld_frag16(kr + kk * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[j][dt] = ld_frag8x2(vtbase + (int64_t)dt * 32 * a.Spad + nk0 + 16 * j);
      kdead = key_dead_mask(kp, nk0, a.S, i);
    }
    float s[16];
    float mx = -INFINITY;
    const bool diag = a.causal && (key0 + 31 > q0);
    if (brow || dead_now || diag) {                      // wave-uniform: slow, fully general path
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = crow(r, hi), key = key0 + j;
        float t = st[r] * sc;
        if (brow && key < a.S) t += bf2f(brow[key]) * LOG2E;
        bool dead = (dead_now >> j) & 1u;
        if (diag) dead |= key > qi;
        t = dead ? -INFINITY : t;
        s[r] = t;
        mx = fmaxf(mx, t);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = st[r] * sc;
        mx = fmaxf(mx, s[r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    float p[16];
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
      ps += p[r];
    }
    ps += __shfl_xor(ps, 32, 64);
    if (__any(m_new != m_run)) {                         // some row's running max grew: rescale O and l
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ot[0][r] *= alpha;
        ot[1][r] *= alpha;
      }
      m_run = m_new;
    }
    l_run += ps;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 pf = pack8(p + 8 * j);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vcur[j][dt], pf, ot[dt], 0, 0, 0);
    }
  }
  if (qi < a.T) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    const float c = (a.c_attn ? a.c_attn[h] : 1.0f) * inv;
    bf16_t* op = a.out + ((int64_t)b * a.T + qi) * a.ldo + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        st4bf(op + dt * 32 + 8 * qq + 4 * hi, ot[dt][4 * qq] * c, ot[dt][4 * qq + 1] * c, ot[dt][4 * qq + 2] * c,
              ot[dt][4 * qq + 3] * c);
    if (hi == 0 && a.lse) a.lse[(int64_t)bh * a.Tpad + qi] = (m_run == -INFINITY ? 0.f : m_run) + log2f(l_run > 0.f ? l_run : 1.f);
  }
}

// ------------------------------------------------------------------------------------------------ backward: delta
// delta[bh, q] = sum_d dout[q,h,d] * out[q,h,d]   (row-sum of dO*O; invariant under the c_attn output scale)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                         float* __restrict__ delta, int B, int heads, int T, int Tpad,
                                                         int64_t ldo) {
  // 8 lanes per (row, head): each lane 8 elements
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t item = gid >> 3;
  const int sub = (int)(gid & 7);
  const int64_t total = (int64_t)B * T * heads;
  float s = 0.f;
  int64_t row = 0;
  int h = 0;
  if (item < total) {
    row = item / heads;
    h = (int)(item % heads);
    float x[8], y[8];
    load_vec<bf16_t>(dout + row * ldo + h * HD + sub * 8, x);
    load_vec<bf16_t>(out + row * ldo + h * HD + sub * 8, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j] * y[j];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (item < total && sub == 0) {
    const int64_t b = row / T, t = row % T;
    delta[(b * heads + h) * Tpad + t] = s;
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// lanes <-> queries.  Per 32-key block: S^T = K Q^T, dP^T = V dO^T (8 MFMAs), dS^T = P^T o (dP^T*c - delta), then
// dQ^T += K^T dS^T (4 MFMAs).  K, V rows and K^T rows of the next block are prefetched under the current one.
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.heads, h = bh % a.heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
  if (q0 >= a.T) return;
  const int qi = q0 + i;
  const int qrow = qi < a.T ? qi : a.T - 1;
  const bf16_t* qp = a.q + ((int64_t)b * a.T + qrow) * a.ldq + h * HD + hi * 8;
  const bf16_t* dop = a.dout + ((int64_t)b * a.T + qrow) * a.ldo + h * HD + hi * 8;
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    qf[kk] = ld_frag16(qp + kk * 16);
    dof[kk] = ld_frag16(dop + kk * 16);
  }
  const float lse_q = a.lse[(int64_t)bh * a.Tpad + qrow];
  const float delta_q = a.delta[(int64_t)bh * a.Tpad + qrow];
  const float c = a.c_attn ? a.c_attn[h] : 1.0f;
  const bf16_t* kbase = a.k + (int64_t)b * a.S * a.ldk + h * HD + hi * 8;
  const bf16_t* vbase = a.v + (int64_t)b * a.S * a.ldk + h * HD + hi * 8;
  const bf16_t* ktbase = a.kt + ((int64_t)bh * HD + i) * a.Spad + 4 * hi;
  const bf16_t* brow = a.bias ? a.bias + ((int64_t)bh * a.T + qrow) * a.S : nullptr;
  bf16_t* dbrow = a.dbias ? a.dbias + ((int64_t)bh * a.T + qrow) * a.S : nullptr;
  const uint8_t* kp = a.kpm ? a.kpm + (int64_t)b * a.S : nullptr;
  const float sc = a.scale * LOG2E;

  f32x16 dqt[2];
  zero16(dqt[0]);
  zero16(dqt[1]);
  int nkb = (a.S + 31) / 32;
  const int nkb_all = nkb;
  if (a.causal) {
    const int lim = (q0 + 31 < a.S - 1 ? q0 + 31 : a.S - 1) / 32 + 1;
    nkb = lim < nkb ? lim : nkb;
  }
  bf16x8 kf[4], vf[4], ktf[2][2];
  uint32_t kdead;
  {
    const int krow = i < a.S ? i : a.S - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[kk] = ld_frag16(kbase + (int64_t)krow * a.ldk + kk * 16);
      vf[kk] = ld_frag16(vbase + (int64_t)krow * a.ldk + kk * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) ktf[j][dt] = ld_frag8x2(ktbase + (int64_t)dt * 32 * a.Spad + 16 * j);
    kdead = key_dead_mask(kp, 0, a.S, i);
  }
  for (int kb = 0; kb < nkb; ++kb) {
    const int key0 = kb * 32;
    f32x16 st, dp;
    zero16(st);
    zero16(dp);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kk], dof[kk], dp, 0, 0, 0);
    }
    const uint32_t dead_now = kdead;
    bf16x8 ktc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) ktc[j][dt] = ktf[j][dt];
    if (kb + 1 < nkb) {
      const int nk0 = key0 + 32;
      const int krow = nk0 + i < a.S ? nk0 + i : a.S - 1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = ld_frag16(kbase + (int64_t)krow * a.ldk + kk * 16);
        vf[kk] = ld_frag16(vbase + (int64_t)krow * a.ldk + kk * 16);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) ktf[j][dt] = ld_frag8x2(ktbase + (int64_t)dt * 32 * a.Spad + nk0 + 16 * j);
      kdead = key_dead_mask(kp, nk0, a.S, i);
    }
    float ds[16];
    const bool diag = a.causal && (key0 + 31 > q0);
    if (brow || dead_now || diag) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = crow(r, hi), key = key0 + j;
        float t = st[r] * sc;
        if (brow && key < a.S) t += bf2f(brow[key]) * LOG2E;
        bool dead = (dead_now >> j) & 1u;
        if (diag) dead |= key > qi;
        const float p = dead ? 0.f : __builtin_amdgcn_exp2f(t - lse_q);
        ds[r] = p * (dp[r] * c - delta_q);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[r] * sc - lse_q);
        ds[r] = p * (dp[r] * c - delta_q);
      }
    }
    if (dbrow && qi < a.T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + crow(r, hi);
        if (key < a.S) dbrow[key] = f2bf(ds[r]);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 dsf = pack8(ds + 8 * j);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dqt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktc[j][dt], dsf, dqt[dt], 0, 0, 0);
    }
  }
  if (dbrow && qi < a.T && nkb < nkb_all) {   // causally skipped blocks: dS == 0
    for (int key = nkb * 32 + 4 * hi; key < a.S; key += 8)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (key + e < a.S) dbrow[key + e] = 0;
  }
  if (qi < a.T) {
    bf16_t* op = a.dq + ((int64_t)b * a.T + qi) * a.ldq + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        st4bf(op + dt * 32 + 8 * qq + 4 * hi, dqt[dt][4 * qq] * a.scale, dqt[dt][4 * qq + 1] * a.scale,
              dqt[dt][4 * qq + 2] * a.scale, dqt[dt][4 * qq + 3] * a.scale);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// lanes <-> keys.  Per 32-query block: S = Q K^T, dP = dO V^T (8 MFMAs), then dV^T += dO^T P and dK^T += Q^T dS
// (8 MFMAs).  Q, dO rows, Q^T, dO^T rows and the lse/delta vectors of the next query block are prefetched.
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.heads, h = bh % a.heads;
  const int key0 = blockIdx.x * 128 + wave * 32;
  if (key0 >= a.S) return;
  const int ki = key0 + i;
  const int krow = ki < a.S ? ki : a.S - 1;
  const bf16_t* kp_ = a.k + ((int64_t)b * a.S + krow) * a.ldk + h * HD + hi * 8;
  const bf16_t* vp_ = a.v + ((int64_t)b * a.S + krow) * a.ldk + h * HD + hi * 8;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    kf[kk] = ld_frag16(kp_ + kk * 16);
    vf[kk] = ld_frag16(vp_ + kk * 16);
  }
  const bool key_dead = ki >= a.S || (a.kpm && a.kpm[(int64_t)b * a.S + krow] != 0);
  const float live = key_dead ? 0.f : 1.f;
  const float c = a.c_attn ? a.c_attn[h] : 1.0f;
  const bf16_t* qbase = a.q + (int64_t)b * a.T * a.ldq + h * HD + hi * 8;
  const bf16_t* dobase = a.dout + (int64_t)b * a.T * a.ldo + h * HD + hi * 8;
  const bf16_t* qtbase = a.qt + ((int64_t)bh * HD + i) * a.Tpad + 4 * hi;
  const bf16_t* dotbase = a.dot + ((int64_t)bh * HD + i) * a.Tpad + 4 * hi;
  const float* lse_b = a.lse + (int64_t)bh * a.Tpad + 4 * hi;
  const float* delta_b = a.delta + (int64_t)bh * a.Tpad + 4 * hi;
  const bf16_t* bcol = a.bias ? a.bias + (int64_t)bh * a.T * a.S + krow : nullptr;
  const float sc = a.scale * LOG2E;

  f32x16 dvt[2], dkt[2];
  zero16(dvt[0]); zero16(dvt[1]); zero16(dkt[0]); zero16(dkt[1]);
  const int nqb = (a.T + 31) / 32;
  const int qb0 = a.causal ? key0 / 32 : 0;
  bf16x8 qf[4], dof[4], qtf[2][2], dotf[2][2];
  float4 l4[4], d4[4];
  auto fetch = [&](int q0) {
    const int qrow = q0 + i < a.T ? q0 + i : a.T - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[kk] = ld_frag16(qbase + (int64_t)qrow * a.ldq + kk * 16);
      dof[kk] = ld_frag16(dobase + (int64_t)qrow * a.ldo + kk * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        qtf[j][dt] = ld_frag8x2(qtbase + (int64_t)dt * 32 * a.Tpad + q0 + 16 * j);
        dotf[j][dt] = ld_frag8x2(dotbase + (int64_t)dt * 32 * a.Tpad + q0 + 16 * j);
      }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      l4[g4] = *reinterpret_cast<const float4*>(lse_b + q0 + 8 * g4);
      d4[g4] = *reinterpret_cast<const float4*>(delta_b + q0 + 8 * g4);
    }
  };
  if (qb0 < nqb) fetch(qb0 * 32);
  for (int qb = qb0; qb < nqb; ++qb) {
    const int q0 = qb * 32;
    f32x16 st, dp;
    zero16(st);
    zero16(dp);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[kk], kf[kk], st, 0, 0, 0);    // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[kk], vf[kk], dp, 0, 0, 0);   // dP[q][key]
    }
    bf16x8 qtc[2][2], dotc[2][2];
    float lv[16], dv16[16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) { qtc[j][dt] = qtf[j][dt]; dotc[j][dt] = dotf[j][dt]; }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      lv[4 * g4] = l4[g4].x; lv[4 * g4 + 1] = l4[g4].y; lv[4 * g4 + 2] = l4[g4].z; lv[4 * g4 + 3] = l4[g4].w;
      dv16[4 * g4] = d4[g4].x; dv16[4 * g4 + 1] = d4[g4].y; dv16[4 * g4 + 2] = d4[g4].z; dv16[4 * g4 + 3] = d4[g4].w;
    }
    if (qb + 1 < nqb) fetch(q0 + 32);
    float p[16], ds[16];
    const bool general = bcol || (q0 + 32 > a.T) || (a.causal && (key0 + 31 > q0));
    if (general) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = q0 + crow(r, hi);
        float t = st[r] * sc;
        if (bcol && q < a.T) t += bf2f(bcol[(int64_t)q * a.S]) * LOG2E;
        bool dead = key_dead || q >= a.T;
        if (a.causal) dead |= ki > q;
        const float pv = dead ? 0.f : __builtin_amdgcn_exp2f(t - lv[r]);
        p[r] = pv;
        ds[r] = dead ? 0.f : pv * (dp[r] * c - dv16[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(st[r] * sc - lv[r]) * live;
        p[r] = pv;
        ds[r] = pv * (dp[r] * c - dv16[r]);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 pf = pack8(p + 8 * j);
      const bf16x8 dsf = pack8(ds + 8 * j);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotc[j][dt], pf, dvt[dt], 0, 0, 0);
        dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtc[j][dt], dsf, dkt[dt], 0, 0, 0);
      }
    }
  }
  if (ki < a.S) {
    bf16_t* dkp = a.dk + ((int64_t)b * a.S + ki) * a.ldk + h * HD;
    bf16_t* dvp = a.dv + ((int64_t)b * a.S + ki) * a.ldk + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int d = dt * 32 + 8 * qq + 4 * hi;
        st4bf(dkp + d, dkt[dt][4 * qq] * a.scale, dkt[dt][4 * qq + 1] * a.scale, dkt[dt][4 * qq + 2] * a.scale,
              dkt[dt][4 * qq + 3] * a.scale);
        st4bf(dvp + d, dvt[dt][4 * qq] * c, dvt[dt][4 * qq + 1] * c, dvt[dt][4 * qq + 2] * c, dvt[dt][4 * qq + 3] * c);
      }
  }
}

// ------------------------------------------------------------------------------------------------ [B,T,C] -> [B,C,Tpad]
template <typename T>
__global__ __launch_bounds__(256) void transpose_heads_kernel(const T* __restrict__ x, T* __restrict__ xt, int Tn, int C,
                                                             int Tpad, int64_t ld) {
  __shared__ T tile[64][66];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int t = t0 + r, c = c0 + tx;
    T v = 0;
    if (t < Tn && c < C) v = x[((int64_t)b * Tn + t) * ld + c];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, t = t0 + tx;
    if (c < C && t < Tpad) xt[((int64_t)b * C + c) * Tpad + t] = tile[tx][r];
  }
}

static int attn_check(int B, int heads, int T, int S, int Tpad, int Spad, int64_t ldq, int64_t ldk, int64_t ldo,
                      int dtype) {
  OFA_REQUIRE(dtype == OFA_BF16, OFA_ERR_UNSUPPORTED, "fused attention is bf16 only (dtype %d); use the unfused path", dtype);
  OFA_REQUIRE(B > 0 && heads > 0 && T > 0 && S > 0, OFA_ERR_INVALID, "attention: bad shape B=%d heads=%d T=%d S=%d", B, heads, T, S);
  OFA_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldo % 8) == 0, OFA_ERR_INVALID, "attention: leading dims must be multiples of 8");
  OFA_REQUIRE(Spad % 32 == 0 && Spad >= S && Tpad % 32 == 0 && Tpad >= T, OFA_ERR_INVALID,
              "attention: Tpad/Spad must be multiples of 32 covering T/S (T=%d Tpad=%d S=%d Spad=%d)", T, Tpad, S, Spad);
  return 0;
}

}  // namespace ofa
using namespace ofa;

extern "C" int ofa_attn_fwd(const void* q, const void* k, const void* vt, const void* bias, const uint8_t* kpm,
                            const float* c_attn, void* out, float* lse, int B, int heads, int T, int S, int Tpad,
                            int Spad, int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal, int dtype,
                            void* stream) {
  if (int rc = attn_check(B, heads, T, S, Tpad, Spad, ldq, ldk, ldo, dtype)) return rc;
  OFA_REQUIRE(q && k && vt && out, OFA_ERR_INVALID, "attn_fwd: null pointer");
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.bias = (const bf16_t*)bias; a.kpm = kpm;
  a.c_attn = c_attn; a.out = (bf16_t*)out; a.lse = lse; a.B = B; a.heads = heads; a.T = T; a.S = S; a.Tpad = Tpad;
  a.Spad = Spad; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.scale = scale; a.causal = causal;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(T, 128), B * heads), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("attn_fwd");
}

extern "C" int ofa_attn_bwd_prep(const void* dout, const void* out, float* delta, int B, int heads, int T, int Tpad,
                                 int64_t ldo, int dtype, void* stream) {
  OFA_REQUIRE(dtype == OFA_BF16, OFA_ERR_UNSUPPORTED, "attn_bwd_prep: bf16 only");
  OFA_REQUIRE(dout && out && delta && (ldo % 8) == 0 && Tpad >= T, OFA_ERR_INVALID, "attn_bwd_prep: bad argument");
  const int64_t threads = (int64_t)B * T * heads * 8;
  hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                     (const bf16_t*)out, delta, B, heads, T, Tpad, ldo);
  return check_launch("attn_bwd_prep");
}

extern "C" int ofa_attn_bwd(const void* q, const void* k, const void* v, const void* qt, const void* kt, const void* dot,
                            const void* dout, const void* bias, const uint8_t* kpm, const float* c_attn, const float* lse,
                            const float* delta, void* dq, void* dk, void* dv, void* dbias, int B, int heads, int T, int S,
                            int Tpad, int Spad, int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal,
                            int dtype, void* stream) {
  if (int rc = attn_check(B, heads, T, S, Tpad, Spad, ldq, ldk, ldo, dtype)) return rc;
  OFA_REQUIRE(q && k && v && qt && kt && dot && dout && lse && delta && dq && dk && dv, OFA_ERR_INVALID,
              "attn_bwd: null pointer");
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.qt = (const bf16_t*)qt;
  a.kt = (const bf16_t*)kt; a.dot = (const bf16_t*)dot; a.dout = (const bf16_t*)dout; a.bias = (const bf16_t*)bias;
  a.kpm = kpm; a.c_attn = c_attn; a.lse = const_cast<float*>(lse); a.delta = delta; a.dq = (bf16_t*)dq;
  a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.dbias = (bf16_t*)dbias; a.B = B; a.heads = heads; a.T = T; a.S = S;
  a.Tpad = Tpad; a.Spad = Spad; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.scale = scale; a.causal = causal;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(cdiv(T, 128), B * heads), dim3(256), 0, st, a);
  int rc = check_launch("attn_bwd_dq");
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(cdiv(S, 128), B * heads), dim3(256), 0, st, a);
  return check_launch("attn_bwd_dkv");
}

// out[b][i] = mean_a p[b][a][i]  (head-averaged attention weights, multihead_attention.py:347-351)
namespace ofa {
template <typename T>
__global__ __launch_bounds__(256) void mean_heads_kernel(const T* __restrict__ p, T* __restrict__ out, int heads, int64_t n) {
  const int b = blockIdx.y;
  const float inv = 1.0f / (float)heads;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int a = 0; a < heads; ++a) s += ld1<T>(p + ((int64_t)b * heads + a) * n + i);
    st1<T>(out + (int64_t)b * n + i, s * inv);
  }
}
}  // namespace ofa

extern "C" int ofa_mean_heads(const void* p, void* out, int B, int heads, int64_t n, int dtype, void* stream) {
  OFA_REQUIRE(p && out && B > 0 && heads > 0 && n > 0, OFA_ERR_INVALID, "mean_heads: bad argument");
  OFA_REQUIRE(dtype == OFA_F32 || dtype == OFA_BF16, OFA_ERR_INVALID, "mean_heads: bad dtype %d", dtype);
  int64_t gx = (n + 255) / 256;
  dim3 grid((unsigned)(gx > 1024 ? 1024 : gx), B), block(256);
  if (dtype == OFA_F32)
    hipLaunchKernelGGL((mean_heads_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)p, (float*)out, heads, n);
  else
    hipLaunchKernelGGL((mean_heads_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)p, (bf16_t*)out, heads, n);
  return check_launch("mean_heads");
}

extern "C" int ofa_transpose_heads(const void* x, void* xt, int B, int T, int C, int Tpad, int64_t ld, int dtype,
                                   void* stream) {
  OFA_REQUIRE(x && xt && B > 0 && T > 0 && C > 0 && Tpad >= T && ld >= C, OFA_ERR_INVALID, "transpose_heads: bad argument");
  dim3 grid(cdiv(Tpad, 64), cdiv(C, 64), B), block(256);
  if (dtype == OFA_BF16)
    hipLaunchKernelGGL((transpose_heads_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)xt,
                       T, C, Tpad, ld);
  else
    hipLaunchKernelGGL((transpose_heads_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)xt, T,
                       C, Tpad, ld);
  return check_launch("transpose_heads");
}
