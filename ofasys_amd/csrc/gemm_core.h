// Shared device-side pieces of the MFMA GEMM kernels (gemm_mfma.hip: the compiler-scheduled loops; gemm_pp.hip: the ping-pong loop):
// 16-bit element helpers, LDS-DMA staging with the source-side swizzle, fragment addressing / inline-asm fragment reads, the
// coalesced LDS-bounce epilogue.  Header-only, every function __device__ __forceinline__.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "gemm.h"

namespace ofa {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
// The 16-bit element type is a template flag (F16): bf16 (default, v_mfma_f32_32x32x16_bf16) or fp16 (v_mfma_f32_32x32x16_f16).  Data
// movement (LDS-DMA, fragment reads, swizzles) is the same for both; only the MFMA and the epilogue's conversions differ.
template <bool F16> __device__ __forceinline__ float lo16(uint32_t w) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16x2_t, w)[0];
  else return __uint_as_float(w << 16);
}
template <bool F16> __device__ __forceinline__ float hi16(uint32_t w) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16x2_t, w)[1];
  else return __uint_as_float(w & 0xffff0000u);
}
template <bool F16> __device__ __forceinline__ uint32_t enc2(float lo, float hi) {
  if constexpr (F16) return pack_f16x2(lo, hi);
  else return pack_bf16x2(lo, hi);
}
template <bool F16> __device__ __forceinline__ float dec1(uint16_t u) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16_t, u);
  else return bf2f(u);
}
template <bool F16, typename A, typename B, typename C>
__device__ __forceinline__ C mfma16(const A& a, const B& b, const C& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int BK = 64;
constexpr int KMAJ_LD = BK + 8;  // elements per LDS row of a k-major tile

template <int R, bool KMAJ> struct TileGeom {
  static constexpr int LD = KMAJ ? KMAJ_LD : (R + 32);
  static constexpr int ELEMS = KMAJ ? R * KMAJ_LD : BK * (R + 32);
  static constexpr int NVEC = R * BK / 8;
};

// Stage one operand tile: global -> registers.
//  KMAJ: element (r, k) at base[(r0+r)*ld + k];   vectors run along k.
// !KMAJ: element (r, k) at base[k*ld + r0 + r];   vectors run along r.
template <int R, bool KMAJ, int NT, int NV>
__device__ __forceinline__ void stage_load(uint4 (&reg)[NV], const bf16_t* __restrict__ base, int64_t ld, int r0,
                                           int rmax, int k0, int kend, int tid) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * NT;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (KMAJ) {
      const int r = v >> 3, c = (v & 7) * 8;
      int rr = r0 + r;
      rr = rr < rmax ? rr : rmax - 1;
      if (k0 + c < kend) val = *reinterpret_cast<const uint4*>(base + (int64_t)rr * ld + k0 + c);
    } else {
      constexpr int VPR = R / 8;
      const int k = v / VPR, c = (v % VPR) * 8;
      // a ragged last vector stays inside ld (launch precondition)
      if (k0 + k < kend && r0 + c < ((rmax + 7) & ~7)) val = *reinterpret_cast<const uint4*>(base + (int64_t)(k0 + k) * ld + r0 + c);
    }
    reg[i] = val;
  }
}

template <int R, bool KMAJ, int NT, int NV>
__device__ __forceinline__ void stage_store(const uint4 (&reg)[NV], bf16_t* __restrict__ lds, int tid) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * NT;
    int off;
    if (KMAJ) {
      off = (v >> 3) * KMAJ_LD + (v & 7) * 8;
    } else {
      constexpr int VPR = R / 8;
      off = (v / VPR) * (R + 32) + (v % VPR) * 8;
    }
    *reinterpret_cast<uint4*>(lds + off) = reg[i];
  }
}

// MFMA operand fragment for rows [rbase, rbase+32) of the tile and k-slice kk (16 wide): lane (i = l&31, hi = l>>5)
// receives elements (row rbase+i, k = kk*16 + hi*8 + 0..7).
template <int R, bool KMAJ>
__device__ __forceinline__ bf16x8 load_frag(const bf16_t* __restrict__ lds, int rbase, int kk, int lane) {
  if (KMAJ) {
    const int i = lane & 31, hi = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(lds + (rbase + i) * KMAJ_LD + kk * 16 + hi * 8);
  } else {
    // transposing read: in each 16-lane group, lane q supplies the address of 4 consecutive rows-elements
    // (k = kb + (q>>2), r = rb + 4*(q&3) .. +3) and receives (k = kb + 0..3, r = rb + q).
    const int g = lane >> 4, q = lane & 15;
    constexpr int LD = R + 32;
    const int kb = kk * 16 + (g >> 1) * 8;
    const int rb = rbase + (g & 1) * 16;
    const bf16_t* p0 = lds + (kb + (q >> 2)) * LD + rb + 4 * (q & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p0));
    bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p0 + 4 * LD));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return r;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA staging (global_load_lds_dwordx4): the tile goes HBM/L2 -> LDS without passing through VGPRs or the
// VGPR->LDS write path (ds_write_b128 sustains only ~79 B/clk/CU, which made the register-staged loop LDS-write
// bound: PMC showed MFMA busy 21%, 65% of wave cycles stalled on issue).  The DMA writes lane-linear (wave-uniform
// base + lane*16 B), so the tile is stored unpadded and the bank-conflict fix is an XOR swizzle of the 16-byte chunk
// index applied to the per-lane SOURCE address and again when reading (cdna_hip_programming.md rule 21):
//   k-major [R][64]:   chunk c (8 per row) of row r lives at c ^ ((r>>1)&7)   -> 16 rows x one k-slice = 16 distinct slots
//   m-major [64][R]:   chunk c of k-row k lives at c ^ ((k&3)<<2) (R=128, 256) / c ^ (((k>>1)&1)<<2) (R=64)
//                      -> the 4 k-rows of a ds_read_b64_tr_b16 group fall in 4 different 64-byte bank quarters.
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <int R, bool KMAJ> __device__ __forceinline__ int swz(int row, int c) {
  if (KMAJ) return c ^ ((row >> 1) & 7);
  return R >= 128 ? (c ^ ((row & 3) << 2)) : (c ^ (((row >> 1) & 1) << 2));
}

template <int R, bool KMAJ, int NT, int NV>
__device__ __forceinline__ void stage_glds(const bf16_t* __restrict__ base, int64_t ld, int r0, int rmax, int k0,
                                           bf16_t* __restrict__ lds, int tid, int wave) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int gidx = tid + i * NT;                       // physical 16-byte granule of the tile
    const bf16_t* src;
    if (KMAJ) {
      const int r = gidx >> 3, c = swz<R, true>(r, gidx & 7);
      int rr = r0 + r;
      rr = rr < rmax ? rr : rmax - 1;
      src = base + (int64_t)rr * ld + k0 + c * 8;
    } else {
      constexpr int CPR = R / 8;
      const int k = gidx / CPR, c = swz<R, false>(k, gidx % CPR);
      int col = r0 + c * 8;
      const int last = ((rmax + 7) & ~7) - 8;
      col = col < last ? col : last;
      src = base + (int64_t)(k0 + k) * ld + col;
    }
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(lds + (wave * 64 + i * NT) * 8), 16, 0, 0);
  }
}

// Fragment reads of the DMA image are issued as inline asm: hipcc treats an in-flight LDS-DMA as a pending LDS write
// and would put `s_waitcnt vmcnt(0)` in front of every compiler-visible ds_read, draining the next tile's DMA before
// the first MFMA (no overlap at all -- seen in the ISA).  The asm reads are invisible to that pass; their own
// completion is waited for with an explicit lgkmcnt statement that names every destination register ("+v"), which is
// what orders the consuming MFMAs behind it (cdna_hip_programming.md section 5.7, form (ii)).
typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
struct FragRegs { bf16x8 v; };

// Per-lane LDS byte addresses of one operand's fragments, computed ONCE per kernel: the K loop is unrolled by two so
// the double-buffer select becomes a compile-time `offset:` immediate and a K-step issues its 16 fragment reads with no
// address arithmetic at all (PMC: instruction issue used to cost as many cycles per K-step as the MFMAs themselves).
template <int R, bool KMAJ> struct FragAddr {
  uint32_t a[KMAJ ? 4 : 1];          // k-major: one address per k-slice (the XOR swizzle depends on kk); m-major: base
  __device__ __forceinline__ void init(uint32_t tile0, int rbase, int lane) {
    if (KMAJ) {
      const int row = rbase + (lane & 31), hi = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[kk] = tile0 + (uint32_t)(row * 64 + swz<R, true>(row, kk * 2 + hi) * 8) * 2u;
    } else {
      const int g = lane >> 4, q = lane & 15;
      const int k = (g >> 1) * 8 + (q >> 2);
      const int col = rbase + (g & 1) * 16 + 4 * (q & 3);
      a[0] = tile0 + (uint32_t)(k * R + swz<R, false>(k, col >> 3) * 8 + (col & 7)) * 2u;   // swz(k + 16*kk + 4, .) == swz(k, .)
    }
  }
};

template <int R, bool KMAJ, int KK, int BUFOFF>
__device__ __forceinline__ void frag_issue(u64x2& d, const FragAddr<R, KMAJ>& fa) {
  if constexpr (KMAJ) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(fa.a[KK]), "i"(BUFOFF));
  } else {
    unsigned long long lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(fa.a[0]), "i"(BUFOFF + KK * 16 * R * 2));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(fa.a[0]), "i"(BUFOFF + KK * 16 * R * 2 + 4 * R * 2));
    d[0] = lo;
    d[1] = hi;
  }
}

// 16 zero bytes: the LDS-DMA source of the contraction rows past the end of an m-major A (glds_ptrs<.., ZERO>).  A weight gradient
// dW = dY^T X contracts over the ROWS; when their number is not a multiple of the 64-row K tile (1568 = 8 x 14 x 14 positions of the
// ResNet trunk at micro-batch 8; any ragged batch), the last tile's missing rows of A are fetched from here -- the DMA source
// address is per lane and arbitrary -- and B's are clamped to its last row: 0 x finite = 0.  Such products used to take the
// register-staged loop (46 us for a product the DMA loop does in ~10: 14 % of the cfg-3 step, round 4 profile).
static __device__ uint4 ofa_zero16 = {0u, 0u, 0u, 0u};   // (one copy per translation unit)

// DMA source pointers of one operand tile, advanced by a constant stride per K-step.
template <int R, bool KMAJ, int NT, int NV, bool ZERO = false>
__device__ __forceinline__ void glds_ptrs(const bf16_t* (&ptr)[NV], const bf16_t* __restrict__ base, int64_t ld, int r0,
                                          int rmax, int k0, int tid, int krows = 0x7fffffff) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int gidx = tid + i * NT;
    if (KMAJ) {
      const int r = gidx >> 3, c = swz<R, true>(r, gidx & 7);
      int rr = r0 + r;
      rr = rr < rmax ? rr : rmax - 1;
      ptr[i] = base + (int64_t)rr * ld + k0 + c * 8;
    } else {
      constexpr int CPR = R / 8;
      const int k = gidx / CPR, c = swz<R, false>(k, gidx % CPR);
      int col = r0 + c * 8;
      const int last = ((rmax + 7) & ~7) - 8;
      col = col < last ? col : last;
      int kr = k0 + k;
      const bool past = kr >= krows;
      kr = past ? krows - 1 : kr;             // rows past the operand's end (zero-padded contraction tail) are clamped ...
      ptr[i] = base + (int64_t)kr * ld + col;
      if (ZERO && past) ptr[i] = reinterpret_cast<const bf16_t*>(&ofa_zero16);      // ... or, for A, read as zeros
    }
  }
}
template <int NT, int NV>
__device__ __forceinline__ void glds_issue(const bf16_t* (&ptr)[NV], int64_t step, bf16_t* __restrict__ lds, int wave) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    __builtin_amdgcn_global_load_lds((gvoid_t*)ptr[i], (lvoid_t*)(lds + (wave * 64 + i * NT) * 8), 16, 0, 0);
    ptr[i] += step;
  }
}

// (tile, K-slice) of this workgroup.  Workgroups are dispatched in linear order (x fastest, then y) round-robin over the 8 XCDs, and
// each XCD has its own L2.  Without split-K the tile ids of an XCD are made contiguous (xcd_remap).  With split-K (gridDim.y
// slices) the remap runs over the FLATTENED (slice, tile) index, slice-major: an XCD then works through whole K-slices -- every
// tile of a slice reads the same [ksplit x M] / [ksplit x N] operand panels, which stay in that XCD's L2 -- instead of every
// XCD touching every slice of a weight-gradient product (round-1 PMC: 2.7x the algorithmic bytes fetched per GEMM launch).
__device__ __forceinline__ void tile_and_slice(int ntiles, int& t, int& ks) {
  if (gridDim.y == 1) {
    t = xcd_remap(blockIdx.x, ntiles);
    ks = 0;
    return;
  }
  const int id = xcd_remap((int)(blockIdx.x + blockIdx.y * gridDim.x), ntiles * (int)gridDim.y);
  ks = id / ntiles;
  t = id - ks * ntiles;
}

// epilogue on 4 consecutive columns n..n+3 of row m
template <bool OUT_F32, bool F16 = false>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, void* Cb, int m, int n, float v0, float v1, float v2,
                                               float v3) {
  float v[4] = {v0, v1, v2, v3};
  if (g.flags & OFA_GEMM_BIAS_COL) {
    const uint2 b = *reinterpret_cast<const uint2*>((const bf16_t*)g.bias + n);
    v[0] += lo16<F16>(b.x); v[1] += hi16<F16>(b.x);
    v[2] += lo16<F16>(b.y); v[3] += hi16<F16>(b.y);
  }
  if (g.flags & OFA_GEMM_BIAS_ROW) {
    const float b = dec1<F16>(((const bf16_t*)g.bias)[m]);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += b;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] *= g.alpha;
  if (OUT_F32) {
    float* p = (float*)Cb + (int64_t)m * g.ldc + n;
    if (g.flags & OFA_GEMM_ACCUM) {
      const float4 o = *reinterpret_cast<const float4*>(p);
      v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
    }
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    bf16_t* p = (bf16_t*)Cb + (int64_t)m * g.ldc + n;
    if (g.flags & OFA_GEMM_ACCUM) {
      const uint2 o = *reinterpret_cast<const uint2*>(p);
      v[0] += lo16<F16>(o.x); v[1] += hi16<F16>(o.x);
      v[2] += lo16<F16>(o.y); v[3] += hi16<F16>(o.y);
    }
    uint2 o;
    o.x = enc2<F16>(v[0], v[1]);
    o.y = enc2<F16>(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = o;
  }
}

// Coalesced epilogue.  The MFMA accumulator layout gives each lane 4 consecutive columns of ONE row per register quad,
// so storing straight from registers makes every store instruction touch 64 different rows (64 x 8 B): the memory pipe
// handles that at one row per clock, ~8 us for a 256x256 tile -- as long as the whole K loop at K = 768.  Instead each
// wave bounces its block through its private slice of the (now idle) LDS stages: quads go in with an XOR swizzle on the
// 16-byte chunk index, come back out as whole 16-byte row segments, and one store instruction writes 2-8 full rows.
// Bias / alpha are applied on the way in, C-accumulation on the way out.  RAW: split-K partials (fp32, no bias/alpha).
template <int TM, int TN, bool F32, bool RAW, bool F16 = false>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& g, const f32x16 (&acc)[TM][TN], unsigned char* __restrict__ wl,
                                             int region_bytes, void* __restrict__ Cb, int64_t ldc, int m_w, int n_w,
                                             int lane) {
  constexpr int E = F32 ? 4 : 2;
  constexpr int ROWB = TN * 32 * E;                    // bytes per staged row
  constexpr int CH = ROWB / 16;                        // 16-byte chunks per row
  constexpr int SH = ROWB < 256 ? 1 : 0;
  constexpr int LPR = CH;                              // lanes per row when reading back
  constexpr int RPI = 64 / LPR;                        // rows per store instruction
  const int hi = lane >> 5, ml = lane & 31;
  const int ipass = region_bytes / (32 * ROWB) < TM ? region_bytes / (32 * ROWB) : TM;   // 32-row tiles per pass
  const bool vec16 = F32 || ((ldc & 7) == 0);
  // column bias of the 4*TN quads this lane owns: loaded ONCE, up front (a load inside the quad loop costs a full
  // memory round trip per quad -- hipcc waits for each one in place)
  float bcol[TN][4][4];
  const bool has_bcol = !RAW && (g.flags & OFA_GEMM_BIAS_COL);
  if (has_bcol) {
    uint2 braw[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n_w + j * 32 + 8 * q + 4 * hi;
        braw[j][q] = n < g.N ? *reinterpret_cast<const uint2*>((const bf16_t*)g.bias + n) : make_uint2(0, 0);
      }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bcol[j][q][0] = lo16<F16>(braw[j][q].x); bcol[j][q][1] = hi16<F16>(braw[j][q].x);
        bcol[j][q][2] = lo16<F16>(braw[j][q].y); bcol[j][q][3] = hi16<F16>(braw[j][q].y);
      }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bcol[j][q][e] = 0.f;
  }
  const float alpha = RAW ? 1.0f : g.alpha;
  // Column statistics of the rounded tile (g.colstat, 16-bit outputs): the 16-byte row segments this lane reads back are 8 columns of
  // rows lane / LPR, + RPI, ...: their sums and sums of squares ride in 16 registers, are folded over the RPI row lanes at the end and
  // land in ONE partial row per wave block -- the statistics pass of the BatchNorm that follows a convolution (csrc/conv.hip) reads
  // nothing but these partial rows (module/resnet.py:105-128: every convolution of the trunk is followed by a BatchNorm).
  const bool stats = !F32 && !RAW && g.colstat != nullptr;
  float cs[8], cq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = cq[j] = 0.f;
  auto stat8 = [&](const uint4& u) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = lo16<F16>(w[j]), hi = hi16<F16>(w[j]);
      cs[2 * j] += lo; cq[2 * j] += lo * lo;
      cs[2 * j + 1] += hi; cq[2 * j + 1] += hi * hi;
    }
  };
  for (int ip0 = 0; ip0 < TM; ip0 += ipass) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i < ip0 || i >= ip0 + ipass) continue;
      const int mloc = (i - ip0) * 32 + ml;
      const int sw = (mloc >> SH) & (CH - 1);
      float brow = 0.f;
      if (!RAW && (g.flags & OFA_GEMM_BIAS_ROW)) {
        const int m = m_w + i * 32 + ml;
        brow = dec1<F16>(((const bf16_t*)g.bias)[m < g.M ? m : g.M - 1]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nloc = j * 32 + 8 * q + 4 * hi;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * q + e] + (bcol[j][q][e] + brow)) * alpha;
          if (F32) {
            *reinterpret_cast<float4*>(wl + mloc * ROWB + (((nloc >> 2) ^ sw) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 o;
            o.x = enc2<F16>(v[0], v[1]);
            o.y = enc2<F16>(v[2], v[3]);
            *reinterpret_cast<uint2*>(wl + mloc * ROWB + (((nloc >> 3) ^ sw) << 4) + ((nloc >> 2) & 1) * 8) = o;
          }
        }
      }
    }
    // read back whole 16-byte row segments (same wave: LDS operations complete in order) and store.  The memory clobber
    // keeps hipcc from hoisting the uint4 reads above the uint2 / float4 writes (different types: TBAA says "no alias").
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int c = lane % LPR;
    const int n = n_w + c * (16 / E);
    const bool acc16 = !RAW && !F32 && (g.flags & OFA_GEMM_ACCUM);     // C += tile, 16-bit: one extra 16-byte load per store
    if (vec16 && m_w + TM * 32 <= g.M && n_w + TN * 32 <= g.N && (RAW || !(g.flags & OFA_GEMM_ACCUM) || acc16)) {
      // interior tile, plain store: straight-line, 8 LDS reads in flight per batch (the guarded loop below pays an LDS
      // round trip plus ~10 branches per 16-byte store -- ~8 us for a 256x256 tile, measured with the K loop removed)
      const int rows_full = (TM - ip0 < ipass ? TM - ip0 : ipass) * 32;
      unsigned char* p = (unsigned char*)Cb + ((int64_t)(m_w + ip0 * 32 + lane / LPR) * ldc + n) * E;
      const int64_t pstep = (int64_t)RPI * ldc * E;
      if (acc16) {
#pragma unroll 8
        for (int r0 = 0; r0 < rows_full; r0 += RPI) {
          const int mloc = r0 + lane / LPR;
          const uint4 u = *reinterpret_cast<const uint4*>(wl + mloc * ROWB + ((c ^ ((mloc >> SH) & (CH - 1))) << 4));
          const uint4 old = *reinterpret_cast<const uint4*>(p);
          uint4 o;
          o.x = enc2<F16>(lo16<F16>(u.x) + lo16<F16>(old.x), hi16<F16>(u.x) + hi16<F16>(old.x));
          o.y = enc2<F16>(lo16<F16>(u.y) + lo16<F16>(old.y), hi16<F16>(u.y) + hi16<F16>(old.y));
          o.z = enc2<F16>(lo16<F16>(u.z) + lo16<F16>(old.z), hi16<F16>(u.z) + hi16<F16>(old.z));
          o.w = enc2<F16>(lo16<F16>(u.w) + lo16<F16>(old.w), hi16<F16>(u.w) + hi16<F16>(old.w));
          *reinterpret_cast<uint4*>(p) = o;
          p += pstep;
        }
        continue;
      }
      if (stats) {
#pragma unroll 8
        for (int r0 = 0; r0 < rows_full; r0 += RPI) {
          const int mloc = r0 + lane / LPR;
          const uint4 u = *reinterpret_cast<const uint4*>(wl + mloc * ROWB + ((c ^ ((mloc >> SH) & (CH - 1))) << 4));
          *reinterpret_cast<uint4*>(p) = u;
          stat8(u);
          p += pstep;
        }
        continue;
      }
#pragma unroll 8
      for (int r0 = 0; r0 < rows_full; r0 += RPI) {
        const int mloc = r0 + lane / LPR;
        const uint4 u = *reinterpret_cast<const uint4*>(wl + mloc * ROWB + ((c ^ ((mloc >> SH) & (CH - 1))) << 4));
        *reinterpret_cast<uint4*>(p) = u;
        p += pstep;
      }
      continue;
    }
    const int rows_here = (TM - ip0 < ipass ? TM - ip0 : ipass) * 32;
    for (int r0 = 0; r0 < rows_here; r0 += RPI) {
      const int mloc = r0 + lane / LPR;
      const int m = m_w + ip0 * 32 + mloc;
      const uint4 u = *reinterpret_cast<const uint4*>(wl + mloc * ROWB + ((c ^ ((mloc >> SH) & (CH - 1))) << 4));
      if (m >= g.M || n >= g.N) continue;
      if (stats) stat8(u);                             // (N % 8 == 0 when statistics are requested: the whole chunk is inside the row)
      if (F32) {
        float* p = (float*)Cb + (int64_t)m * ldc + n;
        float4 o = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        if (!RAW && (g.flags & OFA_GEMM_ACCUM)) {
          const float4 old = *reinterpret_cast<const float4*>(p);
          o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *reinterpret_cast<float4*>(p) = o;
      } else {
        bf16_t* p = (bf16_t*)Cb + (int64_t)m * ldc + n;
        const bool second = n + 4 < g.N;                // the chunk's second quad (inside ldc by the launch precondition)
        uint4 o = u;
        if (g.flags & OFA_GEMM_ACCUM) {
          uint4 old = make_uint4(0, 0, 0, 0);
          if (vec16 && second) old = *reinterpret_cast<const uint4*>(p);
          else {
            const uint2 a = *reinterpret_cast<const uint2*>(p);
            old.x = a.x; old.y = a.y;
            if (second) { const uint2 b = *reinterpret_cast<const uint2*>(p + 4); old.z = b.x; old.w = b.y; }
          }
          o.x = enc2<F16>(lo16<F16>(u.x) + lo16<F16>(old.x), hi16<F16>(u.x) + hi16<F16>(old.x));
          o.y = enc2<F16>(lo16<F16>(u.y) + lo16<F16>(old.y), hi16<F16>(u.y) + hi16<F16>(old.y));
          o.z = enc2<F16>(lo16<F16>(u.z) + lo16<F16>(old.z), hi16<F16>(u.z) + hi16<F16>(old.z));
          o.w = enc2<F16>(lo16<F16>(u.w) + lo16<F16>(old.w), hi16<F16>(u.w) + hi16<F16>(old.w));
        }
        if (vec16 && second) *reinterpret_cast<uint4*>(p) = o;
        else {
          *reinterpret_cast<uint2*>(p) = make_uint2(o.x, o.y);
          if (second) *reinterpret_cast<uint2*>(p + 4) = make_uint2(o.z, o.w);
        }
      }
    }
  }
  if (stats) {
    // fold over the RPI lanes that hold the same 8 columns (lane % LPR), then lanes 0 .. LPR - 1 write the wave block's partial row
#pragma unroll
    for (int mask = LPR; mask < 64; mask <<= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cs[j] += __shfl_xor(cs[j], mask, 64);
        cq[j] += __shfl_xor(cq[j], mask, 64);
      }
    }
    const int grp = m_w / (TM * 32);
    const int n = n_w + (lane % LPR) * 8;
    if (lane < LPR && m_w < g.M && n < g.N) {
      double* ps = g.colstat + ((int64_t)grp * 2) * g.N + n;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ps[j] = (double)cs[j];
        ps[g.N + j] = (double)cq[j];
      }
    }
  }
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// per-lane LDS byte addresses (relative to LDS base, buffer 0 of the operand)
template <int R, bool KMAJ> struct BigAddr {
  uint32_t a[4];   // k-major: one per k-slice (tile index is an immediate); m-major: one per 32-row tile (k-slice immediate)
  __device__ __forceinline__ void init(uint32_t op0, int rbase, int lane) {
    if (KMAJ) {
      const int row = rbase + (lane & 31), hi = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[kk] = op0 + (uint32_t)(row * 64 + swz<R, true>(row, kk * 2 + hi) * 8) * 2u;
    } else {
      const int g = lane >> 4, q = lane & 15;
      const int k = (g >> 1) * 8 + (q >> 2);
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const int col = rbase + ti * 32 + (g & 1) * 16 + 4 * (q & 3);
        a[ti] = op0 + (uint32_t)(k * R + swz<R, false>(k, col >> 3) * 8 + (col & 7)) * 2u;
      }
    }
  }
};

// issue the reads of fragment `TI` (32 rows) of k-slice KK from the buffer at byte offset BUFOFF
template <int R, bool KMAJ, int KK, int TI, int BUFOFF>
__device__ __forceinline__ void big_frag(u64x2& d, const BigAddr<R, KMAJ>& fa) {
  if constexpr (KMAJ) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(fa.a[KK]), "i"(BUFOFF + TI * 32 * 128));
  } else {
    unsigned long long lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(fa.a[TI]), "i"(BUFOFF + KK * 16 * R * 2));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(fa.a[TI]), "i"(BUFOFF + KK * 16 * R * 2 + 4 * R * 2));
    d[0] = lo;
    d[1] = hi;
  }
}

// which fragment read (0..nr-1, or -1) follows MFMA number t of a slice: one read behind each of the first nr MFMAs
__host__ __device__ constexpr int big_read_after(int t, int nr) { return t < nr ? t : -1; }

// Grouped weight-gradient launch (ofa_gemm_group_tn): argument block shared by gemm_mfma.hip and gemm_pp.hip
constexpr int GROUP_MAX = 16;                  // (two layers' worth of weight gradients: 2 x 6 for a decoder layer)
struct GroupItem {
  const void* A; const void* B; float* ws;
  void* out; int64_t ldo; float alpha;     // out != nullptr: ONE K-slice, accumulated straight onto out (16-bit) in the epilogue
  int64_t lda, ldb;
  int M, N, K, krows, ksplit, splits, tiles_m, tiles_n, first;   // K: rounded up to whole K tiles, krows: the real row count; first: index of the item's first workgroup
};
struct GroupArgs { GroupItem it[GROUP_MAX]; int n, total; };

// The product, tile and K-slice of this workgroup of a grouped launch, and the product's argument block.  XCD chunks over the
// flattened (item, slice, tile) order: the tiles of one slice of one product share operand panels in L2 -- which these products
// depend on (a 256 x 256 tile of a 13312-row contraction reads 13.6 MB; unshared, 216 of them are 2.9 GB in ~320 us).  The round-5
// experiment of giving every XCD 1 / 8 of EVERY product (equal work per XCD whatever the mix of contraction lengths) ran a decoder
// pair 2.7 x slower and an encoder pair 20 % slower for exactly that reason; mixed-length groups are avoided by the caller instead
// (ops._Wgrads flushes when the row count changes).
__device__ __forceinline__ const GroupItem* group_enter(const GroupArgs& ga, GemmArgs& g, int& t, int& ks) {
  const int id = xcd_remap((int)blockIdx.x, ga.total);
  int p = 0;
  for (int q = 1; q < ga.n; ++q)
    if (id >= ga.it[q].first) p = q;
  const GroupItem& it = ga.it[p];
  g.A = it.A; g.B = it.B; g.C = it.out; g.bias = nullptr;
  g.M = it.M; g.N = it.N; g.K = it.K; g.transA = 1; g.transB = 0;
  g.lda = it.lda; g.ldb = it.ldb; g.ldc = it.ldo; g.strideA = g.strideB = g.strideC = 0;
  g.alpha = it.out ? it.alpha : 1.f; g.flags = it.out ? OFA_GEMM_ACCUM : 0; g.batch_inner = 1; g.strideA2 = g.strideB2 = g.strideC2 = 0; g.a_krows = g.b_krows = it.krows;
  g.colstat = nullptr; g.colstat_rows = 0;
  const int ntiles = it.tiles_m * it.tiles_n, local = id - it.first;
  ks = local / ntiles;
  t = local - ks * ntiles;
  return &it;
}

bool gemm_group_pp_launch(int variant, const GroupArgs& ga, bool f16, hipStream_t st);   // gemm_pp.hip

}  // namespace ofa
