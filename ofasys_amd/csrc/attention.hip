// Fused multi-head attention for gfx950 (bf16 in/out, fp32 softmax, head_dim 64), forward + backward.
//
// Reference arithmetic: the slow path of MultiheadAttention.forward, module/multihead_attention.py:218-346
// (q*scaling, bmm(q,k^T), += attn_bias, causal -inf mask, key-padding -inf, softmax(dtype=fp32), bmm(p,v),
// per-head c_attn scale) and, with bias == NULL and scale = head_dim^-0.5, the F.multi_head_attention_forward
// fast path (:155-186).  The [B*A,T,S] score/probability tensors are never written to HBM.
//
// CDNA4 mapping.  One 64-lane wavefront owns 32 query rows (32 key rows in the dK/dV kernel); a workgroup is 4 waves
// (128 rows) and walks the OTHER sequence in blocks of 32 rows.  Every product is a v_mfma_f32_32x32x16_bf16:
//     S^T = K Q^T        (contract d)    A <- K tile rows,   B <- Q rows (registers, loaded once)
//     O^T = V^T P^T      (contract key)  A <- V tile, transposed on read,  B <- P^T straight from the S^T accumulator
//                                         registers (C-layout rows = keys = MFMA k-slots)
//   backward:
//     dP^T = V dO^T, dQ^T = K^T dS^T                        (dQ kernel, lanes <-> queries)
//     S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS  (dK/dV kernel, lanes <-> keys)
// The "swapped" orientation keeps each softmax row inside one lane (+ one cross-half exchange), so the online
// max/sum needs no 32-lane butterfly.  The k-slot permutation used for the key contraction
// (slot (hi,e) <-> key 16j + 8*(e>>2) + 4*hi + (e&3)) is applied identically to both operands, so the accumulator
// registers feed the next MFMA without any cross-lane traffic.
//
// Data movement.  The first generation of these kernels let every wave pull its K / V rows straight from L2: that
// reads each byte four times through the CU's vector-memory pipe (64 B/clk, the measured bound: 148 TF forward,
// 101 TF backward) and needed key-contiguous COPIES of V, K, Q and dO in HBM for the products that contract over the
// sequence.  Now each 32-row x 64-col tile of the other sequence is staged ONCE per workgroup through LDS:
//   * LDS-DMA (global_load_lds_dwordx4), double buffered: the next block's two 4 KiB tiles land while this block computes;
//   * row-major tiles, 16-byte chunks XOR-swizzled by ((row>>1)&7) on the source address (the DMA writes lane-linear);
//   * operands whose contraction index is the tile ROW (V^T P^T, K^T dS^T, dO^T P, Q^T dS) are fetched with
//     ds_read_b64_tr_b16, the CDNA4 transposing LDS read -- no transposed copy exists anywhere;
//   * operands contracting over head_dim are plain ds_read_b128 of the same tiles.
// Fragment reads are inline asm (an in-flight LDS-DMA makes hipcc put vmcnt(0) in front of every visible ds_read).
// MFMA-bound in principle; algorithmic flops 4*B*heads*T*S*64 forward, 2.5x that backward.
#include "common.h"

namespace ofa {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int TILE_BYTES = 32 * HD * 2;   // 4 KiB: 32 rows x 64 bf16

struct AttnL {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dout; const bf16_t* bias; const uint8_t* kpm;
  const void* c_attn; int c_dt; bf16_t* out; float* lse; float* delta;   // c_attn: [heads] in the dtype code c_dt
  bf16_t* dq; bf16_t* dk; bf16_t* dv; bf16_t* dbias;
  int B, heads, T, S, Tpad;
  int64_t ldq, ldk, ldo;
  float scale; int causal;
  const int* seg;   // ragged ("packed rows") mode: int32 [B][4] = {q_off, q_len, k_off, k_len}, see seg_enter()
  int rows_q, rows_k;   // ragged mode: total packed rows (filler rows between / behind the segments are zero-filled here)
  // additive bias addressing: bias[b * bias_bs + h * bias_hs + q * bias_ld + key].  The reference's tensor is [B*A, T, S] dense
  // (bias_ld = S, bias_hs = T*S, bias_bs = A*T*S); a POSITION bias is the same for every sample (positions do not depend on the
  // batch row), so it arrives once -- [A, Tb, Sb], bias_bs = 0 -- indexed by the position inside the sample (also in ragged mode)
  int64_t bias_ld, bias_hs, bias_bs;
  // the batch-shared position bias once more, in the two tile-swizzled images the forward / dQ kernels (bias_sr: a lane's 16 keys
  // of its query row) and the dK/dV kernel (bias_sc: a lane's 16 query rows of its key) read with two 16-byte loads per lane and
  // 32 x 32 block: see BiasSwz / csrc/bias.hip bias_build_kernel.  bias_nqt / bias_nkt: 32-row / 32-column tiles per head.
  const bf16_t* bias_sr; const bf16_t* bias_sc;
  int bias_nqt, bias_nkt;
  // backward only, optional: fp32 partial rows of the COLUMN SUMS of dq / dk / dv (= the bias gradients of the q / k / v projections,
  // multihead_attention.py:199-217) and of sum_t delta / c (= the gradient of c_attn, :342-345).  One partial row per (sample, 128-row
  // tile, wave): wave w of workgroup (x, b * heads + h) of the dQ kernel writes columns [64 h, 64 h + 64) of row (b * nqt + x) * 4 + w of
  // cs_q (row stride cs_ldq) and element ((b * nqt + x) * 4 + w) * heads + h of cs_c; the dK/dV kernel likewise rows
  // (b * nkt + x) * 4 + w of cs_k / cs_v (row stride cs_ldk).  Every partial row is written by every call (empty tiles of a ragged batch
  // write zeros); ofa_fold_batched sums them.
  float* cs_q; float* cs_k; float* cs_v; float* cs_c;
  int64_t cs_ldq, cs_ldk;
};

// Ragged mode: the rows between sample b's last row and sample b+1's first (alignment filler, and behind the last sample the
// bucket filler up to `rows`) belong to no segment.  They must hold finite values -- they flow through the row-wise kernels
// downstream, where 0 * NaN would poison the weight gradients -- so the workgroups of ONE extra grid column zero them: 64
// columns (head h) of rows [off + len, next_off) of up to two outputs.  Replaces a memset of the whole output per call.
__device__ __forceinline__ void seg_zero_fill(const int* seg, int B, int b, int h, bool k_side, int rows, bf16_t* o1, int64_t ld1,
                                              bf16_t* o2, int64_t ld2, int tid, float* stat_h = nullptr) {
  const int4 s = reinterpret_cast<const int4*>(seg)[b];
  const int lo = k_side ? s.z + s.w : s.x + s.y;
  int hi = rows;
  if (b + 1 < B) {
    const int4 n = reinterpret_cast<const int4*>(seg)[b + 1];
    hi = k_side ? n.z : n.x;
  }
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < (hi - lo) * 8; i += 256) {                 // 8 x 16-byte chunks per (row, head)
    const int r = lo + (i >> 3), c = (i & 7) * 8;
    if (o1) *reinterpret_cast<uint4*>(o1 + (int64_t)r * ld1 + h * HD + c) = z;
    if (o2) *reinterpret_cast<uint4*>(o2 + (int64_t)r * ld2 + h * HD + c) = z;
  }
  if (stat_h) {                                                     // a per-row fp32 statistic of head h (delta): same filler rows
    for (int r = lo + tid; r < hi; r += 256) stat_h[r] = 0.f;
    if (b == 0)
      for (int r = tid; r < (k_side ? s.z : s.x); r += 256) stat_h[r] = 0.f;
  }
  if (b == 0) {                                                     // rows in front of the first segment (normally none)
    const int first = k_side ? s.z : s.x;
    for (int i = tid; i < first * 8; i += 256) {
      const int r = i >> 3, c = (i & 7) * 8;
      if (o1) *reinterpret_cast<uint4*>(o1 + (int64_t)r * ld1 + h * HD + c) = z;
      if (o2) *reinterpret_cast<uint4*>(o2 + (int64_t)r * ld2 + h * HD + c) = z;
    }
  }
}

// Ragged mode.  The reference pads every sample of a batch to the longest one and masks (multihead_attention.py:319-326); the
// metric counts non-pad positions only, so here a batch may arrive PACKED: q / out / dout / dq are [rows_q, ld] with sample b's
// queries in rows q_off .. q_off + q_len - 1, k / v / dk / dv are [rows_k, ld] likewise, lse / delta are [heads, Tpad] indexed by
// the packed query row.  A workgroup (q-tile x, (b, h)) rebases every pointer to its sample and then runs the dense code with
// B = 1, T = q_len, S = k_len: keys beyond k_len are never loaded, tiles beyond q_len exit.  Returns false when the tile is empty.
__device__ __forceinline__ bool seg_enter(AttnL& a, int& b, int& bh, int h, int tile0, bool tile_is_query) {
  if (!a.seg) return true;
  const int4 s = reinterpret_cast<const int4*>(a.seg)[b];
  if (s.y <= 0 || s.w <= 0 || tile0 >= (tile_is_query ? s.y : s.w)) return false;
  a.q += (int64_t)s.x * a.ldq;
  if (a.out) a.out += (int64_t)s.x * a.ldo;
  if (a.dout) a.dout += (int64_t)s.x * a.ldo;
  if (a.dq) a.dq += (int64_t)s.x * a.ldq;
  if (a.lse) a.lse += s.x;
  if (a.delta) a.delta += s.x;
  a.k += (int64_t)s.z * a.ldk;
  a.v += (int64_t)s.z * a.ldk;
  if (a.dk) a.dk += (int64_t)s.z * a.ldk;
  if (a.dv) a.dv += (int64_t)s.z * a.ldk;
  a.T = s.y;
  a.S = s.w;
  b = 0;
  bh = h;
  return true;
}

__device__ __forceinline__ float head_scale(const AttnL& a, int h) {
  if (!a.c_attn) return 1.0f;
  return a.c_dt == OFA_BF16 ? bf2f(((const bf16_t*)a.c_attn)[h]) : a.c_dt == OFA_F16 ? (float)((const f16_t*)a.c_attn)[h] : ((const float*)a.c_attn)[h];
}
// The 16-bit storage type is a template flag (F16): bf16 (v_mfma_f32_32x32x16_bf16) or fp16 (v_mfma_f32_32x32x16_f16); tiles, swizzles and
// fragment reads are the same for both (bf16_t stands for "16-bit element" in the pointer types below).
template <bool F16> __device__ __forceinline__ uint32_t enc2(float lo, float hi) {
  if constexpr (F16) return pack_f16x2(lo, hi);
  else return pack_bf16x2(lo, hi);
}
template <bool F16> __device__ __forceinline__ float lo16(uint32_t w) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16x2_t, w)[0];
  else return __uint_as_float(w << 16);
}
template <bool F16> __device__ __forceinline__ float hi16(uint32_t w) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16x2_t, w)[1];
  else return __uint_as_float(w & 0xffff0000u);
}
template <bool F16> __device__ __forceinline__ float dec1(uint16_t u) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16_t, u);
  else return bf2f(u);
}
template <bool F16> __device__ __forceinline__ uint16_t enc1(float v) {
  if constexpr (F16) return __builtin_bit_cast(uint16_t, (f16_t)v);
  else return f2bf(v);
}
__device__ __forceinline__ bf16x8 ld16(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
template <bool F16>
__device__ __forceinline__ bf16x8 pack8f(const float* f) {
  union { uint4 u; bf16x8 v; } r;
  r.u.x = enc2<F16>(f[0], f[1]);
  r.u.y = enc2<F16>(f[2], f[3]);
  r.u.z = enc2<F16>(f[4], f[5]);
  r.u.w = enc2<F16>(f[6], f[7]);
  return r.v;
}
__device__ __forceinline__ void zero16f(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
__device__ __forceinline__ int crowl(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
template <bool F16>
__device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
  uint2 o;
  o.x = enc2<F16>(a, b);
  o.y = enc2<F16>(c, d);
  *reinterpret_cast<uint2*>(p) = o;
}
// "dead key" flag of this lane's key in a 32-key block (out of range or padded).  The byte load is issued one block
// AHEAD (before that block's DMA) and turned into a wave-uniform 32-bit mask with a ballot when the block is consumed,
// so its latency hides under the previous block's compute.
__device__ __forceinline__ int dead_flag(const uint8_t* kp, int key0, int S, int i) {
  const int key = key0 + i;
  int dead = key >= S;
  if (kp && !dead) dead = kp[key];
  return dead;
}
__device__ __forceinline__ uint32_t dead_ballot(int flag) { return (uint32_t)__ballot(flag != 0); }

// The additive bias is fetched ONE BLOCK AHEAD (raw, in front of that block's LDS-DMA) and decoded when the block is
// consumed: loads issued inside the block would sit behind the in-flight DMA in the vmcnt queue and drain it (the biased
// kernels ran 2-4.5x slower than the bias-free ones that way).  Row form (forward / dQ: 16 keys of one query row, 4 x 8-byte
// loads when S % 4 == 0 and the block is whole) and column form (dK/dV: one key, 16 query rows, 16 x 2-byte loads).
struct BiasRow {
  uint2 raw[4];
  bool vec;
  __device__ __forceinline__ void issue(const bf16_t* brow, int key0, int S, int hi, int64_t ld) {
    vec = brow && (ld & 3) == 0 && key0 + 32 <= S;
    if (vec) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) raw[g4] = *reinterpret_cast<const uint2*>(brow + key0 + 8 * g4 + 4 * hi);
    }
  }
};
template <bool F16> __device__ __forceinline__ void bias16(const bf16_t* brow, int key0, int S, int hi, float (&b)[16], bool aligned = true);
template <bool F16>
__device__ __forceinline__ void bias_row_take(const BiasRow& p, const bf16_t* brow, int key0, int S, int hi, float (&b)[16]) {
  if (!brow) return;
  if (p.vec) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      b[4 * g4] = lo16<F16>(p.raw[g4].x) * LOG2E;
      b[4 * g4 + 1] = hi16<F16>(p.raw[g4].x) * LOG2E;
      b[4 * g4 + 2] = lo16<F16>(p.raw[g4].y) * LOG2E;
      b[4 * g4 + 3] = hi16<F16>(p.raw[g4].y) * LOG2E;
    }
  } else {
    bias16<F16>(brow, key0, S, hi, b, false);     // ragged tail block / unaligned rows: in place, element by element
  }
}
struct BiasCol {
  unsigned short raw[16];
  __device__ __forceinline__ void issue(const bf16_t* bcol, int q0, int T, int64_t S, int hi) {
    if (!bcol) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + crowl(r, hi);
      raw[r] = q < T ? bcol[(int64_t)q * S] : (unsigned short)0;
    }
  }
  template <bool F16> __device__ __forceinline__ void take(const bf16_t* bcol, float (&b)[16]) const {
    if (!bcol) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] = dec1<F16>(raw[r]) * LOG2E;
  }
};

// BIAS mode 2 -- the batch-shared position bias in a tile-swizzled image: the 16 values lane l = (hi << 5 | i) needs for the 32 x 32
// block (qt, kt) are 32 contiguous bytes at ((qt * nkt + kt) * 64 + l) * 16 (row image; the column image swaps the roles), written
// once per layer by bias_build_kernel.  A wave fetches a block's bias as 2 KiB of consecutive bytes (the row-major tensor cost it
// 32 cache lines per load instruction), and the values go in as the INITIAL ACCUMULATOR of the score MFMAs -- times 1 / scale, the
// scale rides in the exponent's multiply-add as in the bias-free kernels -- so a biased block runs the bias-free arithmetic.
struct BiasSwz {
  uint4 raw[2];
  __device__ __forceinline__ void issue(const bf16_t* lane_base, int blk) {
    const uint4* p = reinterpret_cast<const uint4*>(lane_base + (int64_t)blk * 1024);
    raw[0] = p[0];
    raw[1] = p[1];
  }
  template <bool F16> __device__ __forceinline__ void take(float inv_scale, float (&b)[16]) const {
    const uint32_t w[8] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      b[2 * j] = lo16<F16>(w[j]) * inv_scale;
      b[2 * j + 1] = hi16<F16>(w[j]) * inv_scale;
    }
  }
};
__device__ __forceinline__ void init16f(f32x16& a, const float (&b)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = b[r];
}
// BIAS mode 3 -- the same image, but a block's 2 KiB travel to LDS by DMA next to the block's K / V (Q / dO) tiles instead of waiting
// in 8 registers per stage: the backward kernels sit at their register limits (dK/dV spilled 16 registers with mode 2, dQ could
// not run three waves per SIMD).  Per stage and wave: [2][1 KiB] (the lanes' first / second 16 bytes), read back with two
// ds_read_b128 right where the values are needed.
constexpr int BIAS_STAGE_BYTES = 4 * 2048;                  // four waves x 2 KiB
__device__ __forceinline__ void bias_dma(const bf16_t* lane_base, int blk, unsigned char* lds_wave_stage) {
  const bf16_t* p = lane_base + (int64_t)blk * 1024;
  __builtin_amdgcn_global_load_lds((gvoid_t*)p, (lvoid_t*)lds_wave_stage, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((gvoid_t*)(p + 8), (lvoid_t*)(lds_wave_stage + 1024), 16, 0, 0);
}
template <bool F16> __device__ __forceinline__ void bias_words(const u64x2& b0, const u64x2& b1, float inv_scale, float (&b)[16]) {
  const uint4 u0 = __builtin_bit_cast(uint4, b0), u1 = __builtin_bit_cast(uint4, b1);
  const uint32_t w[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    b[2 * j] = lo16<F16>(w[j]) * inv_scale;
    b[2 * j + 1] = hi16<F16>(w[j]) * inv_scale;
  }
}

// additive bias of the 16 scores a lane holds for a 32-key block (keys key0 + crowl(r, hi)), pre-multiplied by log2(e)
template <bool F16>
__device__ __forceinline__ void bias16(const bf16_t* brow, int key0, int S, int hi, float (&b)[16], bool aligned) {
  if (aligned && (S & 3) == 0 && key0 + 32 <= S) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const uint2 u = *reinterpret_cast<const uint2*>(brow + key0 + 8 * g4 + 4 * hi);
      b[4 * g4] = lo16<F16>(u.x) * LOG2E;
      b[4 * g4 + 1] = hi16<F16>(u.x) * LOG2E;
      b[4 * g4 + 2] = lo16<F16>(u.y) * LOG2E;
      b[4 * g4 + 3] = hi16<F16>(u.y) * LOG2E;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + crowl(r, hi);
      b[r] = key < S ? dec1<F16>(brow[key]) * LOG2E : 0.f;
    }
  }
}

// Chunk swizzle of the [32 x 64] LDS tile images: chunk c (16 bytes, 8 per 128-byte row) of row r lives at c ^ tile_swz(r),
// tile_swz(r) = ((r>>1)&7) rotated right by one bit.  Any bijection of (r>>1)&7 makes the 16 rows of a ds_read_b128 lane group hit
// 16 distinct 16-byte slots; the rotation is for the TRANSPOSING reads: a 32-lane pass of ds_read_b64_tr_b16 covers rows
// base + 0..3 x 64 bytes, rows {0,1} and {2,3} differ in bit 0 of (r>>1), and that bit must move the chunk to the OTHER aligned
// group of four (xor 4) -- with the plain (r>>1)&7 it only xors chunk bit 0, rows 2,3 land on the slots of rows 0,1 and every
// pass is a 2-way bank conflict (round-1 PMC: SQ_LDS_BANK_CONFLICT = 25-33% of SQ_LDS_IDX_ACTIVE in all three kernels).
// (Measured: the conflicts were not what bounds these kernels -- same times before and after, profiles/round2_attention_experiments.txt.)
__device__ __forceinline__ int tile_swz(int row) {
  const int x = (row >> 1) & 7;
  return ((x & 1) << 2) | (x >> 1);
}

// One workgroup-wide DMA of a [32 x 64] tile: rows row0.. of a [*, ld] matrix, columns col0..col0+63.
// 256 granules of 16 B, one per thread; LDS image row-major 128-B rows, chunk c of row r stored at c ^ tile_swz(r).
__device__ __forceinline__ void tile_dma(const bf16_t* __restrict__ base, int64_t ld, int row0, int rmax, int col0,
                                         bf16_t* __restrict__ lds_tile, int tid, int wave_u) {
  const int r = tid >> 3, pc = tid & 7;
  const int c = pc ^ tile_swz(r);
  int rr = row0 + r;
  rr = rr < rmax ? rr : rmax - 1;
  const bf16_t* src = base + (int64_t)rr * ld + col0 + c * 8;
  __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(lds_tile + wave_u * 64 * 8), 16, 0, 0);
}

// per-lane byte offsets inside a tile for the two read flavours
struct TileAddr {
  uint32_t km[4];   // ds_read_b128: row (lane&31), 16-byte chunk kk*2 + (lane>>5), kk = 0..3
  uint32_t tr[2];   // ds_read_b64_tr_b16, output-column block dt (0/1), rows 16j + 4hi + 0..3   (immediate adds 16j rows)
  uint32_t trx[2];  // same for rows 16j + 8 + 4hi + 0..3: the +8 flips bit 2 of the row swizzle (immediate adds 16j + 8 rows)
  __device__ __forceinline__ void init(uint32_t lds0, int lane) {
    const int row = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) km[kk] = lds0 + (uint32_t)(row * 128 + (((kk * 2 + hi) ^ tile_swz(row)) << 4));
    // transposing read: lane q' of a 16-lane group supplies the address of 4 contiguous elements (row kb + (q'>>2),
    // col cb + 4*(q'&3)) and receives column cb + q' of the 4-row block.  Row swizzle ((row>>1)&7) of
    // row = 16j + 8m + 4hi + t is 4m + 2hi + (t>>1): independent of j, bit 2 set by m.
    const int g = lane >> 4, q = lane & 15;
    const int trow = 4 * (g >> 1) + (q >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int col = dt * 32 + (g & 1) * 16 + 4 * (q & 3);
      const int sw = tile_swz(trow);          // trow < 8; the immediates add 16j rows (swizzle unchanged) ...
      tr[dt] = lds0 + (uint32_t)(trow * 128 + (((col >> 3) ^ sw) << 4) + (col & 7) * 2);
      trx[dt] = lds0 + (uint32_t)(trow * 128 + (((col >> 3) ^ sw ^ 2) << 4) + (col & 7) * 2);   // ... or 16j + 8: bit 2 of (r>>1) -> xor 2
    }
  }
};

// fragment reads (asm, not waited here)
template <int OFF>
__device__ __forceinline__ void rd128(u64x2& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
// transposed 8-element fragment for key/query sub-block j (16 rows) of a tile at byte offset TOFF:
//   elements 0..3 <- rows 16j + 4hi + 0..3, elements 4..7 <- rows 16j + 8 + 4hi + 0..3 (the k-slot permutation of the
//   accumulator layout).  Rows 16j + {0..7} have swizzle bit2 = ((row>>1)&4) = 0 for even m... handled via XOR64 flag.
template <int TOFF, int J>
__device__ __forceinline__ void rdtr(u64x2& d, uint32_t addr_lo, uint32_t addr_hi) {
  unsigned long long lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr_lo), "i"(TOFF + J * 16 * 128));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr_hi), "i"(TOFF + J * 16 * 128 + 8 * 128));
  d[0] = lo;
  d[1] = hi;
}
// stage barrier: this wave's LDS-DMA pieces have landed (explicit vmcnt wait -- hipcc only inserts one in front of LDS reads it
// can see, and the fragment reads here are inline asm), then everybody else's
#define ATT_SYNC()                                     \
  do {                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
    __syncthreads();                                   \
  } while (0)
#define ATT_WAIT4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
template <bool F16, typename A, typename B, typename C>
__device__ __forceinline__ C att_mfma(const A& a, const B& b, const C& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#define ATT_MFMA(A, B, C) att_mfma<F16>(A, B, C)

// Phase-timestamp probe of tools/attn_timeline.py (measurement build only: -DOFA_ATTN_TIMELINE is never set by the Makefile; in
// the product the macros are empty).  Wave 0 of each forward workgroup records s_memtime at entry / first tiles landed / exit and,
// inside its fifth key block, at every phase boundary of fwd_block.
#ifdef OFA_ATTN_TIMELINE
__device__ unsigned long long g_attn_tl[8192 * 16];
#define ATT_FS(i, dep) do { asm volatile("" : "+v"(dep)); if (fs) fs[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define ATT_FS_ARG , unsigned long long* fs
#define ATT_FS_PASS(cond) , ((cond) ? fsv : nullptr)
#else
#define ATT_FS(i, dep) do { } while (0)
#define ATT_FS_ARG
#define ATT_FS_PASS(cond)
#endif

// max / sum of a value with its partner lane in the other 32-lane half: one v_permlane32_swap (gfx950) instead of a
// ds_bpermute round trip through the LDS crossbar (tools/attn_timeline.py: 160 + 136 clocks per key block for the two exchanges
// of the online softmax).  swap(a, b) exchanges a's upper half with b's lower half: with a = b = v the pair (a, b) then holds
// {v[lane], v[lane ^ 32]} in some order in EVERY lane.
__device__ __forceinline__ float xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}


// Column sums of the 32 x 64 block a wave holds in two accumulator tiles (lane (i, hi): row i of the block, columns
// 32 dt + 8 (r >> 2) + 4 hi + (r & 3) for register r of tile dt), over the 32 rows, of the fp32 accumulators times `mul` (NOT re-rounded to
// the 16-bit storage type: the sums differ from the column sums of the stored tensor by its rounding noise, ~2^-9 / sqrt(rows) relative --
// emulating the rounding cost a third of this epilogue's instructions in kernels that are bound by VALU issue; rows with !row_ok count
// as zero).  A transposing butterfly over the 32 lanes of each half: at every level a lane keeps the half of
// its values selected by one bit of its lane number and adds the partner's copy of them -- 16 + 8 + 4 + 2 + 1 = 31 exchanges instead of
// 32 x 5.  The levels run over lane bits 3, 2, 1, 0 as DPP operands of the add (row_mirror, row_half_mirror, quad_perm: partners that flip
// the level's bit -- and lower ones, which later levels do not mind) and only the last, single exchange (bit 4: across 16-lane rows)
// is a shuffle through the LDS crossbar: the first form (five ds_bpermute levels, then a workgroup-wide fold through LDS behind two
// barriers) cost ~1 us per workgroup and sum (tools r6 attn_cs_bench: dK/dV at 32 x 12 x 448^2 +2.9 us per sum), as much as the column-sum
// launches it replaced.  Fixed order: deterministic.  Lane l ends up with the sum of column col64(l).
__device__ __forceinline__ int col64(int lane) {
  const int i = lane & 31, hi = lane >> 5;
  const int vidx = ((i & 15) << 1) | (i >> 4), dt = vidx >> 4, r = vidx & 15;
  return dt * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
}
template <int M, int CTRL> __device__ __forceinline__ void colsum_level(float (&v)[32], int lane, int bit) {
  const bool up = (lane & bit) != 0;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const float keep = up ? v[j + M] : v[j], give = up ? v[j] : v[j + M];
    float got;
    if constexpr (CTRL != 0) got = dpp_f<CTRL, 0xf>(0.f, give);
    else got = __shfl_xor(give, 16, 64);
    v[j] = keep + got;
  }
}
template <bool F16>
__device__ __forceinline__ float wave_colsum64(const f32x16 (&acc)[2], float mul, bool row_ok, int lane) {
  float v[32];
  const float m = row_ok ? mul : 0.f;                      // (one multiply per value is the scale AND the row mask)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[dt * 16 + r] = acc[dt][r] * m;
  colsum_level<16, 0x140>(v, lane, 8);     // row_mirror:      lane i <-> 15 - i of its row of 16
  colsum_level<8, 0x141>(v, lane, 4);      // row_half_mirror: i <-> 7 - i of its 8
  colsum_level<4, 0x4E>(v, lane, 2);       // quad_perm [2,3,0,1]
  colsum_level<2, 0xB1>(v, lane, 1);       // quad_perm [1,0,3,2]
  colsum_level<1, 0>(v, lane, 16);         // lane i <-> i ^ 16
  return v[0];
}

// ------------------------------------------------------------------------------------------------ forward
template <int BUF, int BIAS, bool F16>   // BUF selects the double-buffer half at compile time (immediates); BIAS: 0 none, 1 dense, 2 swizzled
__device__ __forceinline__ void fwd_block(const AttnL& a, const TileAddr& ta, const uint32_t* trx, const bf16x8 (&qf)[4],
                                          f32x16 (&ot)[2], float& m_run, float& l_run, int key0, int q0, int qi, int hi,
                                          uint32_t dead_now, const float (&bz)[16], float sc ATT_FS_ARG) {
  constexpr int KOFF = BUF * 2 * TILE_BYTES, VOFF = KOFF + TILE_BYTES;
  constexpr bool has_bias = BIAS == 1;
  ATT_FS(0, m_run);
  u64x2 kf[4];
  rd128<KOFF>(kf[0], ta.km[0]);
  rd128<KOFF>(kf[1], ta.km[1]);
  rd128<KOFF>(kf[2], ta.km[2]);
  rd128<KOFF>(kf[3], ta.km[3]);
  ATT_WAIT4(kf[0], kf[1], kf[2], kf[3]);
  ATT_FS(1, m_run);
  u64x2 vf[2][2];   // [j][dt]
  rdtr<VOFF, 0>(vf[0][0], ta.tr[0], trx[0]);
  rdtr<VOFF, 0>(vf[0][1], ta.tr[1], trx[1]);
  rdtr<VOFF, 1>(vf[1][0], ta.tr[0], trx[0]);
  rdtr<VOFF, 1>(vf[1][1], ta.tr[1], trx[1]);
  f32x16 st;
  if constexpr (BIAS == 2) init16f(st, bz);
  else zero16f(st);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) st = ATT_MFMA(kf[kk], qf[kk], st);
  float s[16];
  float mx = -INFINITY;
  const bool diag = a.causal && (key0 + 31 > q0);
  const bool general = has_bias || dead_now || diag;
  if (general) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = crowl(r, hi), key = key0 + j;
      float t = st[r] * sc;
      if (has_bias) t += bz[r];
      bool dead = (dead_now >> j) & 1u;
      if (diag) dead |= key > qi;
      t = dead ? -INFINITY : t;
      s[r] = t;
      mx = fmaxf(mx, t);
    }
  } else {                                 // whole block alive: the row max is taken on the raw scores (sc > 0), the scale rides
#pragma unroll                             // in the exponent's multiply-add
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
    mx *= sc;
  }
  ATT_FS(2, mx);
  mx = xhalf_max(mx);
  ATT_FS(3, mx);
  const float m_new = fmaxf(m_run, mx);
  const float m_use = m_new == -INFINITY ? 0.f : m_new;
  float p[16];
  float ps = 0.f;
  if (general) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
      ps += p[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], sc, -m_use));
      ps += p[r];
    }
  }
  ATT_FS(4, ps);
  ps = xhalf_sum(ps);
  ATT_FS(5, ps);
  if (__any(m_new != m_run)) {
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
  ATT_WAIT4(vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
  ATT_FS(6, l_run);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16x8 pf = pack8f<F16>(p + 8 * j);
    ot[0] = ATT_MFMA(vf[j][0], pf, ot[0]);
    ot[1] = ATT_MFMA(vf[j][1], pf, ot[1]);
  }
  ATT_FS(7, l_run);
}

template <int BIAS, bool F16>
__device__ __forceinline__ void attn_fwd_body(AttnL a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);         // [2 buffers][K tile | V tile], 4 KiB each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int i = lane & 31, hi = lane >> 5;
  int bh = blockIdx.y, b = bh / a.heads;
  const int h = bh % a.heads;
  const int qb0 = blockIdx.x * 128;
  if (a.seg && blockIdx.x == gridDim.x - 1) {            // ragged mode: the extra grid column zero-fills the filler rows
    seg_zero_fill(a.seg, a.B, b, h, false, a.rows_q, a.out, a.ldo, nullptr, 0, tid);
    return;
  }
#ifdef OFA_ATTN_TIMELINE
  unsigned long long fsv[16];
  for (int z = 0; z < 16; ++z) fsv[z] = 0;
  fsv[8] = __builtin_amdgcn_s_memtime();
  fsv[12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  fsv[13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  fsv[14] = __builtin_amdgcn_s_memrealtime();
  const int tl_wg = blockIdx.y * gridDim.x + blockIdx.x;
#endif
  if (!seg_enter(a, b, bh, h, qb0, true)) return;        // (workgroup-uniform)
  const int q0 = qb0 + wave * 32;
  const int qi = q0 + i;
  const int qrow = qi < a.T ? qi : a.T - 1;
  const bf16_t* qp = a.q + ((int64_t)b * a.T + qrow) * a.ldq + h * HD + hi * 8;
  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = ld16(qp + kk * 16);
  const bf16_t* kbase = a.k + (int64_t)b * a.S * a.ldk;
  const bf16_t* vbase = a.v + (int64_t)b * a.S * a.ldk;
  const bf16_t* brow = BIAS == 1 ? a.bias + (int64_t)(blockIdx.y / a.heads) * a.bias_bs + (int64_t)h * a.bias_hs + (int64_t)qrow * a.bias_ld : nullptr;
  const bf16_t* bswz = nullptr;                          // BIAS 2: this lane's 16 values of block (q0 / 32, 0) of head h
  if constexpr (BIAS == 2) {
    const int qt = (q0 >> 5) < a.bias_nqt ? (q0 >> 5) : a.bias_nqt - 1;
    bswz = a.bias_sr + (((int64_t)h * a.bias_nqt + qt) * a.bias_nkt * 64 + lane) * 16;
  }
  const float inv_scale = 1.0f / a.scale;
  const uint8_t* kp = a.kpm ? a.kpm + (int64_t)b * a.S : nullptr;
  const float sc = a.scale * LOG2E;
  TileAddr ta;
  ta.init((uint32_t)(uintptr_t)smem, lane);
  const uint32_t* trx = ta.trx;

  f32x16 ot[2];
  zero16f(ot[0]);
  zero16f(ot[1]);
  float m_run = -INFINITY, l_run = 0.f;
  int nkb = (a.S + 31) / 32;
  if (a.causal) {                                        // the workgroup walks as far as its LAST wave needs
    const int qlast = qb0 + 127 < a.T - 1 ? qb0 + 127 : a.T - 1;
    const int lim = (qlast < a.S - 1 ? qlast : a.S - 1) / 32 + 1;
    nkb = lim < nkb ? lim : nkb;
  }
  int kflag = dead_flag(kp, 0, a.S, i);
  BiasRow bpre;
  BiasSwz spre;
  if constexpr (BIAS == 1) bpre.issue(brow, 0, a.S, hi, a.bias_ld);
  if constexpr (BIAS == 2) spre.issue(bswz, 0);
  tile_dma(kbase, a.ldk, 0, a.S, h * HD, lds, tid, wave_u);
  tile_dma(vbase, a.ldk, 0, a.S, h * HD, lds + TILE_BYTES / 2, tid, wave_u);
  ATT_SYNC();
#ifdef OFA_ATTN_TIMELINE
  fsv[9] = __builtin_amdgcn_s_memtime();
#endif
  const bool live_wave = q0 < a.T;
  float bz[16];
  for (int kb = 0; kb < nkb; kb += 2) {
    {
      const uint32_t dead_now = dead_ballot(kflag);
      const BiasRow bcur = bpre;
      const BiasSwz scur = spre;
      if (kb + 1 < nkb) {
        kflag = dead_flag(kp, (kb + 1) * 32, a.S, i);
        if constexpr (BIAS == 1) bpre.issue(brow, (kb + 1) * 32, a.S, hi, a.bias_ld);
        if constexpr (BIAS == 2) spre.issue(bswz, kb + 1);
        tile_dma(kbase, a.ldk, (kb + 1) * 32, a.S, h * HD, lds + TILE_BYTES, tid, wave_u);
        tile_dma(vbase, a.ldk, (kb + 1) * 32, a.S, h * HD, lds + TILE_BYTES + TILE_BYTES / 2, tid, wave_u);
      }
      const bool need = live_wave && !(a.causal && kb * 32 > q0 + 31);
      if (need) {
        if constexpr (BIAS == 1) bias_row_take<F16>(bcur, brow, kb * 32, a.S, hi, bz);
        if constexpr (BIAS == 2) scur.template take<F16>(inv_scale, bz);
        fwd_block<0, BIAS, F16>(a, ta, trx, qf, ot, m_run, l_run, kb * 32, q0, qi, hi, dead_now, bz, sc ATT_FS_PASS(kb == 4 && wave_u == 0));
      }
      ATT_SYNC();
#ifdef OFA_ATTN_TIMELINE
      if (kb == 4) fsv[10] = __builtin_amdgcn_s_memtime();
#endif
    }
    if (kb + 1 < nkb) {
      const uint32_t dead_now = dead_ballot(kflag);
      const BiasRow bcur = bpre;
      const BiasSwz scur = spre;
      if (kb + 2 < nkb) {
        kflag = dead_flag(kp, (kb + 2) * 32, a.S, i);
        if constexpr (BIAS == 1) bpre.issue(brow, (kb + 2) * 32, a.S, hi, a.bias_ld);
        if constexpr (BIAS == 2) spre.issue(bswz, kb + 2);
        tile_dma(kbase, a.ldk, (kb + 2) * 32, a.S, h * HD, lds, tid, wave_u);
        tile_dma(vbase, a.ldk, (kb + 2) * 32, a.S, h * HD, lds + TILE_BYTES / 2, tid, wave_u);
      }
      const bool need = live_wave && !(a.causal && (kb + 1) * 32 > q0 + 31);
      if (need) {
        if constexpr (BIAS == 1) bias_row_take<F16>(bcur, brow, (kb + 1) * 32, a.S, hi, bz);
        if constexpr (BIAS == 2) scur.template take<F16>(inv_scale, bz);
        fwd_block<1, BIAS, F16>(a, ta, trx, qf, ot, m_run, l_run, (kb + 1) * 32, q0, qi, hi, dead_now, bz, sc ATT_FS_PASS(false));
      }
      ATT_SYNC();
    }
  }
  if (qi < a.T) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    const float c = head_scale(a, h) * inv;
    bf16_t* op = a.out + ((int64_t)b * a.T + qi) * a.ldo + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        st4<F16>(op + dt * 32 + 8 * qq + 4 * hi, ot[dt][4 * qq] * c, ot[dt][4 * qq + 1] * c, ot[dt][4 * qq + 2] * c,
            ot[dt][4 * qq + 3] * c);
    if (hi == 0 && a.lse) a.lse[(int64_t)bh * a.Tpad + qi] = (m_run == -INFINITY ? 0.f : m_run) + log2f(l_run > 0.f ? l_run : 1.f);
  }
#ifdef OFA_ATTN_TIMELINE
  fsv[11] = __builtin_amdgcn_s_memtime();
  fsv[15] = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && tl_wg < 8192)
    for (int z = 0; z < 16; ++z) g_attn_tl[tl_wg * 16 + z] = fsv[z];
#endif
}

// three waves per SIMD for the bias-free form (its registers fit 168); the biased one keeps two
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attn_fwd_lds_kernel(AttnL a) { attn_fwd_body<0, false>(a); }
__global__ __launch_bounds__(256) void attn_fwd_bias_lds_kernel(AttnL a) { attn_fwd_body<1, false>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attn_fwd_f16_lds_kernel(AttnL a) { attn_fwd_body<0, true>(a); }
__global__ __launch_bounds__(256) void attn_fwd_bias_f16_lds_kernel(AttnL a) { attn_fwd_body<1, true>(a); }
// the batch-shared position bias (swizzled image as the score accumulators' initial value): the bias-free arithmetic + two loads
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attn_fwd_sbias_lds_kernel(AttnL a) { attn_fwd_body<2, false>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attn_fwd_sbias_f16_lds_kernel(AttnL a) { attn_fwd_body<2, true>(a); }

// ------------------------------------------------------------------------------------------------ backward: dQ
template <int BUF, int BIAS, bool F16>
__device__ __forceinline__ void dq_block(const AttnL& a, const TileAddr& ta, const uint32_t* trx, const bf16x8 (&qf)[4],
                                         const bf16x8 (&dof)[4], f32x16 (&dqt)[2], int key0, int q0, int qi, int hi,
                                         uint32_t dead_now, const float (&bz)[16], bf16_t* dbrow, float sc, float lse_q,
                                         float delta_q, float c) {
  constexpr int KOFF = BUF * 2 * TILE_BYTES, VOFF = KOFF + TILE_BYTES;
  constexpr bool has_bias = BIAS == 1;
  u64x2 kf[4], vf[4];
  rd128<KOFF>(kf[0], ta.km[0]); rd128<KOFF>(kf[1], ta.km[1]); rd128<KOFF>(kf[2], ta.km[2]); rd128<KOFF>(kf[3], ta.km[3]);
  rd128<VOFF>(vf[0], ta.km[0]); rd128<VOFF>(vf[1], ta.km[1]); rd128<VOFF>(vf[2], ta.km[2]); rd128<VOFF>(vf[3], ta.km[3]);
  ATT_WAIT4(kf[0], kf[1], kf[2], kf[3]);
  ATT_WAIT4(vf[0], vf[1], vf[2], vf[3]);
  f32x16 st, dp;
  if constexpr (BIAS == 2) init16f(st, bz);
  else zero16f(st);
  zero16f(dp);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    st = ATT_MFMA(kf[kk], qf[kk], st);
    dp = ATT_MFMA(vf[kk], dof[kk], dp);
  }
  __builtin_amdgcn_sched_barrier(0);       // the transposed K reads go out behind the MFMAs, into the registers those have consumed
  u64x2 ktf[2][2];
  rdtr<KOFF, 0>(ktf[0][0], ta.tr[0], trx[0]);
  rdtr<KOFF, 0>(ktf[0][1], ta.tr[1], trx[1]);
  rdtr<KOFF, 1>(ktf[1][0], ta.tr[0], trx[0]);
  rdtr<KOFF, 1>(ktf[1][1], ta.tr[1], trx[1]);
  float ds[16];
  const bool diag = a.causal && (key0 + 31 > q0);
  if (has_bias || dead_now || diag) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = crowl(r, hi), key = key0 + j;
      float t = st[r] * sc;
      if (has_bias) t += bz[r];
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
  if (BIAS == 1 && dbrow && qi < a.T) {
    if ((a.S & 3) == 0 && key0 + 32 <= a.S) {             // 4 consecutive keys per register quad: 8-byte stores
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        st4<F16>(dbrow + key0 + 8 * g4 + 4 * hi, ds[4 * g4], ds[4 * g4 + 1], ds[4 * g4 + 2], ds[4 * g4 + 3]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + crowl(r, hi);
        if (key < a.S) dbrow[key] = enc1<F16>(ds[r]);
      }
    }
  }
  ATT_WAIT4(ktf[0][0], ktf[0][1], ktf[1][0], ktf[1][1]);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16x8 dsf = pack8f<F16>(ds + 8 * j);
    dqt[0] = ATT_MFMA(ktf[j][0], dsf, dqt[0]);
    dqt[1] = ATT_MFMA(ktf[j][1], dsf, dqt[1]);
  }
}

template <int BIAS, bool F16>
__device__ __forceinline__ void attn_bwd_dq_body(AttnL a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int i = lane & 31, hi = lane >> 5;
  int bh = blockIdx.y, b = bh / a.heads;
  const int h = bh % a.heads;
  const int qb0 = blockIdx.x * 128;
  if (a.seg && blockIdx.x == gridDim.x - 1) {
    seg_zero_fill(a.seg, a.B, b, h, false, a.rows_q, a.dq, a.ldq, nullptr, 0, tid, a.out ? a.delta + (int64_t)h * a.Tpad : nullptr);
    return;
  }
  // partial row of the column sums (cs_q) / of sum delta / c (cs_c) this workgroup owns: (sample, query tile), see AttnL.  (Computed
  // where it is used -- here for an empty tile, else behind the loop: nothing of it is live across the loop.)
  const int b_sample = __builtin_amdgcn_readfirstlane(b);      // (seg_enter rebases b to 0; a scalar register across the loop)
  auto cs_rows = [&](float*& csq, float*& csc) {              // this WAVE's partial row
    const int64_t slot = ((int64_t)b_sample * (gridDim.x - (a.seg ? 1 : 0)) + blockIdx.x) * 4 + wave_u;
    csq = a.cs_q ? a.cs_q + slot * a.cs_ldq + h * HD : nullptr;
    csc = a.cs_c ? a.cs_c + slot * a.heads + h : nullptr;
  };
  if (!seg_enter(a, b, bh, h, qb0, true)) {
    float *csq, *csc;
    cs_rows(csq, csc);
    if (csq) csq[lane] = 0.f;
    if (csc && lane == 0) *csc = 0.f;
    return;
  }
  const int q0 = qb0 + wave * 32;
  const int qi = q0 + i;
  const int qrow = qi < a.T ? qi : a.T - 1;
  const bf16_t* qp = a.q + ((int64_t)b * a.T + qrow) * a.ldq + h * HD + hi * 8;
  const bf16_t* dop = a.dout + ((int64_t)b * a.T + qrow) * a.ldo + h * HD + hi * 8;
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    qf[kk] = ld16(qp + kk * 16);
    dof[kk] = ld16(dop + kk * 16);
  }
  const float lse_q = a.lse[(int64_t)bh * a.Tpad + qrow];
  float delta_q;
  if (a.out) {
    // delta = rowsum(dO * O) of this lane's query row, from the dO fragments already in registers and the matching 32 values
    // of O (the other 32 live in the partner lane); written out for the dK/dV kernel and the c_attn gradient
    const bf16_t* outp = a.out + ((int64_t)b * a.T + qrow) * a.ldo + h * HD + hi * 8;
    float sdo = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 of = ld16(outp + kk * 16);
      const uint4 ow = __builtin_bit_cast(uint4, of), dw = __builtin_bit_cast(uint4, dof[kk]);
      sdo += lo16<F16>(ow.x) * lo16<F16>(dw.x) + hi16<F16>(ow.x) * hi16<F16>(dw.x);
      sdo += lo16<F16>(ow.y) * lo16<F16>(dw.y) + hi16<F16>(ow.y) * hi16<F16>(dw.y);
      sdo += lo16<F16>(ow.z) * lo16<F16>(dw.z) + hi16<F16>(ow.z) * hi16<F16>(dw.z);
      sdo += lo16<F16>(ow.w) * lo16<F16>(dw.w) + hi16<F16>(ow.w) * hi16<F16>(dw.w);
    }
    delta_q = xhalf_sum(sdo);
    if (hi == 0 && qi < a.T) a.delta[(int64_t)bh * a.Tpad + qi] = delta_q;
  } else {
    delta_q = a.delta[(int64_t)bh * a.Tpad + qrow];
  }
  const float c = head_scale(a, h);
  const bf16_t* kbase = a.k + (int64_t)b * a.S * a.ldk;
  const bf16_t* vbase = a.v + (int64_t)b * a.S * a.ldk;
  const bf16_t* brow = BIAS == 1 ? a.bias + (int64_t)(blockIdx.y / a.heads) * a.bias_bs + (int64_t)h * a.bias_hs + (int64_t)qrow * a.bias_ld : nullptr;
  bf16_t* dbrow = (BIAS == 1 && a.dbias) ? a.dbias + ((int64_t)bh * a.T + qrow) * a.S : nullptr;      // (dense [B*A, T, S] only)
  const bf16_t* bswz = nullptr;                          // BIAS 2: this lane's 16 values of block (q0 / 32, 0) of head h
  if constexpr (BIAS == 2) {
    const int qt = (q0 >> 5) < a.bias_nqt ? (q0 >> 5) : a.bias_nqt - 1;
    bswz = a.bias_sr + (((int64_t)h * a.bias_nqt + qt) * a.bias_nkt * 64 + lane) * 16;
  }
  const float inv_scale = 1.0f / a.scale;
  const uint8_t* kp = a.kpm ? a.kpm + (int64_t)b * a.S : nullptr;
  const float sc = a.scale * LOG2E;
  TileAddr ta;
  ta.init((uint32_t)(uintptr_t)smem, lane);
  const uint32_t* trx = ta.trx;

  f32x16 dqt[2];
  zero16f(dqt[0]);
  zero16f(dqt[1]);
  int nkb = (a.S + 31) / 32;
  const int nkb_all = nkb;
  if (a.causal) {
    const int qlast = qb0 + 127 < a.T - 1 ? qb0 + 127 : a.T - 1;
    const int lim = (qlast < a.S - 1 ? qlast : a.S - 1) / 32 + 1;
    nkb = lim < nkb ? lim : nkb;
  }
  int kflag = dead_flag(kp, 0, a.S, i);
  BiasRow bpre;
  BiasSwz spre;
  if constexpr (BIAS == 1) bpre.issue(brow, 0, a.S, hi, a.bias_ld);
  if constexpr (BIAS == 2) spre.issue(bswz, 0);
  tile_dma(kbase, a.ldk, 0, a.S, h * HD, lds, tid, wave_u);
  tile_dma(vbase, a.ldk, 0, a.S, h * HD, lds + TILE_BYTES / 2, tid, wave_u);
  ATT_SYNC();
  const bool live_wave = q0 < a.T;
  float bz[16];
  int my_last = -1;      // last key block this wave actually visited (for zero-filling dbias beyond it)
  for (int kb = 0; kb < nkb; kb += 2) {
    {
      const uint32_t dead_now = dead_ballot(kflag);
      const BiasRow bcur = bpre;
      const BiasSwz scur = spre;
      if (kb + 1 < nkb) {
        kflag = dead_flag(kp, (kb + 1) * 32, a.S, i);
        if constexpr (BIAS == 1) bpre.issue(brow, (kb + 1) * 32, a.S, hi, a.bias_ld);
        if constexpr (BIAS == 2) spre.issue(bswz, kb + 1);
        tile_dma(kbase, a.ldk, (kb + 1) * 32, a.S, h * HD, lds + TILE_BYTES, tid, wave_u);
        tile_dma(vbase, a.ldk, (kb + 1) * 32, a.S, h * HD, lds + TILE_BYTES + TILE_BYTES / 2, tid, wave_u);
      }
      const bool need = live_wave && !(a.causal && kb * 32 > q0 + 31);
      if (need) {
        if constexpr (BIAS == 1) bias_row_take<F16>(bcur, brow, kb * 32, a.S, hi, bz);
        if constexpr (BIAS == 2) scur.template take<F16>(inv_scale, bz);
        dq_block<0, BIAS, F16>(a, ta, trx, qf, dof, dqt, kb * 32, q0, qi, hi, dead_now, bz, dbrow, sc, lse_q, delta_q, c);
        my_last = kb;
      }
      ATT_SYNC();
    }
    if (kb + 1 < nkb) {
      const uint32_t dead_now = dead_ballot(kflag);
      const BiasRow bcur = bpre;
      const BiasSwz scur = spre;
      if (kb + 2 < nkb) {
        kflag = dead_flag(kp, (kb + 2) * 32, a.S, i);
        if constexpr (BIAS == 1) bpre.issue(brow, (kb + 2) * 32, a.S, hi, a.bias_ld);
        if constexpr (BIAS == 2) spre.issue(bswz, kb + 2);
        tile_dma(kbase, a.ldk, (kb + 2) * 32, a.S, h * HD, lds, tid, wave_u);
        tile_dma(vbase, a.ldk, (kb + 2) * 32, a.S, h * HD, lds + TILE_BYTES / 2, tid, wave_u);
      }
      const bool need = live_wave && !(a.causal && (kb + 1) * 32 > q0 + 31);
      if (need) {
        if constexpr (BIAS == 1) bias_row_take<F16>(bcur, brow, (kb + 1) * 32, a.S, hi, bz);
        if constexpr (BIAS == 2) scur.template take<F16>(inv_scale, bz);
        dq_block<1, BIAS, F16>(a, ta, trx, qf, dof, dqt, (kb + 1) * 32, q0, qi, hi, dead_now, bz, dbrow, sc, lse_q, delta_q, c);
        my_last = kb + 1;
      }
      ATT_SYNC();
    }
  }
  if (dbrow && qi < a.T && my_last + 1 < nkb_all) {       // causally skipped blocks: dS == 0
    for (int key = (my_last + 1) * 32 + 4 * hi; key < a.S; key += 8)
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
        st4<F16>(op + dt * 32 + 8 * qq + 4 * hi, dqt[dt][4 * qq] * a.scale, dqt[dt][4 * qq + 1] * a.scale,
            dqt[dt][4 * qq + 2] * a.scale, dqt[dt][4 * qq + 3] * a.scale);
  }
  float *csq, *csc;
  cs_rows(csq, csc);
  if (csq || csc) {
    // (the lane number afresh from the hardware: nothing may live across the loop for this epilogue -- three waves per SIMD, 168 registers)
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (csq) csq[col64(ln)] = wave_colsum64<F16>(dqt, a.scale, qi < a.T, ln);        // (uniform) bias gradient of the q projection
    if (csc) {                                                                       // (uniform) gradient of c_attn
      const float dsum = wave_sum((ln < 32 && qi < a.T) ? delta_q : 0.f);
      if (ln == 0) *csc = dsum / head_scale(a, h);
    }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attn_bwd_dq_lds_kernel(AttnL a) { attn_bwd_dq_body<0, false>(a); }
__global__ __launch_bounds__(256) void attn_bwd_dq_bias_lds_kernel(AttnL a) { attn_bwd_dq_body<1, false>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attn_bwd_dq_f16_lds_kernel(AttnL a) { attn_bwd_dq_body<0, true>(a); }
__global__ __launch_bounds__(256) void attn_bwd_dq_bias_f16_lds_kernel(AttnL a) { attn_bwd_dq_body<1, true>(a); }
__global__ __launch_bounds__(256) void attn_bwd_dq_sbias_lds_kernel(AttnL a) { attn_bwd_dq_body<2, false>(a); }
__global__ __launch_bounds__(256) void attn_bwd_dq_sbias_f16_lds_kernel(AttnL a) { attn_bwd_dq_body<2, true>(a); }
// (the bias through LDS as in dK/dV + three waves per SIMD: 9 spilled registers, backward 280 -> 274 us at 32 x 12 x 448^2, nothing in the step)
// (three waves per SIMD cost this form 16 spilled registers and measured the same: 302 vs 300 us backward at 32 x 12 x 448^2)

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// Register budget.  Round 1's form of this kernel held 400 registers (one wave per SIMD: nothing ran while a wave waited for
// fragments, the stage barrier or its own softmax arithmetic).  Now: the additive bias is a template parameter (its 16 + 2 x 16
// staging registers exist only in the biased instantiation); the per-row softmax statistics (lse, delta) of a query block travel
// with the block's Q / dO tiles through LDS (256 bytes per stage, one 4-byte LDS-DMA per lane of wave 0) instead of two
// prefetched register sets of 32; and the transposed Q / dO fragments are read AFTER the S / dP MFMAs have been issued, into the
// registers of the row-major fragments those MFMAs have consumed.
constexpr int STAT_BYTES = 256;                            // per stage: lse[32] | delta[32] of the query block (fp32)
// Rows of the block beyond the sample's T queries read row T - 1 (their scores are masked: the value is never used).  Without the clamp
// the last block of a sample whose length is not a multiple of 32 read up to 31 floats past q_off + q_len -- for the LAST head of the
// LAST sample of a packed batch that ends in the bucket's final rows, past the END of the [heads, Tpad] buffers: harmless wherever the
// allocator had mapped memory behind them, a GPU memory access fault where the tensor closed a segment (round 5's cfg-2b graph fault;
// found with tools/capture_audit.py --dump-before-replay + tools/r6/postmortem.py, profiles/round6_graph_fault_root_cause.txt).
__device__ __forceinline__ void stat_dma(const float* __restrict__ lse_bh, const float* __restrict__ delta_bh, int q0, int T,
                                         unsigned char* __restrict__ lds_stat, int lane, int wave_u) {
  if (wave_u == 0) {                                       // lanes 0..31: lse[q0 + lane], lanes 32..63: delta[q0 + lane - 32]
    int row = q0 + (lane & 31);
    row = row < T ? row : T - 1;
    const float* src = (lane < 32 ? lse_bh : delta_bh) + row;
    __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)lds_stat, 4, 0, 0);
  }
}

template <int BUF, int BIAS, bool F16>
__device__ __forceinline__ void dkv_block(const AttnL& a, const TileAddr& ta, const uint32_t* trx, uint32_t stat_addr,
                                          const bf16x8 (&kf)[4], const bf16x8 (&vf)[4], f32x16 (&dvt)[2], f32x16 (&dkt)[2], int q0,
                                          int key0, int ki, int hi, bool key_dead, float live, const float (&bz)[16], float sc,
                                          float c, float inv_scale, uint32_t bias_addr) {
  constexpr int QOFF = BUF * 2 * TILE_BYTES, DOOFF = QOFF + TILE_BYTES, SOFF = BUF * STAT_BYTES;
  u64x2 qf[4], dof[4];
  rd128<QOFF>(qf[0], ta.km[0]); rd128<QOFF>(qf[1], ta.km[1]); rd128<QOFF>(qf[2], ta.km[2]); rd128<QOFF>(qf[3], ta.km[3]);
  rd128<DOOFF>(dof[0], ta.km[0]); rd128<DOOFF>(dof[1], ta.km[1]); rd128<DOOFF>(dof[2], ta.km[2]); rd128<DOOFF>(dof[3], ta.km[3]);
  // lse / delta of this lane's 16 query rows (rows 8g + 4hi + 0..3): four 16-byte reads each, the 32 lanes of a half read the
  // same address (broadcast)
  u64x2 l2[4], d2[4];
  rd128<SOFF>(l2[0], stat_addr); rd128<SOFF + 32>(l2[1], stat_addr); rd128<SOFF + 64>(l2[2], stat_addr); rd128<SOFF + 96>(l2[3], stat_addr);
  rd128<SOFF + 128>(d2[0], stat_addr); rd128<SOFF + 160>(d2[1], stat_addr); rd128<SOFF + 192>(d2[2], stat_addr); rd128<SOFF + 224>(d2[3], stat_addr);
  ATT_WAIT4(qf[0], qf[1], qf[2], qf[3]);
  ATT_WAIT4(dof[0], dof[1], dof[2], dof[3]);
  ATT_WAIT4(l2[0], l2[1], l2[2], l2[3]);
  ATT_WAIT4(d2[0], d2[1], d2[2], d2[3]);
  f32x16 st, dp;
  zero16f(st);
  zero16f(dp);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    st = ATT_MFMA(qf[kk], kf[kk], st);     // S[q][key]
    dp = ATT_MFMA(dof[kk], vf[kk], dp);    // dP[q][key]
  }
  __builtin_amdgcn_sched_barrier(0);       // the transposed reads go out behind the MFMAs (hipcc would hoist them: +32 registers)
  u64x2 qtf[2][2], dotf[2][2];
  rdtr<QOFF, 0>(qtf[0][0], ta.tr[0], trx[0]);
  rdtr<QOFF, 0>(qtf[0][1], ta.tr[1], trx[1]);
  rdtr<QOFF, 1>(qtf[1][0], ta.tr[0], trx[0]);
  rdtr<QOFF, 1>(qtf[1][1], ta.tr[1], trx[1]);
  rdtr<DOOFF, 0>(dotf[0][0], ta.tr[0], trx[0]);
  rdtr<DOOFF, 0>(dotf[0][1], ta.tr[1], trx[1]);
  rdtr<DOOFF, 1>(dotf[1][0], ta.tr[0], trx[0]);
  rdtr<DOOFF, 1>(dotf[1][1], ta.tr[1], trx[1]);
  float lv[16], dv16[16];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const float4 l4 = __builtin_bit_cast(float4, l2[g4]), d4 = __builtin_bit_cast(float4, d2[g4]);
    lv[4 * g4] = l4.x; lv[4 * g4 + 1] = l4.y; lv[4 * g4 + 2] = l4.z; lv[4 * g4 + 3] = l4.w;
    dv16[4 * g4] = d4.x; dv16[4 * g4 + 1] = d4.y; dv16[4 * g4 + 2] = d4.z; dv16[4 * g4 + 3] = d4.w;
  }
  if constexpr (BIAS == 3) {                // the swizzled shared bias of this block from the stage's LDS image (bias_dma), decoded only
                                            // now -- the kernel is at its register limit while the row-major fragments are live -- and added
                                            // onto the raw scores, times 1 / scale
    constexpr int BOFF = 4 * TILE_BYTES + 2 * STAT_BYTES + BUF * BIAS_STAGE_BYTES;
    u64x2 b0, b1;
    rd128<BOFF>(b0, bias_addr);
    rd128<BOFF + 1024>(b1, bias_addr);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1));
    float bs[16];
    bias_words<F16>(b0, b1, inv_scale, bs);
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] += bs[r];
  }
  float p[16], ds[16];
  const bool general = BIAS == 1 || (q0 + 32 > a.T) || (a.causal && (key0 + 31 > q0));
  if (general) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + crowl(r, hi);
      float t = st[r] * sc;
      if (BIAS == 1) t += bz[r];
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
  ATT_WAIT4(qtf[0][0], qtf[0][1], qtf[1][0], qtf[1][1]);
  ATT_WAIT4(dotf[0][0], dotf[0][1], dotf[1][0], dotf[1][1]);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bf16x8 pf = pack8f<F16>(p + 8 * j);
    const bf16x8 dsf = pack8f<F16>(ds + 8 * j);
    dvt[0] = ATT_MFMA(dotf[j][0], pf, dvt[0]);
    dvt[1] = ATT_MFMA(dotf[j][1], pf, dvt[1]);
    dkt[0] = ATT_MFMA(qtf[j][0], dsf, dkt[0]);
    dkt[1] = ATT_MFMA(qtf[j][1], dsf, dkt[1]);
  }
}

template <int BIAS, bool F16>
__device__ __forceinline__ void attn_bwd_dkv_body(AttnL a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);         // [2 buffers][Q tile | dO tile], then [2][lse | delta]
  unsigned char* lds_stat = smem + 4 * TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int i = lane & 31, hi = lane >> 5;
  int bh = blockIdx.y, b = bh / a.heads;
  const int h = bh % a.heads;
  const int kb0 = blockIdx.x * 128;
  if (a.seg && blockIdx.x == gridDim.x - 1) {
    seg_zero_fill(a.seg, a.B, b, h, true, a.rows_k, a.dk, a.ldk, a.dv, a.ldk, tid);
    return;
  }
  // partial rows of the column sums of dk / dv this workgroup owns: (sample, key tile), see AttnL (computed where they are used)
  const int b_sample = __builtin_amdgcn_readfirstlane(b);      // (seg_enter rebases b to 0; a scalar register across the loop)
  auto cs_rows = [&](float*& csk, float*& csv) {              // this WAVE's partial rows
    const int64_t slot = ((int64_t)b_sample * (gridDim.x - (a.seg ? 1 : 0)) + blockIdx.x) * 4 + wave_u;
    csk = a.cs_k ? a.cs_k + slot * a.cs_ldk + h * HD : nullptr;
    csv = a.cs_v ? a.cs_v + slot * a.cs_ldk + h * HD : nullptr;
  };
  if (!seg_enter(a, b, bh, h, kb0, false)) {
    float *csk, *csv;
    cs_rows(csk, csv);
    if (csk) csk[lane] = 0.f;
    if (csv) csv[lane] = 0.f;
    return;
  }
  const int key0 = kb0 + wave * 32;
  const int ki = key0 + i;
  const int krow = ki < a.S ? ki : a.S - 1;
  const bf16_t* kp_ = a.k + ((int64_t)b * a.S + krow) * a.ldk + h * HD + hi * 8;
  const bf16_t* vp_ = a.v + ((int64_t)b * a.S + krow) * a.ldk + h * HD + hi * 8;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    kf[kk] = ld16(kp_ + kk * 16);
    vf[kk] = ld16(vp_ + kk * 16);
  }
  const bool key_dead = ki >= a.S || (a.kpm && a.kpm[(int64_t)b * a.S + krow] != 0);
  const float live = key_dead ? 0.f : 1.f;
  const float c = head_scale(a, h);
  const bf16_t* qbase = a.q + (int64_t)b * a.T * a.ldq;
  const bf16_t* dobase = a.dout + (int64_t)b * a.T * a.ldo;
  const float* lse_bh = a.lse + (int64_t)bh * a.Tpad;
  const float* delta_bh = a.delta + (int64_t)bh * a.Tpad;
  const bf16_t* bcol = BIAS == 1 ? a.bias + (int64_t)(blockIdx.y / a.heads) * a.bias_bs + (int64_t)h * a.bias_hs + krow : nullptr;
  const bf16_t* bswz = nullptr;                          // BIAS 3: this lane's 16 values of block (0, key0 / 32) of head h (column image)
  if constexpr (BIAS == 3) {
    const int kt = (key0 >> 5) < a.bias_nkt ? (key0 >> 5) : a.bias_nkt - 1;
    bswz = a.bias_sc + (((int64_t)h * a.bias_nkt + kt) * a.bias_nqt * 64 + lane) * 16;
  }
  const float inv_scale = 1.0f / a.scale;
  const float sc = a.scale * LOG2E;
  TileAddr ta;
  ta.init((uint32_t)(uintptr_t)smem, lane);
  const uint32_t* trx = ta.trx;
  const uint32_t stat_addr = (uint32_t)(uintptr_t)lds_stat + 16 * hi;

  f32x16 dvt[2], dkt[2];
  zero16f(dvt[0]); zero16f(dvt[1]); zero16f(dkt[0]); zero16f(dkt[1]);
  const int nqb = (a.T + 31) / 32;
  const int qb_first = a.causal ? kb0 / 32 : 0;          // the workgroup starts where its FIRST wave needs
  const bool live_wave = key0 < a.S;
  BiasCol bA, bB;                                        // bias columns of the even / odd query block in flight (BIAS 1 only)
  unsigned char* lds_bias = smem + 4 * TILE_BYTES + 2 * STAT_BYTES + wave_u * 2048;      // (BIAS 3) this wave's 2 KiB of stage 0
  const uint32_t bias_addr = (uint32_t)(uintptr_t)smem + wave * 2048 + lane * 16;
  float bz[16];
  if (qb_first < nqb) {
    if constexpr (BIAS == 1) bA.issue(bcol, qb_first * 32, a.T, a.bias_ld, hi);
    if constexpr (BIAS == 3) bias_dma(bswz, qb_first, lds_bias);
    tile_dma(qbase, a.ldq, qb_first * 32, a.T, h * HD, lds, tid, wave_u);
    tile_dma(dobase, a.ldo, qb_first * 32, a.T, h * HD, lds + TILE_BYTES / 2, tid, wave_u);
    stat_dma(lse_bh, delta_bh, qb_first * 32, a.T, lds_stat, lane, wave_u);
  }
  ATT_SYNC();
  for (int qb = qb_first; qb < nqb; qb += 2) {
    {
      if (qb + 1 < nqb) {
        if constexpr (BIAS == 1) bB.issue(bcol, (qb + 1) * 32, a.T, a.bias_ld, hi);   // ordinary loads BEFORE the DMA: their wait leaves the DMA in flight
        if constexpr (BIAS == 3) bias_dma(bswz, qb + 1, lds_bias + BIAS_STAGE_BYTES);
        tile_dma(qbase, a.ldq, (qb + 1) * 32, a.T, h * HD, lds + TILE_BYTES, tid, wave_u);
        tile_dma(dobase, a.ldo, (qb + 1) * 32, a.T, h * HD, lds + TILE_BYTES + TILE_BYTES / 2, tid, wave_u);
        stat_dma(lse_bh, delta_bh, (qb + 1) * 32, a.T, lds_stat + STAT_BYTES, lane, wave_u);
      }
      const bool need = live_wave && !(a.causal && qb * 32 + 31 < key0);
      if (need) {
        if constexpr (BIAS == 1) bA.template take<F16>(bcol, bz);
        dkv_block<0, BIAS, F16>(a, ta, trx, stat_addr, kf, vf, dvt, dkt, qb * 32, key0, ki, hi, key_dead, live, bz, sc, c, inv_scale, bias_addr);
      }
      ATT_SYNC();
    }
    if (qb + 1 < nqb) {
      if (qb + 2 < nqb) {
        if constexpr (BIAS == 1) bA.issue(bcol, (qb + 2) * 32, a.T, a.bias_ld, hi);
        if constexpr (BIAS == 3) bias_dma(bswz, qb + 2, lds_bias);
        tile_dma(qbase, a.ldq, (qb + 2) * 32, a.T, h * HD, lds, tid, wave_u);
        tile_dma(dobase, a.ldo, (qb + 2) * 32, a.T, h * HD, lds + TILE_BYTES / 2, tid, wave_u);
        stat_dma(lse_bh, delta_bh, (qb + 2) * 32, a.T, lds_stat, lane, wave_u);
      }
      const bool need = live_wave && !(a.causal && (qb + 1) * 32 + 31 < key0);
      if (need) {
        if constexpr (BIAS == 1) bB.template take<F16>(bcol, bz);
        dkv_block<1, BIAS, F16>(a, ta, trx, stat_addr, kf, vf, dvt, dkt, (qb + 1) * 32, key0, ki, hi, key_dead, live, bz, sc, c, inv_scale, bias_addr);
      }
      ATT_SYNC();
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
        st4<F16>(dkp + d, dkt[dt][4 * qq] * a.scale, dkt[dt][4 * qq + 1] * a.scale, dkt[dt][4 * qq + 2] * a.scale,
            dkt[dt][4 * qq + 3] * a.scale);
        st4<F16>(dvp + d, dvt[dt][4 * qq] * c, dvt[dt][4 * qq + 1] * c, dvt[dt][4 * qq + 2] * c, dvt[dt][4 * qq + 3] * c);
      }
  }
  float *csk, *csv;
  cs_rows(csk, csv);
  if (csk || csv) {                                        // (uniform) bias gradients of the k / v projections
    // (the lane number afresh from the hardware: the kernel is at its register limit inside the loop, nothing may live across it for this)
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (csk) csk[col64(ln)] = wave_colsum64<F16>(dkt, a.scale, ki < a.S, ln);
    if (csv) csv[col64(ln)] = wave_colsum64<F16>(dvt, c, ki < a.S, ln);
  }
}

// two waves per SIMD for both forms: the bias-free one fits 256 registers; the biased one spills 16 and is still 16 % faster that way
// (tools/attn_bias_bench.py, 448 x 448: backward 409 -> 343 us)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dkv_lds_kernel(AttnL a) { attn_bwd_dkv_body<0, false>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dkv_bias_lds_kernel(AttnL a) { attn_bwd_dkv_body<1, false>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dkv_f16_lds_kernel(AttnL a) { attn_bwd_dkv_body<0, true>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dkv_bias_f16_lds_kernel(AttnL a) { attn_bwd_dkv_body<1, true>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dkv_sbias_lds_kernel(AttnL a) { attn_bwd_dkv_body<3, false>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dkv_sbias_f16_lds_kernel(AttnL a) { attn_bwd_dkv_body<3, true>(a); }


// ------------------------------------------------------------------------------------------------ backward: sum over the batch of dS
// The gradient of a batch-SHARED additive bias [A, Tb, Sb] (the position bias: abs-pos + rel-pos, identical for every sample) is
// G[h][i][j] = sum_b dS[b][h][i][j].  The reference gets it by materialising dS as [B*A, T, S] and letting autograd reduce the expand
// (154 MB per layer at cfg-2b, 1.97 GB at cfg-4 / B = 32).  Here one workgroup owns a [128 query positions x 64 key positions] tile of
// ONE head and a CHUNK of the batch, walks its samples, recomputes S = K Q^T and dP = V dO^T of that tile for every sample (two of
// the five products of the backward pass; lse / delta come from the forward and the dQ kernel) and adds dS = P (dP c - delta) into
// registers: no atomics, no [B*A, T, S] tensor.  One chunk (long sequences: the tiles alone fill the chip) writes G itself; several
// chunks (short sequences) write fp32 partials [chunk][A][Tb][Sb] that ofa_fold_batched adds in chunk order -- deterministic either
// way.  The next sample's query-side operands (Q / dO rows, lse, delta) are fetched while the current one is being computed.
// Ragged mode: the tile is addressed by the position inside the sample, samples shorter than the tile's origin are skipped.
struct DsumQ {                    // the query-side operands of one sample for this lane's row
  bf16x8 qf[4], dof[4];
  float lse, delta;
};
template <bool F16>
__device__ __forceinline__ void attn_bwd_dsum_body(AttnL a, float* __restrict__ G, int64_t g_ld, int64_t g_hs, int64_t g_cs, int Tb, int Sb,
                                                   int nchunk, int bper, int out16) {   // out16: G is the 16-bit gradient itself (one chunk)
  constexpr int NKB = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);         // [2 buffers][2 key blocks][K tile | V tile], then the chunk's geometry table
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int i = lane & 31, hi = lane >> 5;
  const int h = blockIdx.z / nchunk, chunk = blockIdx.z % nchunk;
  const int b_lo = chunk * bper, b_hi = (b_lo + bper < a.B) ? b_lo + bper : a.B;
  const int qb0 = blockIdx.y * 128, kc0 = blockIdx.x * NKB;      // first query position / first key block of the tile
  const int q0 = qb0 + wave * 32, qi = q0 + i;
  const float sc = a.scale * LOG2E;
  const float c = head_scale(a, h);
  TileAddr ta;
  ta.init((uint32_t)(uintptr_t)smem, lane);
  // the tile's bias values (times log2 e), loaded once: they do not depend on the sample
  float bz[NKB][16];
  {
    const int qrow = qi < Tb ? qi : Tb - 1;
    const bf16_t* brow = a.bias + (int64_t)h * a.bias_hs + (int64_t)qrow * a.bias_ld;
#pragma unroll
    for (int j = 0; j < NKB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (kc0 + j) * 32 + crowl(r, hi);
        bz[j][r] = key < Sb ? dec1<F16>(brow[key]) * LOG2E : 0.f;
      }
  }
  float acc[NKB][16];
#pragma unroll
  for (int j = 0; j < NKB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // The geometry {q_len, k_len, q_off, k_off} of this chunk's samples goes through LDS once (behind the tiles): in ragged mode every
  // look-up of the segment table would otherwise be a scalar memory round trip, several per stage.
  int4* geo = reinterpret_cast<int4*>(smem + 8 * TILE_BYTES);
  const int nb = b_hi - b_lo;
  for (int t = tid; t < nb; t += 256) {
    const int b = b_lo + t;
    int4 g4;
    if (a.seg) {
      const int4 sg = reinterpret_cast<const int4*>(a.seg)[b];
      g4 = make_int4(sg.y, sg.w, sg.x, sg.z);
    } else {
      g4 = make_int4(a.T, a.S, b * a.T, b * a.S);
    }
    geo[t] = g4;
  }
  __syncthreads();
  auto geom = [&](int b) {                               // (uniform: one LDS read, broadcast)
    const int4 g4 = geo[b - b_lo];
    return make_int4(__builtin_amdgcn_readfirstlane(g4.x), __builtin_amdgcn_readfirstlane(g4.y), __builtin_amdgcn_readfirstlane(g4.z),
                     __builtin_amdgcn_readfirstlane(g4.w));
  };
  // One stage per SAMPLE: the K / V tiles of both key blocks of the tile land together (4 tiles, 16 KiB, double buffered), so the
  // stage barrier and the DMA latency are paid once per two blocks of work.  A sample is live when it reaches the tile (uniform).
  auto next_live = [&](int b) {
    ++b;
    while (b < b_hi) {
      const int4 g4 = geom(b);
      if (qb0 < g4.x && kc0 * 32 < g4.y) break;
      ++b;
    }
    return b;
  };
  auto stage = [&](const int4& g4, int buf) {
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const int key0 = (kc0 + j) * 32;                   // (a block beyond the sample's keys re-reads its last row: masked below)
      tile_dma(a.k + (int64_t)g4.w * a.ldk, a.ldk, key0, g4.y, h * HD, lds + (buf * 2 * NKB + 2 * j) * (TILE_BYTES / 2), tid, wave_u);
      tile_dma(a.v + (int64_t)g4.w * a.ldk, a.ldk, key0, g4.y, h * HD, lds + (buf * 2 * NKB + 2 * j + 1) * (TILE_BYTES / 2), tid, wave_u);
    }
  };
  auto fetch_q = [&](int b, const int4& g4, DsumQ& d) {  // ordinary loads, always issued in front of a stage's DMA
    const int qrow = qi < g4.x ? qi : g4.x - 1;
    const bf16_t* qp = a.q + ((int64_t)g4.z + qrow) * a.ldq + h * HD + hi * 8;
    const bf16_t* dop = a.dout + ((int64_t)g4.z + qrow) * a.ldo + h * HD + hi * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      d.qf[kk] = ld16(qp + kk * 16);
      d.dof[kk] = ld16(dop + kk * 16);
    }
    const int64_t srow = a.seg ? (int64_t)h * a.Tpad + g4.z + qrow : ((int64_t)b * a.heads + h) * a.Tpad + qrow;
    d.lse = a.lse[srow];
    d.delta = a.delta[srow];
  };
  auto kflag_of = [&](int b, int j, const int4& g4) {    // dead-key flag of this lane's key in block j of sample b (byte load: a stage AHEAD)
    const uint8_t* kp = a.kpm ? a.kpm + (int64_t)b * a.S : nullptr;
    return dead_flag(kp, (kc0 + j) * 32, g4.y, i);
  };
  int b = next_live(b_lo - 1);
  DsumQ cur, nxt;
  int4 gc = make_int4(0, 0, 0, 0), gn = gc;
  int kflag[NKB], kflag_n[NKB];
#pragma unroll
  for (int j = 0; j < NKB; ++j) kflag[j] = kflag_n[j] = 0;
  if (b < b_hi) {
    gc = geom(b);
    fetch_q(b, gc, cur);
#pragma unroll
    for (int j = 0; j < NKB; ++j) kflag[j] = kflag_of(b, j, gc);
    stage(gc, 0);
  }
  ATT_SYNC();
  int buf = 0;
  while (b < b_hi) {
    const int bn = next_live(b);
    const int T_b = gc.x;
    uint32_t dead_now[NKB];
#pragma unroll
    for (int j = 0; j < NKB; ++j) dead_now[j] = dead_ballot(kflag[j]);
    if (bn < b_hi) {                                     // the next sample's rows, flags and tiles travel while this one computes
      gn = geom(bn);
      fetch_q(bn, gn, nxt);
#pragma unroll
      for (int j = 0; j < NKB; ++j) kflag_n[j] = kflag_of(bn, j, gn);
      stage(gn, buf ^ 1);
    }
    if (q0 < T_b) {
      const bool rowok = qi < T_b;
#pragma unroll
      for (int j = 0; j < NKB; ++j) {
        const int key0 = (kc0 + j) * 32;
        if (key0 >= gc.y) continue;                      // (uniform)
        u64x2 kf[4], vf[4];
        const uint32_t koff = (uint32_t)((buf * 2 * NKB + 2 * j) * TILE_BYTES);
        rd128<0>(kf[0], ta.km[0] + koff); rd128<0>(kf[1], ta.km[1] + koff); rd128<0>(kf[2], ta.km[2] + koff); rd128<0>(kf[3], ta.km[3] + koff);
        rd128<TILE_BYTES>(vf[0], ta.km[0] + koff); rd128<TILE_BYTES>(vf[1], ta.km[1] + koff); rd128<TILE_BYTES>(vf[2], ta.km[2] + koff);
        rd128<TILE_BYTES>(vf[3], ta.km[3] + koff);
        ATT_WAIT4(kf[0], kf[1], kf[2], kf[3]);
        ATT_WAIT4(vf[0], vf[1], vf[2], vf[3]);
        f32x16 st, dp;
        zero16f(st);
        zero16f(dp);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          st = ATT_MFMA(kf[kk], cur.qf[kk], st);
          dp = ATT_MFMA(vf[kk], cur.dof[kk], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jk = crowl(r, hi), key = key0 + jk;
          bool dead = !rowok || ((dead_now[j] >> jk) & 1u);
          if (a.causal) dead |= key > qi;
          const float t = st[r] * sc + bz[j][r];
          const float p = dead ? 0.f : __builtin_amdgcn_exp2f(t - cur.lse);
          acc[j][r] += p * (dp[r] * c - cur.delta);
        }
      }
    }
    ATT_SYNC();
    buf ^= 1;
    if (bn < b_hi) {                                     // (landed: ATT_SYNC waited for vmcnt(0))
      cur = nxt;
      gc = gn;
#pragma unroll
      for (int j = 0; j < NKB; ++j) kflag[j] = kflag_n[j];
    }
    b = bn;
  }
  // G[h][qi][keys of the tile] (or this chunk's partial)
  if (qi < Tb && out16) {                                  // one chunk and a 16-bit gradient: rounded here, no fold launch
    bf16_t* gp = reinterpret_cast<bf16_t*>(G) + (int64_t)h * g_hs + (int64_t)qi * g_ld;
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const int key0 = (kc0 + j) * 32;
      if ((g_ld & 3) == 0 && key0 + 32 <= Sb) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          st4<F16>(gp + key0 + 8 * g4 + 4 * hi, acc[j][4 * g4], acc[j][4 * g4 + 1], acc[j][4 * g4 + 2], acc[j][4 * g4 + 3]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + crowl(r, hi);
          if (key < Sb) gp[key] = enc1<F16>(acc[j][r]);
        }
      }
    }
  } else if (qi < Tb) {
    float* gp = G + (int64_t)chunk * g_cs + (int64_t)h * g_hs + (int64_t)qi * g_ld;
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const int key0 = (kc0 + j) * 32;
      if ((g_ld & 3) == 0 && key0 + 32 <= Sb) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<float4*>(gp + key0 + 8 * g4 + 4 * hi) = make_float4(acc[j][4 * g4], acc[j][4 * g4 + 1], acc[j][4 * g4 + 2], acc[j][4 * g4 + 3]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + crowl(r, hi);
          if (key < Sb) gp[key] = acc[j][r];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dsum_kernel(AttnL a, float* G, int64_t g_ld, int64_t g_hs, int64_t g_cs, int Tb, int Sb, int nchunk, int bper, int out16) { attn_bwd_dsum_body<false>(a, G, g_ld, g_hs, g_cs, Tb, Sb, nchunk, bper, out16); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_bwd_dsum_f16_kernel(AttnL a, float* G, int64_t g_ld, int64_t g_hs, int64_t g_cs, int Tb, int Sb, int nchunk, int bper, int out16) { attn_bwd_dsum_body<true>(a, G, g_ld, g_hs, g_cs, Tb, Sb, nchunk, bper, out16); }

static int attnl_check(int B, int heads, int T, int S, int Tpad, int64_t ldq, int64_t ldk, int64_t ldo, int dtype) {
  OFA_REQUIRE(dtype == OFA_BF16 || dtype == OFA_F16, OFA_ERR_UNSUPPORTED, "fused attention is bf16 / fp16 only (dtype %d); use the unfused path", dtype);
  OFA_REQUIRE(B > 0 && heads > 0 && T > 0 && S > 0, OFA_ERR_INVALID, "attention: bad shape B=%d heads=%d T=%d S=%d", B, heads, T, S);
  OFA_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldo % 8) == 0, OFA_ERR_INVALID, "attention: leading dims must be multiples of 8");
  OFA_REQUIRE(Tpad % 32 == 0 && Tpad >= T, OFA_ERR_INVALID, "attention: Tpad must be a multiple of 32 covering T (T=%d Tpad=%d)", T, Tpad);
  return 0;
}

}  // namespace ofa
using namespace ofa;

extern "C" int ofa_attn_fwd(const void* q, const void* k, const void* v, const void* bias, const uint8_t* kpm,
                            const void* c_attn, int c_attn_dtype, void* out, float* lse, int B, int heads, int T, int S,
                            int Tpad, int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal, const int32_t* seg,
                            int rows_q, int rows_k, int dtype, void* stream) {
  if (int rc = attnl_check(B, heads, T, S, Tpad, ldq, ldk, ldo, dtype)) return rc;
  OFA_REQUIRE(q && k && v && out, OFA_ERR_INVALID, "attn_fwd: null pointer");
  OFA_REQUIRE(scale > 0.f, OFA_ERR_INVALID, "attn_fwd: the score scale must be positive (row maxima are taken on the raw scores), got %g", (double)scale);
  OFA_REQUIRE(!seg || (!bias && !kpm && lse && !((uintptr_t)seg & 15)), OFA_ERR_INVALID,
              "attn_fwd: the ragged (seg) mode takes no bias / key-padding mask, needs lse and a 16-byte aligned table");
  OFA_REQUIRE(OFA_DT_OK(c_attn_dtype), OFA_ERR_INVALID, "attn_fwd: bad c_attn dtype %d", c_attn_dtype);
  AttnL a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.bias = (const bf16_t*)bias; a.kpm = kpm;
  a.c_attn = c_attn; a.c_dt = c_attn_dtype; a.out = (bf16_t*)out; a.lse = lse; a.B = B; a.heads = heads; a.T = T; a.S = S; a.Tpad = Tpad;
  a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.scale = scale; a.causal = causal; a.seg = seg; a.rows_q = rows_q; a.rows_k = rows_k;
  a.bias_ld = S; a.bias_hs = (int64_t)T * S; a.bias_bs = (int64_t)heads * T * S;
  OFA_REQUIRE(!seg || (rows_q > 0 && rows_k > 0 && Tpad >= rows_q), OFA_ERR_INVALID, "attn_fwd: ragged mode needs rows_q / rows_k and Tpad >= rows_q");
  const dim3 grid(cdiv(T, 128) + (seg ? 1 : 0), B * heads);
  auto kern = dtype == OFA_F16 ? (bias ? attn_fwd_bias_f16_lds_kernel : attn_fwd_f16_lds_kernel) : (bias ? attn_fwd_bias_lds_kernel : attn_fwd_lds_kernel);
  hipLaunchKernelGGL(kern, grid, dim3(256), 4 * TILE_BYTES, (hipStream_t)stream, a);
  return check_launch("attn_fwd");
}

// the optional column-sum outputs of the backward kernels (AttnL::cs_*)
struct AttnCs { float* q; int64_t ldq; float* k; float* v; int64_t ldk; float* c; };
static int attn_cs_set(AttnL& a, const AttnCs& cs, int heads, const void* c_attn) {
  OFA_REQUIRE((!cs.q || cs.ldq >= (int64_t)heads * HD) && ((!cs.k && !cs.v) || cs.ldk >= (int64_t)heads * HD), OFA_ERR_INVALID,
              "attn_bwd: a column-sum partial row holds heads * 64 = %d floats (row strides %lld / %lld)", heads * HD, (long long)cs.ldq, (long long)cs.ldk);
  OFA_REQUIRE(!cs.c || c_attn, OFA_ERR_INVALID, "attn_bwd: cs_c (partial sums of the c_attn gradient) without c_attn");
  a.cs_q = cs.q; a.cs_k = cs.k; a.cs_v = cs.v; a.cs_c = cs.c; a.cs_ldq = cs.ldq; a.cs_ldk = cs.ldk;
  return 0;
}

static int attn_bwd_impl(const void* q, const void* k, const void* v, const void* dout, const void* bias,
                            const uint8_t* kpm, const void* c_attn, int c_attn_dtype, const float* lse, float* delta,
                            const void* out, void* dq, void* dk, void* dv, void* dbias, int B, int heads, int T, int S, int Tpad,
                            int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal, const int32_t* seg, int rows_q,
                            int rows_k, int dtype, void* stream, const AttnCs& cs) {
  if (int rc = attnl_check(B, heads, T, S, Tpad, ldq, ldk, ldo, dtype)) return rc;
  OFA_REQUIRE(!seg || (!bias && !kpm && !dbias && !((uintptr_t)seg & 15)), OFA_ERR_INVALID,
              "attn_bwd: the ragged (seg) mode takes no bias / key-padding mask / dbias and a 16-byte aligned table");
  OFA_REQUIRE(OFA_DT_OK(c_attn_dtype), OFA_ERR_INVALID, "attn_bwd: bad c_attn dtype %d", c_attn_dtype);
  OFA_REQUIRE(q && k && v && dout && lse && delta && dq && dk && dv, OFA_ERR_INVALID, "attn_bwd: null pointer");
  AttnL a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dout = (const bf16_t*)dout;
  a.bias = (const bf16_t*)bias; a.kpm = kpm; a.c_attn = c_attn; a.c_dt = c_attn_dtype; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.out = (bf16_t*)const_cast<void*>(out);   // != NULL: the dQ kernel computes delta itself (and writes it)
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.dbias = (bf16_t*)dbias; a.B = B; a.heads = heads; a.T = T;
  a.S = S; a.Tpad = Tpad; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.scale = scale; a.causal = causal; a.seg = seg;
  a.rows_q = rows_q; a.rows_k = rows_k;
  a.bias_ld = S; a.bias_hs = (int64_t)T * S; a.bias_bs = (int64_t)heads * T * S;
  OFA_REQUIRE(!seg || (rows_q > 0 && rows_k > 0 && Tpad >= rows_q), OFA_ERR_INVALID, "attn_bwd: ragged mode needs rows_q / rows_k and Tpad >= rows_q");
  if (int rc = attn_cs_set(a, cs, heads, c_attn)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const dim3 q_grid(cdiv(T, 128) + (seg ? 1 : 0), B * heads);
  auto dq_kern = dtype == OFA_F16 ? (bias ? attn_bwd_dq_bias_f16_lds_kernel : attn_bwd_dq_f16_lds_kernel) : (bias ? attn_bwd_dq_bias_lds_kernel : attn_bwd_dq_lds_kernel);
  hipLaunchKernelGGL(dq_kern, q_grid, dim3(256), 4 * TILE_BYTES, st, a);
  int rc = check_launch("attn_bwd_dq");
  if (rc) return rc;
  const dim3 kv_grid(cdiv(S, 128) + (seg ? 1 : 0), B * heads);
  auto kv_kern = dtype == OFA_F16 ? (bias ? attn_bwd_dkv_bias_f16_lds_kernel : attn_bwd_dkv_f16_lds_kernel) : (bias ? attn_bwd_dkv_bias_lds_kernel : attn_bwd_dkv_lds_kernel);
  hipLaunchKernelGGL(kv_kern, kv_grid, dim3(256), 4 * TILE_BYTES + 2 * STAT_BYTES, st, a);
  return check_launch("attn_bwd_dkv");
}

extern "C" int ofa_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const void* bias,
                            const uint8_t* kpm, const void* c_attn, int c_attn_dtype, const float* lse, float* delta,
                            const void* out, void* dq, void* dk, void* dv, void* dbias, int B, int heads, int T, int S, int Tpad,
                            int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal, const int32_t* seg, int rows_q,
                            int rows_k, int dtype, void* stream) {
  return attn_bwd_impl(q, k, v, dout, bias, kpm, c_attn, c_attn_dtype, lse, delta, out, dq, dk, dv, dbias, B, heads, T, S, Tpad, ldq, ldk, ldo,
                       scale, causal, seg, rows_q, rows_k, dtype, stream, AttnCs{});
}

extern "C" int ofa_attn_cs_slots(int B, int rows) { return B * cdiv(rows, 128) * 4; }   // one partial row per wave of a (sample, tile) workgroup

extern "C" int ofa_attn_bwd_cs(const void* q, const void* k, const void* v, const void* dout, const void* bias,
                               const uint8_t* kpm, const void* c_attn, int c_attn_dtype, const float* lse, float* delta,
                               const void* out, void* dq, void* dk, void* dv, void* dbias, int B, int heads, int T, int S, int Tpad,
                               int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal, const int32_t* seg, int rows_q,
                               int rows_k, int dtype, float* cs_q, int64_t cs_ldq, float* cs_k, float* cs_v, int64_t cs_ldk, float* cs_c,
                               void* stream) {
  return attn_bwd_impl(q, k, v, dout, bias, kpm, c_attn, c_attn_dtype, lse, delta, out, dq, dk, dv, dbias, B, heads, T, S, Tpad, ldq, ldk, ldo,
                       scale, causal, seg, rows_q, rows_k, dtype, stream, AttnCs{cs_q, cs_ldq, cs_k, cs_v, cs_ldk, cs_c});
}


// ---- batch-SHARED position bias [heads, Tb, Sb]: the swizzled-image kernels (BIAS 2) + the batch-summed dS kernel
static int sbias_check(const void* bias_sr, int Tb, int Sb, int T, int S, const int32_t* seg) {
  OFA_REQUIRE(bias_sr && Tb > 0 && Sb > 0, OFA_ERR_INVALID, "attn_sbias: the swizzled bias image of [heads, Tb, Sb] is required (Tb=%d Sb=%d)", Tb, Sb);
  OFA_REQUIRE(!((uintptr_t)bias_sr & 15), OFA_ERR_INVALID, "attn_sbias: the swizzled bias image must be 16-byte aligned");
  OFA_REQUIRE(Tb >= T && Sb >= S, OFA_ERR_INVALID, "attn_sbias: the bias covers %d x %d positions, the call needs %d x %d", Tb, Sb, T, S);
  (void)seg;
  return 0;
}

// bias_swz_row: the row image of the [heads, Tb, Sb] bias written by ofa_bias_build (ofa_bias_swz_elems elements)
extern "C" int ofa_attn_sbias_fwd(const void* q, const void* k, const void* v, const void* bias_swz_row, int Tb, int Sb, const uint8_t* kpm,
                                  const void* c_attn, int c_attn_dtype, void* out, float* lse, int B, int heads, int T, int S, int Tpad,
                                  int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal, const int32_t* seg, int rows_q,
                                  int rows_k, int dtype, void* stream) {
  if (int rc = attnl_check(B, heads, T, S, Tpad, ldq, ldk, ldo, dtype)) return rc;
  if (int rc = sbias_check(bias_swz_row, Tb, Sb, T, S, seg)) return rc;
  OFA_REQUIRE(q && k && v && out && lse, OFA_ERR_INVALID, "attn_sbias_fwd: null pointer");
  OFA_REQUIRE(scale > 0.f, OFA_ERR_INVALID, "attn_sbias_fwd: the score scale must be positive, got %g", (double)scale);
  OFA_REQUIRE(!seg || (!kpm && !((uintptr_t)seg & 15) && rows_q > 0 && rows_k > 0 && Tpad >= rows_q), OFA_ERR_INVALID,
              "attn_sbias_fwd: the ragged (seg) mode takes no key-padding mask, a 16-byte aligned table, rows_q / rows_k and Tpad >= rows_q");
  OFA_REQUIRE(OFA_DT_OK(c_attn_dtype), OFA_ERR_INVALID, "attn_sbias_fwd: bad c_attn dtype %d", c_attn_dtype);
  AttnL a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.kpm = kpm;
  a.c_attn = c_attn; a.c_dt = c_attn_dtype; a.out = (bf16_t*)out; a.lse = lse; a.B = B; a.heads = heads; a.T = T; a.S = S; a.Tpad = Tpad;
  a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.scale = scale; a.causal = causal; a.seg = seg; a.rows_q = rows_q; a.rows_k = rows_k;
  a.bias_sr = (const bf16_t*)bias_swz_row; a.bias_nqt = cdiv(Tb, 32); a.bias_nkt = cdiv(Sb, 32);
  const dim3 grid(cdiv(T, 128) + (seg ? 1 : 0), B * heads);
  auto kern = dtype == OFA_F16 ? attn_fwd_sbias_f16_lds_kernel : attn_fwd_sbias_lds_kernel;
  hipLaunchKernelGGL(kern, grid, dim3(256), 4 * TILE_BYTES, (hipStream_t)stream, a);
  return check_launch("attn_sbias_fwd");
}

extern "C" int ofa_attn_sbias_chunks(int B, int heads, int Tb, int Sb) {
  const int64_t tiles = (int64_t)cdiv(Sb, 64) * cdiv(Tb, 128) * heads;
  // enough workgroups for one round of the 256 CUs; more chunks only add partial slabs to fold (sweep of the target at
  // 32 x 12 heads: 448^2 119-120 us at 256 / 1024, 142 us at 2048; 64 x 448 35.5 us at 256, 41.3 at 1024, 48.3 at 2048)
  const int64_t target = 256;
  int64_t n = (target + tiles - 1) / tiles;
  n = n < 1 ? 1 : (n > B ? B : n);
  const int bper = cdiv(B, (int)n);
  return cdiv(B, bper);
}

// bias: the row-major [heads, Tb, Sb] tensor (read by the batch-sum kernel only: may be NULL when dbias_sum is);
// bias_swz_row / bias_swz_col: its two swizzled images (ofa_bias_build)
static int attn_sbias_bwd_impl(const void* q, const void* k, const void* v, const void* dout, const void* bias,
                                  const void* bias_swz_row, const void* bias_swz_col, int Tb, int Sb, const uint8_t* kpm, const void* c_attn, int c_attn_dtype, const float* lse, float* delta,
                                  const void* out, void* dq, void* dk, void* dv, void* dbias_sum, int dbias_dtype, float* ws, int64_t ws_bytes, int B,
                                  int heads, int T, int S, int Tpad, int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal,
                                  const int32_t* seg, int rows_q, int rows_k, int dtype, void* stream, const AttnCs& cs) {
  if (int rc = attnl_check(B, heads, T, S, Tpad, ldq, ldk, ldo, dtype)) return rc;
  if (int rc = sbias_check(bias_swz_row, Tb, Sb, T, S, seg)) return rc;
  OFA_REQUIRE(bias_swz_col && !((uintptr_t)bias_swz_col & 15) && (bias || !dbias_sum), OFA_ERR_INVALID,
              "attn_sbias_bwd: the column image of the bias (16-byte aligned) and, for dbias_sum, the row-major tensor are required");
  OFA_REQUIRE(!seg || (!kpm && !((uintptr_t)seg & 15) && rows_q > 0 && rows_k > 0 && Tpad >= rows_q), OFA_ERR_INVALID,
              "attn_sbias_bwd: the ragged (seg) mode takes no key-padding mask, a 16-byte aligned table, rows_q / rows_k and Tpad >= rows_q");
  OFA_REQUIRE(OFA_DT_OK(c_attn_dtype), OFA_ERR_INVALID, "attn_sbias_bwd: bad c_attn dtype %d", c_attn_dtype);
  OFA_REQUIRE(q && k && v && dout && lse && delta && out && dq && dk && dv, OFA_ERR_INVALID, "attn_sbias_bwd: null pointer");
  AttnL a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dout = (const bf16_t*)dout;
  a.bias = (const bf16_t*)bias; a.kpm = kpm; a.c_attn = c_attn; a.c_dt = c_attn_dtype; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.out = (bf16_t*)const_cast<void*>(out);
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.B = B; a.heads = heads; a.T = T;
  a.S = S; a.Tpad = Tpad; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.scale = scale; a.causal = causal; a.seg = seg;
  a.rows_q = rows_q; a.rows_k = rows_k;
  a.bias_ld = Sb; a.bias_hs = (int64_t)Tb * Sb; a.bias_bs = 0;
  a.bias_sr = (const bf16_t*)bias_swz_row; a.bias_sc = (const bf16_t*)bias_swz_col; a.bias_nqt = cdiv(Tb, 32); a.bias_nkt = cdiv(Sb, 32);
  if (int rc = attn_cs_set(a, cs, heads, c_attn)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const dim3 q_grid(cdiv(T, 128) + (seg ? 1 : 0), B * heads);
  auto dq_kern = dtype == OFA_F16 ? attn_bwd_dq_sbias_f16_lds_kernel : attn_bwd_dq_sbias_lds_kernel;
  hipLaunchKernelGGL(dq_kern, q_grid, dim3(256), 4 * TILE_BYTES, st, a);
  int rc = check_launch("attn_sbias_bwd_dq");
  if (rc) return rc;
  const dim3 kv_grid(cdiv(S, 128) + (seg ? 1 : 0), B * heads);
  auto kv_kern = dtype == OFA_F16 ? attn_bwd_dkv_sbias_f16_lds_kernel : attn_bwd_dkv_sbias_lds_kernel;
  hipLaunchKernelGGL(kv_kern, kv_grid, dim3(256), 4 * TILE_BYTES + 2 * STAT_BYTES + 2 * BIAS_STAGE_BYTES, st, a);
  rc = check_launch("attn_sbias_bwd_dkv");
  if (rc || !dbias_sum) return rc;
  // G = sum_b dS (after the dQ kernel: it wrote delta): [128 x 64] tiles of one head; the batch is cut into chunks when the tiles
  // alone would leave the chip idle, the chunks' fp32 partials are folded in chunk order
  const int nchunk = ofa_attn_sbias_chunks(B, heads, Tb, Sb);
  OFA_REQUIRE(OFA_DT_OK(dbias_dtype), OFA_ERR_INVALID, "attn_sbias_bwd: bad dbias dtype %d", dbias_dtype);
  const bool out16 = nchunk == 1 && dbias_dtype == dtype;          // one chunk, gradient in the operands' 16-bit type: rounded in the kernel
  const bool direct = (nchunk == 1 && dbias_dtype == OFA_F32) || out16;   // the kernel writes dbias_sum itself; otherwise the fold (+ cast) does
  OFA_REQUIRE(direct || (ws && ws_bytes >= (int64_t)nchunk * heads * Tb * Sb * 4), OFA_ERR_INVALID,
              "attn_sbias_bwd: the batch-sum kernel needs %lld bytes of workspace for %d chunks", (long long)nchunk * heads * Tb * Sb * 4, nchunk);
  const int bper = cdiv(B, nchunk);
  {
    const dim3 g(cdiv(Sb, 64), cdiv(Tb, 128), heads * nchunk);
    auto kn = dtype == OFA_F16 ? attn_bwd_dsum_f16_kernel : attn_bwd_dsum_kernel;
    float* dst = direct ? (float*)dbias_sum : ws;
    const size_t lds = 8 * TILE_BYTES + (size_t)bper * 16;
    OFA_REQUIRE(lds <= 64 * 1024, OFA_ERR_UNSUPPORTED, "attn_sbias_bwd: %d samples per chunk exceed the batch-sum kernel's geometry table", bper);
    hipLaunchKernelGGL(kn, g, dim3(256), lds, st, a, dst, (int64_t)Sb, (int64_t)Tb * Sb, (int64_t)heads * Tb * Sb, Tb, Sb, nchunk, bper, (int)out16);
    rc = check_launch("attn_sbias_bwd_dsum");
    if (rc || direct) return rc;
    ofa_fold_job job{ws, dbias_sum, (int64_t)heads * Tb * Sb, (int64_t)heads * Tb * Sb, nchunk, 0, 1.0f, dbias_dtype};
    return ofa_fold_batched(&job, 1, stream);
  }
}

extern "C" int ofa_attn_sbias_bwd(const void* q, const void* k, const void* v, const void* dout, const void* bias,
                                  const void* bias_swz_row, const void* bias_swz_col, int Tb, int Sb, const uint8_t* kpm, const void* c_attn, int c_attn_dtype, const float* lse, float* delta,
                                  const void* out, void* dq, void* dk, void* dv, void* dbias_sum, int dbias_dtype, float* ws, int64_t ws_bytes, int B,
                                  int heads, int T, int S, int Tpad, int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal,
                                  const int32_t* seg, int rows_q, int rows_k, int dtype, void* stream) {
  return attn_sbias_bwd_impl(q, k, v, dout, bias, bias_swz_row, bias_swz_col, Tb, Sb, kpm, c_attn, c_attn_dtype, lse, delta, out, dq, dk, dv, dbias_sum,
                             dbias_dtype, ws, ws_bytes, B, heads, T, S, Tpad, ldq, ldk, ldo, scale, causal, seg, rows_q, rows_k, dtype, stream, AttnCs{});
}

extern "C" int ofa_attn_sbias_bwd_cs(const void* q, const void* k, const void* v, const void* dout, const void* bias,
                                     const void* bias_swz_row, const void* bias_swz_col, int Tb, int Sb, const uint8_t* kpm, const void* c_attn, int c_attn_dtype, const float* lse, float* delta,
                                     const void* out, void* dq, void* dk, void* dv, void* dbias_sum, int dbias_dtype, float* ws, int64_t ws_bytes, int B,
                                     int heads, int T, int S, int Tpad, int64_t ldq, int64_t ldk, int64_t ldo, float scale, int causal,
                                     const int32_t* seg, int rows_q, int rows_k, int dtype, float* cs_q, int64_t cs_ldq, float* cs_k, float* cs_v,
                                     int64_t cs_ldk, float* cs_c, void* stream) {
  return attn_sbias_bwd_impl(q, k, v, dout, bias, bias_swz_row, bias_swz_col, Tb, Sb, kpm, c_attn, c_attn_dtype, lse, delta, out, dq, dk, dv, dbias_sum,
                             dbias_dtype, ws, ws_bytes, B, heads, T, S, Tpad, ldq, ldk, ldo, scale, causal, seg, rows_q, rows_k, dtype, stream,
                             AttnCs{cs_q, cs_ldq, cs_k, cs_v, cs_ldk, cs_c});
}

#ifdef OFA_ATTN_TIMELINE
extern "C" int ofa_debug_attn_timeline(void* dst_host, int64_t bytes) {   // measurement build only (tools/attn_timeline.py)
  return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_attn_tl), (size_t)bytes);
}
#endif
