// bf16 MFMA GEMM for gfx950 (CDNA4): C[M,N] = alpha*(op(A) op(B) + bias) (+C), fp32 accumulate.
//
// Covers the three contractions of a linear layer without any operand transposes in HBM:
//   forward   Y  = X  W^T      (A k-major,  B k-major : "NT",  transA=0, transB=1)
//   dgrad     dX = dY W        (A k-major,  B m-major : "NN",  transA=0, transB=0)
//   wgrad     dW = dY^T X      (A m-major,  B m-major : "TN",  transA=1, transB=0)
// (multihead_attention.py:199-217,346; transformer_layer.py:194,202; adaptor/text.py:94-96 output projection).
//
// Structure (MI355X_MICROARCH / cdna_hip_programming.md section 5):
//   * block tile (64*WM) x (64*WN) x 64, one wave per 64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 accumulators
//     (64 fp32 accumulator registers per lane);
//   * operands are staged global -> registers -> LDS with 16-byte vectors; the next K-tile's global loads are issued
//     before the current tile's MFMAs and written to the other LDS buffer afterwards (one barrier per K-tile);
//   * k-major tiles sit in LDS as [row][64+8] (144-byte rows: every 16-lane ds_read_b128 group hits 16 distinct
//     16-byte slots); m-major tiles (contraction index is the slow dimension in memory) sit as [k][rows+32] and are
//     read with ds_read_b64_tr_b16, the CDNA4 transposing LDS read, so no transpose pass ever touches HBM;
//   * the MFMA is issued "swapped" (A-operand = weight-side tile, B-operand = activation-side tile) so each lane ends
//     up with 4 consecutive output columns of ONE output row -> 8-byte (bf16) / 16-byte (fp32) stores;
//   * workgroup ids are remapped so that each XCD (private 4 MiB L2) walks a contiguous run of tiles;
//   * skinny outputs (wgrad: M,N ~ 768..3072, K = tokens) use split-K into an fp32 workspace + a reduce/epilogue
//     kernel (deterministic, no atomics).
// Roofline: MFMA-bound; algorithmic flops = 2*M*N*K.
#include <mutex>
#include <utility>
#include <vector>

#include "gemm_core.h"

namespace ofa {


// Phase-timestamp probe of tools/gemm_timeline.py (measurement build only: the Makefile never sets -DOFA_GEMM_TIMELINE; in the
// product every macro below is empty).  Thread 0 of each workgroup stamps the shader clock into 32 slots of `ws`: 0 entry,
// 1 first tile landed, 2 K loop done, 3 epilogue stores issued, 4 stores acknowledged, 5 HW_ID, 6 XCC_ID, 7 / 31 the 100 MHz
// clock at entry / exit, 8..23 K-step ends.
#ifdef OFA_GEMM_TIMELINE
#define OFA_TL(i) do { if (threadIdx.x == 0 && (i) < 24) tl[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define OFA_TL_BEGIN                                                                                                           \
  unsigned long long* tl = (unsigned long long*)ws + 32 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));   \
  int tlk = 8;                                                                                                                 \
  if (threadIdx.x == 0) {                                                                                                      \
    tl[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                                                         \
    tl[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                                                        \
    tl[7] = __builtin_amdgcn_s_memrealtime();                                                                                  \
  }                                                                                                                            \
  OFA_TL(0)
#define OFA_TL_STEP do { OFA_TL(tlk); ++tlk; } while (0)
#define OFA_TL_END                                                        \
  do {                                                                    \
    OFA_TL(3);                                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      \
    OFA_TL(4);                                                            \
    if (threadIdx.x == 0) tl[31] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
#else
#define OFA_TL(i) do { } while (0)
#define OFA_TL_BEGIN do { } while (0)
#define OFA_TL_STEP do { } while (0)
#define OFA_TL_END do { } while (0)
#endif

template <int WM, int WN, bool A_KMAJ, bool B_KMAJ, bool OUT_F32, bool GLDS, bool F16 = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_mfma_kernel(GemmArgs g, int tiles_m, int tiles_n, int ksplit,
                                                               float* __restrict__ ws) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
  typedef TileGeom<BM, A_KMAJ> GA;
  typedef TileGeom<BN, B_KMAJ> GB;
  constexpr int NVA = GA::NVEC / NT, NVB = GB::NVEC / NT;
  static_assert(GA::NVEC % NT == 0 && GB::NVEC % NT == 0, "tile/threads mismatch");
  constexpr int EA = GLDS ? BM * BK : GA::ELEMS, EB = GLDS ? BN * BK : GB::ELEMS;   // the DMA image is unpadded
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sA[2];
  bf16_t* sB[2];
  sA[0] = reinterpret_cast<bf16_t*>(smem_raw);
  sA[1] = sA[0] + EA;
  sB[0] = sA[1] + EA;
  sB[1] = sB[0] + EB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntiles = tiles_m * tiles_n;
  OFA_TL_BEGIN;
  int t, ks;
  tile_and_slice(ntiles, t, ks);
  // grouped order inside the XCD's run: consecutive ids walk GM tile-rows before moving to the next tile-column, so the
  // ~64 workgroups resident on an XCD cover a GM x (64/GM) block of tiles and every A / B panel they pull into the
  // XCD's 4 MiB L2 is shared by ~8 of them.
  constexpr int GM = 8;
  const int gsz = GM * tiles_n;
  const int gid = t / gsz, first_m = gid * GM;
  const int rows_in_group = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int tm = first_m + (t % gsz) % rows_in_group, tn = (t % gsz) / rows_in_group;
  const int m0 = tm * BM, n0 = tn * BN;
  const int bz = blockIdx.z;
  const bf16_t* A = (const bf16_t*)g.A + batch_off(bz, g.batch_inner, g.strideA, g.strideA2);
  const bf16_t* B = (const bf16_t*)g.B + batch_off(bz, g.batch_inner, g.strideB, g.strideB2);
  const int kbeg = ks * ksplit;
  const int kend = (kbeg + ksplit < g.K) ? kbeg + ksplit : g.K;
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  if constexpr (GLDS) {
    // K (and every split) is a multiple of BK: all tiles are full, rows/columns outside M/N are clamped reads
    const bf16_t* pa[NVA];
    const bf16_t* pb[NVB];
    glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, kbeg, tid, g.a_krows);
    glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, kbeg, tid, g.b_krows);
    const int64_t stepA = A_KMAJ ? BK : (int64_t)BK * g.lda, stepB = B_KMAJ ? BK : (int64_t)BK * g.ldb;
    int knext = kbeg;
    if (nk > 0) {
      glds_issue<NT, NVA>(pa, stepA, sA[0], wave_u);
      glds_issue<NT, NVB>(pb, stepB, sB[0], wave_u);
      knext += BK;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    OFA_TL(1);
    if (nk > 1) {                      // tile 1 travels under the whole of K-step 0
      if (!A_KMAJ && knext + BK > g.a_krows)     // ragged contraction tail: A's missing k rows read as zeros
        glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, knext, tid, g.a_krows);
      if (!B_KMAJ && knext + BK > g.b_krows)     // zero-padded contraction tail: clamp B's k rows
        glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
      glds_issue<NT, NVA>(pa, stepA, sA[1], wave_u);
      glds_issue<NT, NVB>(pb, stepB, sB[1], wave_u);
      knext += BK;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
    FragAddr<BM, A_KMAJ> fax[2];
    FragAddr<BN, B_KMAJ> faw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fax[i].init(lds0, wm * 64 + i * 32, lane);
      faw[i].init(lds0, wn * 64 + i * 32, lane);
    }
    constexpr int OA0 = 0, OA1 = EA * 2, OB0 = 2 * EA * 2, OB1 = 2 * EA * 2 + EB * 2;   // byte offsets of the 4 buffers
    static_assert(OB1 + 3 * 16 * (BM > BN ? BM : BN) * 2 + 4 * (BM > BN ? BM : BN) * 2 < 65536, "ds offset field");
    // The K-step, pinned instruction by instruction (sched_barrier): two complete fragment sets, the reads of k-slice kk+1
    // go out one per MFMA gap of slice kk and are waited for only in front of slice kk+1's first MFMA.  Left to itself
    // hipcc sinks the (side-effect-free) MFMAs below the asm reads so that ONE register set suffices -- every slice then
    // issued its reads behind its own MFMAs and sat out a full LDS round trip with an empty matrix pipe (seen in the ISA;
    // tools/gemm_timeline.py: 165-225 clocks per 4-MFMA slice instead of 128).  The stage barrier sits in front of the
    // LAST slice's MFMAs; behind it that slice multiplies while the first fragments of the next stage are read and the
    // DMA pieces of tile kt+2 are sent for into the stage just retired (prefetch distance: one K-step, as before).
#define OFA_SB __builtin_amdgcn_sched_barrier(0)
#define OFA_WAIT(X, W) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(X[1]), "+v"(W[0]), "+v"(W[1]))
#define OFA_MF(X, W, I, J)                                                                                     \
    acc[I][J] = mfma16<F16>(W[J], X[I], acc[I][J])
#define OFA_SLICE(X, W, XN, WN, KKN, OA, OB)                                        \
    OFA_MF(X, W, 0, 0); frag_issue<BM, A_KMAJ, KKN, OA>(XN[0], fax[0]); OFA_SB;     \
    OFA_MF(X, W, 0, 1); frag_issue<BM, A_KMAJ, KKN, OA>(XN[1], fax[1]); OFA_SB;     \
    OFA_MF(X, W, 1, 0); frag_issue<BN, B_KMAJ, KKN, OB>(WN[0], faw[0]); OFA_SB;     \
    OFA_MF(X, W, 1, 1); frag_issue<BN, B_KMAJ, KKN, OB>(WN[1], faw[1]); OFA_SB
#define OFA_DMA(Q, DST_A, DST_B)                                                                                        \
    static_for<(Q) * (NVA + NVB) / 4, ((Q) + 1) * (NVA + NVB) / 4>([&](auto pc) {                                       \
      constexpr int pi = decltype(pc)::value;                                                                           \
      if constexpr (pi < NVA) {                                                                                         \
        __builtin_amdgcn_global_load_lds((gvoid_t*)pa[pi], (lvoid_t*)(DST_A + (wave_u * 64 + pi * NT) * 8), 16, 0, 0); \
        pa[pi] += stepA;                                                                                                \
      } else {                                                                                                          \
        __builtin_amdgcn_global_load_lds((gvoid_t*)pb[pi - NVA], (lvoid_t*)(DST_B + (wave_u * 64 + (pi - NVA) * NT) * 8), 16, 0, 0); \
        pb[pi - NVA] += stepB;                                                                                          \
      }                                                                                                                 \
    })
#define OFA_KSTEP(OA, OB, OAN, OBN, CUR_A, CUR_B, HAS_NEXT, MORE2)                                     \
    {                                                                                                  \
      OFA_WAIT(x0, w0); OFA_SB;                                                                        \
      OFA_SLICE(x0, w0, x1, w1, 1, OA, OB);                                                            \
      OFA_WAIT(x1, w1); OFA_SB;                                                                        \
      OFA_SLICE(x1, w1, x0, w0, 2, OA, OB);                                                            \
      OFA_WAIT(x0, w0); OFA_SB;                                                                        \
      OFA_SLICE(x0, w0, x1, w1, 3, OA, OB);                                                            \
      OFA_WAIT(x1, w1);                /* every fragment of this stage is in registers */              \
      if (HAS_NEXT) {                                                                                  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  /* tile kt+1 has landed (explicit: hipcc does not see the DMA) */ \
        __syncthreads();               /* ... for every wave, and every wave is done reading this stage */ \
      }                                                                                                \
      const bool more2 = (MORE2);                                                                      \
      if (more2 && !A_KMAJ && knext + BK > g.a_krows)   /* ragged contraction tail: A's missing k rows read as zeros */ \
        glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, knext, tid, g.a_krows);            \
      if (more2 && !B_KMAJ && knext + BK > g.b_krows)   /* zero-padded contraction tail: clamp B's k rows */ \
        glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);                  \
      OFA_SB;                                                                                          \
      /* last slice: its MFMAs cover the next stage's first fragment reads and the DMA issue of tile kt+2 (after the */ \
      /* very last K-step the reads fetch a stale stage; they are waited for and dropped) */            \
      OFA_MF(x1, w1, 0, 0); frag_issue<BM, A_KMAJ, 0, OAN>(x0[0], fax[0]); if (more2) { OFA_DMA(0, CUR_A, CUR_B); } OFA_SB; \
      OFA_MF(x1, w1, 0, 1); frag_issue<BM, A_KMAJ, 0, OAN>(x0[1], fax[1]); if (more2) { OFA_DMA(1, CUR_A, CUR_B); } OFA_SB; \
      OFA_MF(x1, w1, 1, 0); frag_issue<BN, B_KMAJ, 0, OBN>(w0[0], faw[0]); if (more2) { OFA_DMA(2, CUR_A, CUR_B); } OFA_SB; \
      OFA_MF(x1, w1, 1, 1); frag_issue<BN, B_KMAJ, 0, OBN>(w0[1], faw[1]); if (more2) { OFA_DMA(3, CUR_A, CUR_B); } OFA_SB; \
      if (more2) knext += BK;                                                                          \
      OFA_TL_STEP;                                                                                     \
    }
    if (nk > 0) {
      u64x2 x0[2], w0[2], x1[2], w1[2];
      frag_issue<BM, A_KMAJ, 0, OA0>(x0[0], fax[0]);
      frag_issue<BM, A_KMAJ, 0, OA0>(x0[1], fax[1]);
      frag_issue<BN, B_KMAJ, 0, OB0>(w0[0], faw[0]);
      frag_issue<BN, B_KMAJ, 0, OB0>(w0[1], faw[1]);
      int kt = 0;
      for (; kt + 1 < nk; kt += 2) {
        OFA_KSTEP(OA0, OB0, OA1, OB1, sA[0], sB[0], true, (kt + 2 < nk));
        OFA_KSTEP(OA1, OB1, OA0, OB0, sA[1], sB[1], (kt + 2 < nk), (kt + 3 < nk));
      }
      if (kt < nk) OFA_KSTEP(OA0, OB0, OA1, OB1, sA[0], sB[0], false, false);
      OFA_WAIT(x0, w0);                // retire the look-ahead reads before the epilogue re-uses LDS and registers
    }
#undef OFA_KSTEP
#undef OFA_DMA
#undef OFA_SLICE
#undef OFA_MF
#undef OFA_WAIT
#undef OFA_SB
  } else {
  uint4 ra[NVA], rb[NVB];
  if (nk > 0) {
    stage_load<BM, A_KMAJ, NT, NVA>(ra, A, g.lda, m0, g.M, kbeg, kend, tid);
    stage_load<BN, B_KMAJ, NT, NVB>(rb, B, g.ldb, n0, g.N, kbeg, kend, tid);
    stage_store<BM, A_KMAJ, NT, NVA>(ra, sA[0], tid);
    stage_store<BN, B_KMAJ, NT, NVB>(rb, sB[0], tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      stage_load<BM, A_KMAJ, NT, NVA>(ra, A, g.lda, m0, g.M, kbeg + (kt + 1) * BK, kend, tid);
      stage_load<BN, B_KMAJ, NT, NVB>(rb, B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK, kend, tid);
    }
    const bf16_t* a_s = sA[cur];
    const bf16_t* b_s = sB[cur];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 fx[2], fw[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fx[i] = load_frag<BM, A_KMAJ>(a_s, wm * 64 + i * 32, kk, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[j] = load_frag<BN, B_KMAJ>(b_s, wn * 64 + j * 32, kk, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = mfma16<F16>(fw[j], fx[i], acc[i][j]);  // D[n][m]
    }
    if (more) {
      stage_store<BM, A_KMAJ, NT, NVA>(ra, sA[cur ^ 1], tid);
      stage_store<BN, B_KMAJ, NT, NVB>(rb, sB[cur ^ 1], tid);
    }
    __syncthreads();
  }
  }

  // epilogue through the wave's slice of the (idle) LDS stages: see epilogue_lds
  __syncthreads();                                     // every wave is done with the fragment reads / the last DMA
  OFA_TL(2);
  {
    const bool split = gridDim.y > 1;
    constexpr int REGION = ((GLDS ? 2 * (EA + EB) * 2 : 2 * (GA::ELEMS + GB::ELEMS) * 2) / (WM * WN)) & ~1023;
    unsigned char* wl = smem_raw + wave * REGION;
    const int m_w = m0 + wm * 64, n_w = n0 + wn * 64;
    if (split) {
      const int64_t n4 = (g.N + 3) & ~3;
      float* wsb = ws + ((int64_t)bz * gridDim.y + ks) * g.M * n4;
      epilogue_lds<2, 2, true, true, F16>(g, acc, wl, REGION, wsb, n4, m_w, n_w, lane);
    } else {
      const int64_t coff = batch_off(bz, g.batch_inner, g.strideC, g.strideC2);
      void* Cb = OUT_F32 ? (void*)((float*)g.C + coff) : (void*)((bf16_t*)g.C + coff);
      epilogue_lds<2, 2, OUT_F32, false, F16>(g, acc, wl, REGION, Cb, g.ldc, m_w, n_w, lane);
    }
  }
  OFA_TL_END;
}


// ---------------------------------------------------------------------------------------------------------------
// Ring-staged kernel for SMALL grids (<= ~2 workgroups per CU: every decoder-side product of the model, 2048 rows).
// With one or two waves per SIMD nothing hides the global->LDS latency of the double-buffered loop above: a K-step
// computes for ~0.35 us and then waits ~1 us for the next tile (2048 x 768 x 768: 12 K-steps, 13.7 us at every tile
// size -- the latency chain, not the flops).  Here four stages are in flight: tiles kt+1..kt+3 are already travelling
// while tile kt multiplies; the wait in front of the stage barrier is `vmcnt(2 * pieces)` -- only the OLDEST tile has to
// have landed.  Same tile geometry, swizzle, fragment reads and epilogue as gemm_mfma_kernel (LDS-DMA path: whole K
// tiles); the stage select is an add on the fragment address registers instead of an offset immediate.
// TMW x TNW: MFMA tiles (32 x 32) per wave.  2 x 2 is the 64 x 64 wave block of the other kernels.  Round 4: the 64 x 64 and 64 x 128
// WORKGROUP tiles of the decoder-side products run as FOUR waves of 32 x 32 / 32 x 64 (1 x 1 / 1 x 2) instead of one / two waves of
// 64 x 64: a K-step's LDS-DMA pieces (16 KiB = 16 wave-instructions at ~60 issue cycles each for a 64 x 64 tile) were all issued
// by ONE wave -- ~1000 cycles per K-step beside 512 MFMA-cycles; four waves issue four pieces each.
template <int WM, int WN, bool A_KMAJ, bool B_KMAJ, bool OUT_F32, int S = 4, bool F16 = false, int TMW = 2, int TNW = 2>
__global__ __launch_bounds__(WM* WN * 64) void gemm_ring_kernel(GemmArgs g, int tiles_m, int tiles_n, int ksplit,
                                                               float* __restrict__ ws) {
  constexpr int BM = 32 * TMW * WM, BN = 32 * TNW * WN, NT = 64 * WM * WN;
  static_assert(S == 3 || S == 4, "ring depth");
  typedef TileGeom<BM, A_KMAJ> GA;
  typedef TileGeom<BN, B_KMAJ> GB;
  constexpr int NVA = GA::NVEC / NT, NVB = GB::NVEC / NT, P = NVA + NVB;
  constexpr int EA = BM * BK, EB = BN * BK;
  constexpr uint32_t STG = (uint32_t)(EA + EB) * 2u;                  // bytes per stage: [A tile | B tile]
  static_assert((S - 2) * P <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntiles = tiles_m * tiles_n;
  int t, ks;
  tile_and_slice(ntiles, t, ks);
  constexpr int GM = 8;
  const int gsz = GM * tiles_n;
  const int gid = t / gsz, first_m = gid * GM;
  const int rows_in_group = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int tm = first_m + (t % gsz) % rows_in_group, tn = (t % gsz) / rows_in_group;
  const int m0 = tm * BM, n0 = tn * BN;
  const int bz = blockIdx.z;
  const bf16_t* A = (const bf16_t*)g.A + batch_off(bz, g.batch_inner, g.strideA, g.strideA2);
  const bf16_t* B = (const bf16_t*)g.B + batch_off(bz, g.batch_inner, g.strideB, g.strideB2);
  const int kbeg = ks * ksplit;
  const int kend = (kbeg + ksplit < g.K) ? kbeg + ksplit : g.K;
  const int nk = (kend - kbeg) / BK;                                   // launcher guarantees whole K tiles

  f32x16 acc[TMW][TNW];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bf16_t* pa[NVA];
  const bf16_t* pb[NVB];
  glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, kbeg, tid, g.a_krows);
  glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, kbeg, tid, g.b_krows);
  const int64_t stepA = A_KMAJ ? BK : (int64_t)BK * g.lda, stepB = B_KMAJ ? BK : (int64_t)BK * g.ldb;
  int knext = kbeg;
  auto dma = [&](int stage) {
    if (!A_KMAJ && knext + BK > g.a_krows)                            // ragged contraction tail: A's missing k rows read as zeros
      glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, knext, tid, g.a_krows);
    if (!B_KMAJ && knext + BK > g.b_krows)                            // zero-padded contraction tail: clamp B's k rows
      glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
    bf16_t* st = reinterpret_cast<bf16_t*>(smem_raw + (size_t)stage * STG);
    glds_issue<NT, NVA>(pa, stepA, st, wave_u);
    glds_issue<NT, NVB>(pb, stepB, st + EA, wave_u);
    knext += BK;
  };
  int issued = 0;
  for (; issued < S - 1 && issued < nk; ++issued) dma(issued);

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  FragAddr<BM, A_KMAJ> fax[TMW];
  FragAddr<BN, B_KMAJ> faw[TNW];
#pragma unroll
  for (int i = 0; i < TMW; ++i) fax[i].init(lds0, wm * 32 * TMW + i * 32, lane);
#pragma unroll
  for (int j = 0; j < TNW; ++j) faw[j].init(lds0 + (uint32_t)EA * 2u, wn * 32 * TNW + j * 32, lane);
  constexpr int NA = A_KMAJ ? 4 : 1, NB = B_KMAJ ? 4 : 1;
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int later = issued - kt - 1;                                 // tiles issued behind tile kt (block-uniform)
    if (S >= 4 && later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
    else if (later >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the bare barrier instruction: __syncthreads() is fence + barrier and hipcc materialises the fence as
    // `s_waitcnt vmcnt(0)` -- every tile in flight would be drained at every step (seen in the ISA; the ring then ran
    // SLOWER than two stages).  What the barrier has to order is covered explicitly: this wave's tile kt by the vmcnt
    // above, its fragment reads of tile kt-1 by the lgkmcnt waits in front of the MFMAs.
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                       // tile kt is complete; everybody is done reading tile kt-1
    __builtin_amdgcn_sched_barrier(0);
    // refill the stage tile kt-1 lived in -- ONE QUARTER of the tile's DMA pieces behind each k-slice's MFMAs: a piece
    // costs the wave ~60 issue cycles, and with one wave per SIMD (small grids) a burst of P pieces in front of the
    // MFMAs is as long as the MFMAs themselves (the ring tolerates the later issue: the tile is not needed for 3 steps)
    const bool refill = issued < nk;
    bf16_t* rst = reinterpret_cast<bf16_t*>(smem_raw + (size_t)((stage + S - 1) % S) * STG);
    if (refill && !A_KMAJ && knext + BK > g.a_krows)
      glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, knext, tid, g.a_krows);
    if (refill && !B_KMAJ && knext + BK > g.b_krows)
      glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
    u64x2 x0[TMW], w0[TNW], x1[TMW], w1[TNW];
#define RING_ISSUE(KK, X, W)                                                                                           \
    static_for<0, TMW>([&](auto ic) { frag_issue<BM, A_KMAJ, KK, 0>(X[decltype(ic)::value], fax[decltype(ic)::value]); }); \
    static_for<0, TNW>([&](auto jc) { frag_issue<BN, B_KMAJ, KK, 0>(W[decltype(jc)::value], faw[decltype(jc)::value]); })
    // (the wait statement names every fragment register: that is what orders the MFMAs behind it)
#define RING_WAIT(X, W)                                                                                                \
    if constexpr (TMW == 2 && TNW == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(X[TMW - 1]), "+v"(W[0]), "+v"(W[TNW - 1])); \
    else if constexpr (TMW == 1 && TNW == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(W[0]), "+v"(W[TNW - 1]));            \
    else if constexpr (TMW == 2 && TNW == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(X[TMW - 1]), "+v"(W[0]));            \
    else { static_assert(TMW <= 2 && TNW <= 2, "RING_WAIT names every fragment register: add the shape");              \
           asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(W[0])); }
#define RING_MMA(X, W)                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TMW; ++i) _Pragma("unroll") for (int j = 0; j < TNW; ++j)                    \
        acc[i][j] = mfma16<F16>(W[j], X[i], acc[i][j])
#define RING_DMA(Q)                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if (refill) {                                                                                                      \
      static_for<(Q) * P / 4, ((Q) + 1) * P / 4>([&](auto pc) {                                                        \
        constexpr int pi = decltype(pc)::value;                                                                        \
        if constexpr (pi < NVA) {                                                                                      \
          __builtin_amdgcn_global_load_lds((gvoid_t*)pa[pi], (lvoid_t*)(rst + (wave_u * 64 + pi * NT) * 8), 16, 0, 0); \
          pa[pi] += stepA;                                                                                             \
        } else {                                                                                                       \
          __builtin_amdgcn_global_load_lds((gvoid_t*)pb[pi - NVA], (lvoid_t*)(rst + EA + (wave_u * 64 + (pi - NVA) * NT) * 8), 16, 0, 0); \
          pb[pi - NVA] += stepB;                                                                                       \
        }                                                                                                              \
      });                                                                                                              \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0)
    RING_ISSUE(0, x0, w0);
    RING_WAIT(x0, w0);
    RING_ISSUE(1, x1, w1);
    RING_MMA(x0, w0);
    RING_DMA(0);
    RING_WAIT(x1, w1);
    RING_ISSUE(2, x0, w0);
    RING_MMA(x1, w1);
    RING_DMA(1);
    RING_WAIT(x0, w0);
    RING_ISSUE(3, x1, w1);
    RING_MMA(x0, w0);
    RING_DMA(2);
    RING_WAIT(x1, w1);
    RING_MMA(x1, w1);
    RING_DMA(3);
    if (refill) {
      knext += BK;
      ++issued;
    }
#undef RING_DMA
#undef RING_ISSUE
#undef RING_WAIT
#undef RING_MMA
    // next stage: wrap the fragment addresses around the ring
    const bool wrap = stage == S - 1;
    const uint32_t delta = wrap ? (uint32_t)(0u - (uint32_t)(S - 1) * STG) : STG;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int q = 0; q < NA; ++q) fax[i].a[q] += delta;
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int q = 0; q < NB; ++q) faw[j].a[q] += delta;
    stage = wrap ? 0 : stage + 1;
  }
  __syncthreads();                                     // every wave is done with the fragment reads
  {
    const bool split = gridDim.y > 1;
    constexpr int REGION = ((int)(S * STG) / (WM * WN)) & ~1023;
    unsigned char* wl = smem_raw + wave * REGION;
    const int m_w = m0 + wm * 32 * TMW, n_w = n0 + wn * 32 * TNW;
    if (split) {
      const int64_t n4 = (g.N + 3) & ~3;
      float* wsb = ws + ((int64_t)bz * gridDim.y + ks) * g.M * n4;
      epilogue_lds<TMW, TNW, true, true, F16>(g, acc, wl, REGION, wsb, n4, m_w, n_w, lane);
    } else {
      const int64_t coff = batch_off(bz, g.batch_inner, g.strideC, g.strideC2);
      void* Cb = OUT_F32 ? (void*)((float*)g.C + coff) : (void*)((bf16_t*)g.C + coff);
      epilogue_lds<TMW, TNW, OUT_F32, false, F16>(g, acc, wl, REGION, Cb, g.ldc, m_w, n_w, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Big-tile kernel: a WGM x WGN grid of waves, each owning a (32*TM) x (32*TN) block of accumulators (TM x TN MFMA tiles), ONE
// workgroup per CU (128 KiB of LDS: 2 stages x (A 32 KiB + B 32 KiB)).  Shipped form: eight waves in a 2 x 4 grid of 128 x 64
// (or 96 x 64) blocks -- a 256 x 256 (192 x 256) tile, two waves per SIMD, 128 accumulator registers per lane.  Why the big
// block: with 64x64 per wave every K-step moves 16 KiB of fragments per wave out of LDS for 16 MFMAs; 128 x 64 needs 24
// fragment reads per 32 MFMAs.  Why two waves per SIMD: round 1's form of this kernel (four waves of 128 x 128, one per
// SIMD) had nothing to run while a wave sat at the stage barrier or waited for fragments (1.47-1.74 us per K-step against
// 1.42-1.55 with eight waves, and a 2.5 instead of 5.8 us epilogue: profiles/round2_gemm_timeline.txt section C).
// The overlap inside a wave is built into its instruction stream: fragments of k-slice kk+1 are read while slice kk
// multiplies, the barrier that retires an LDS stage sits in front of the LAST slice's MFMAs, and the DMA of tile t+2 is
// issued right behind it, one piece per MFMA gap.


// (t, ks, bz): output tile, K-slice and batch index of this workgroup; nsplit > 1 or to_ws: the raw fp32 tile goes to
// slab ks of ws ([nsplit][M][(N+3)&~3] per batch) instead of C
template <int TM, int TN, bool A_KMAJ, bool B_KMAJ, bool OUT_F32, int WGM, int WGN, bool F16>
__device__ __forceinline__ void gemm_big_body(const GemmArgs& g, int tiles_m, int tiles_n, int ksplit,
                                              float* __restrict__ ws, int t, int ks, int bz, int nsplit, bool to_ws) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, NT = 64 * WGM * WGN;
  static_assert(BM <= 256 && BN <= 256, "operand stage is 256 rows");
  static_assert(A_KMAJ || BM == 256, "m-major tiles are 256 wide (swizzle)");
  static_assert(B_KMAJ || BN == 256, "m-major tiles are 256 wide (swizzle)");
  constexpr int NVA = BM * 8 / NT, NVB = BN * 8 / NT;
  constexpr int STAGE = 256 * BK;                           // elements per operand stage (32 KiB, also for BM = 192):
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // a power of two, so "other stage" is addr ^ 32768
  bf16_t* sA0 = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* sA1 = sA0 + STAGE;
  bf16_t* sB0 = sA1 + STAGE;
  bf16_t* sB1 = sB0 + STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  OFA_TL_BEGIN;
  constexpr int GM = 8;
  const int gsz = GM * tiles_n;
  const int gid = t / gsz, first_m = gid * GM;
  const int rows_in_group = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int tm = first_m + (t % gsz) % rows_in_group, tn = (t % gsz) / rows_in_group;
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16_t* A = (const bf16_t*)g.A + batch_off(bz, g.batch_inner, g.strideA, g.strideA2);
  const bf16_t* B = (const bf16_t*)g.B + batch_off(bz, g.batch_inner, g.strideB, g.strideB2);
  const int kbeg = ks * ksplit;
  const int kend = (kbeg + ksplit < g.K) ? kbeg + ksplit : g.K;
  const int nk = (kend - kbeg) / BK;                        // launcher guarantees whole K tiles

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bf16_t* pa[NVA];
  const bf16_t* pb[NVB];
  glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, kbeg, tid, g.a_krows);
  glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, kbeg, tid, g.b_krows);
  const int64_t stepA = A_KMAJ ? BK : (int64_t)BK * g.lda, stepB = B_KMAJ ? BK : (int64_t)BK * g.ldb;
  int knext = kbeg;
  auto dma = [&](bf16_t* da, bf16_t* db) {
    if (!A_KMAJ && knext + BK > g.a_krows)                  // ragged contraction tail: A's missing k rows read as zeros
      glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, knext, tid, g.a_krows);
    if (!B_KMAJ && knext + BK > g.b_krows)                  // zero-padded contraction tail: clamp B's k rows
      glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
    glds_issue<NT, NVA>(pa, stepA, da, wave_u);
    glds_issue<NT, NVB>(pb, stepB, db, wave_u);
    knext += BK;
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  BigAddr<BM, A_KMAJ> fax;
  BigAddr<BN, B_KMAJ> faw;
  fax.init(lds0, wm * TM * 32, lane);
  faw.init(lds0 + 2 * STAGE * 2, wn * TN * 32, lane);

  u64x2 xa[2][TM], wb[2][TN];
#ifndef OFA_BIG_RPG_TN
#define OFA_BIG_RPG_TN 1
#endif
  // fragment reads woven behind each MFMA.  For the all-m-major (weight-gradient) form, whose fragments are two transposing
  // reads each, issuing them earlier (2 or 3 fragments per gap) is slower: 216 / 221 / 230 us for the encoder layer's group
  // (tools/gemm_group_bench.py) -- the LDS pipe wants them spread out
  constexpr int RPG = (!A_KMAJ && !B_KMAJ) ? OFA_BIG_RPG_TN : 1;
#define BIG_ISSUE(KK, SET)                                                                                        \
  static_for<0, TM>([&](auto ic) { big_frag<BM, A_KMAJ, KK, decltype(ic)::value, 0>(xa[SET][decltype(ic)::value], fax); }); \
  static_for<0, TN>([&](auto ic) { big_frag<BN, B_KMAJ, KK, decltype(ic)::value, 0>(wb[SET][decltype(ic)::value], faw); })
#define BIG_WAITCNT "s_waitcnt lgkmcnt(0)"
#define BIG_WAIT(SET)                                                                  \
  if constexpr (TM == 4 && TN == 2)                                                    \
    asm volatile(BIG_WAITCNT : "+v"(xa[SET][0]), "+v"(xa[SET][1]), "+v"(xa[SET][2]), "+v"(xa[SET][3]),   \
                 "+v"(wb[SET][0]), "+v"(wb[SET][TN - 1]));                                                          \
  else if constexpr (TM == 3 && TN == 2)                                               \
    asm volatile(BIG_WAITCNT : "+v"(xa[SET][0]), "+v"(xa[SET][1]), "+v"(xa[SET][2]),                     \
                 "+v"(wb[SET][0]), "+v"(wb[SET][TN - 1]));                                                          \
  else if constexpr (TM == 4 && TN == 4)                                               \
    asm volatile(BIG_WAITCNT : "+v"(xa[SET][0]), "+v"(xa[SET][1]), "+v"(xa[SET][2]), "+v"(xa[SET][3]),   \
                 "+v"(wb[SET][0]), "+v"(wb[SET][1]), "+v"(wb[SET][2]), "+v"(wb[SET][TN - 1]));                      \
  else                                                                                 \
    static_assert(TM == 4 && TN == 2, "BIG_WAIT names every fragment register: add the shape")
  // One k-slice: TM*TN MFMAs on fragment set SET, with the NEXT slice's fragment reads (k-slice KKN into the other set)
  // woven in between them.  Issuing the 8 reads in one burst in front of the MFMAs serialises the two: all four waves
  // run in lockstep behind the stage barrier, their 32 reads queue up in the LDS pipe and the in-order wave cannot reach
  // its MFMAs until its own reads are accepted (measured: 1272 TF with burst reads vs 1990 TF with no reads at all).
  // sched_barrier pins the order; without it hipcc hoists the (dependence-free) MFMAs above the asm reads.
#define BIG_SB __builtin_amdgcn_sched_barrier(0)
#define BIG_SLICE(SET, KKN)                                                                                         \
  static_for<0, TM * TN>([&](auto tc) {                                                                             \
    constexpr int t = decltype(tc)::value, i = t / TN, j = t % TN;                                                  \
    acc[i][j] = mfma16<F16>(wb[SET][j], xa[SET][i], acc[i][j]); \
    static_for<0, RPG>([&](auto qc) {                                                                               \
      constexpr int r = big_read_after(t * RPG + decltype(qc)::value, TM + TN);                                     \
      if constexpr (r >= 0 && r < TM) big_frag<BM, A_KMAJ, KKN, (r < TM ? r : 0), 0>(xa[1 - (SET)][r < TM ? r : 0], fax); \
      if constexpr (r >= TM) big_frag<BN, B_KMAJ, KKN, (r >= TM ? r - TM : 0), 0>(wb[1 - (SET)][r >= TM ? r - TM : 0], faw); \
    });                                                                                                             \
    BIG_SB;                                                                                                         \
  })
  if (nk > 0) {
    dma(sA0, sB0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    OFA_TL(1);
    if (nk > 1) dma(sA1, sB1);
    BIG_ISSUE(0, 0);
    bf16_t* curA = sA0;                // the stage being multiplied; its buffers are refilled with tile kt+2
    bf16_t* curB = sB0;
    // ONE copy of the K-step (the stage toggle is an XOR on the 8 fragment address registers, not an unrolled immediate):
    // with 256 accumulator registers live, any second copy of the loop body makes hipcc spill at the joins
    for (int kt = 0; kt < nk; ++kt) {
      BIG_WAIT(0); BIG_SB; BIG_SLICE(0, 1);
      BIG_WAIT(1); BIG_SB; BIG_SLICE(1, 2);
      BIG_WAIT(0); BIG_SB; BIG_SLICE(0, 3);
      BIG_WAIT(1);
      if (kt + 1 < nk) {
        // tile kt+1 has landed and every wave is done reading this stage.  The vmcnt wait is explicit: with the DMA
        // builtins inside conditional blocks hipcc emitted only lgkmcnt(0) in front of this barrier (seen in the ISA;
        // wrong results under load, when a piece takes longer than one K-step to land)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      const bool more2 = kt + 2 < nk;     // refill the retired stage with tile kt+2 ...
      if (more2 && !A_KMAJ && knext + BK > g.a_krows)          // ragged contraction tail: A's missing k rows read as zeros
        glds_ptrs<BM, A_KMAJ, NT, NVA, true>(pa, A, g.lda, m0, g.M, knext, tid, g.a_krows);
      if (more2 && !B_KMAJ && knext + BK > g.b_krows)          // zero-padded contraction tail: clamp B's k rows
        glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fax.a[i] ^= (uint32_t)(STAGE * 2);
        faw.a[i] ^= (uint32_t)(STAGE * 2);
      }
      BIG_SB;
      // ... one LDS-DMA piece per MFMA gap: a piece costs the wave ~60 issue cycles, and 16 of them in one burst in
      // front of the MFMAs left the matrix pipe idle for ~1000 cycles per K-step (GRBM cycles 1.83M vs 1.31M without DMA).
      // (after the last K-step the reads below fetch a stale stage; they are waited for and dropped)
      static_for<0, TM * TN>([&](auto tc) {
        constexpr int t = decltype(tc)::value, i = t / TN, j = t % TN;
        acc[i][j] = mfma16<F16>(wb[1][j], xa[1][i], acc[i][j]);
        static_for<0, RPG>([&](auto qc) {
          constexpr int r = big_read_after(t * RPG + decltype(qc)::value, TM + TN);
          if constexpr (r >= 0 && r < TM) big_frag<BM, A_KMAJ, 0, (r < TM ? r : 0), 0>(xa[0][r < TM ? r : 0], fax);
          if constexpr (r >= TM) big_frag<BN, B_KMAJ, 0, (r >= TM ? r - TM : 0), 0>(wb[0][r >= TM ? r - TM : 0], faw);
        });
        if (more2) {
          constexpr int NP = NVA + NVB, NM = TM * TN;
          static_for<t * NP / NM, (t + 1) * NP / NM>([&](auto pc) {
            constexpr int pi = decltype(pc)::value;
            if constexpr (pi < NVA) {
              __builtin_amdgcn_global_load_lds((gvoid_t*)pa[pi], (lvoid_t*)(curA + (wave_u * 64 + pi * NT) * 8), 16, 0, 0);
              pa[pi] += stepA;
            } else {
              __builtin_amdgcn_global_load_lds((gvoid_t*)pb[pi - NVA], (lvoid_t*)(curB + (wave_u * 64 + (pi - NVA) * NT) * 8), 16, 0, 0);
              pb[pi - NVA] += stepB;
            }
          });
        }
        BIG_SB;
      });
      if (more2) knext += BK;
      OFA_TL_STEP;
      curA = (bf16_t*)((uintptr_t)curA ^ (uintptr_t)(STAGE * 2));
      curB = (bf16_t*)((uintptr_t)curB ^ (uintptr_t)(STAGE * 2));
    }
    BIG_WAIT(0);
  }
#undef BIG_SLICE
#undef BIG_SB
#undef BIG_WAIT
#undef BIG_ISSUE
  __syncthreads();                                     // every wave is done with the fragment reads
  OFA_TL(2);
  {
    const bool split = to_ws || nsplit > 1;
    constexpr int REGION = 4 * STAGE * 2 / (WGM * WGN);   // 32 KiB per wave (16 KiB with eight waves)
    unsigned char* wl = smem_raw + wave * REGION;
    const int m_w = m0 + wm * TM * 32, n_w = n0 + wn * TN * 32;
    if (split) {
      const int64_t n4 = (g.N + 3) & ~3;
      float* wsb = ws + ((int64_t)bz * nsplit + ks) * g.M * n4;
      epilogue_lds<TM, TN, true, true, F16>(g, acc, wl, REGION, wsb, n4, m_w, n_w, lane);
    } else {
      const int64_t coff = batch_off(bz, g.batch_inner, g.strideC, g.strideC2);
      void* Cb = OUT_F32 ? (void*)((float*)g.C + coff) : (void*)((bf16_t*)g.C + coff);
      epilogue_lds<TM, TN, OUT_F32, false, F16>(g, acc, wl, REGION, Cb, g.ldc, m_w, n_w, lane);
    }
  }
  OFA_TL_END;
}

template <int TM, int TN, bool A_KMAJ, bool B_KMAJ, bool OUT_F32, int WGM = 2, int WGN = 2, bool F16 = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_big_kernel(GemmArgs g, int tiles_m, int tiles_n, int ksplit,
                                                                 float* __restrict__ ws) {
  int t, ks;
  tile_and_slice(tiles_m * tiles_n, t, ks);
  gemm_big_body<TM, TN, A_KMAJ, B_KMAJ, OUT_F32, WGM, WGN, F16>(g, tiles_m, tiles_n, ksplit, ws, t, ks, (int)blockIdx.z,
                                                                (int)gridDim.y, false);
}

// Two tile heights in ONE launch, against round quantisation.  The big-tile grid is counted in rounds of the 256 CUs: the FFN's first
// product (13312 x 3072 x 768) is 624 tiles of 256 x 256 = 2.44 rounds and takes three (76 us where the loop's own rate says 62; the
// vendor kernel, which balances with stream-K, 68).  Here the first `rows_big` row tiles are 256 rows high and the remaining rows are cut
// into `rows_small` tiles of 192 -- 16 x 12 + 48 x 12 = 768 workgroups = exactly three per CU, 4/5 of them the short kind -- chosen by
// gemm_plan from a simulation of the dispatch (mixed_time_us).  Every XCD gets an eighth of EACH kind (the hardware deals workgroups to
// the XCDs round-robin; a contiguous eighth of a list sorted by height would give two XCDs all the tall tiles), tall ones first.  Same
// main loop, same per-element summation order as gemm_big_kernel: bit-identical results.  k-major A only (an m-major A half is 128 wide).
template <bool B_KMAJ, bool OUT_F32, bool F16>
__global__ __launch_bounds__(512) void gemm_big_mixed_kernel(GemmArgs g, int tiles_n, int rows_big, int rows_small) {
  const int nbig = rows_big * tiles_n, nsmall = rows_small * tiles_n;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  auto lo = [](int n, int x) { const int q = n >> 3, r = n & 7; return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q; };
  const int b0 = lo(nbig, xcd), nb = lo(nbig, xcd + 1) - b0, s0 = lo(nsmall, xcd), ns = lo(nsmall, xcd + 1) - s0;
  if (local < nb) {
    gemm_big_body<4, 2, true, B_KMAJ, OUT_F32, 2, 4, F16>(g, rows_big, tiles_n, g.K, nullptr, b0 + local, 0, 0, 1, false);
  } else if (local - nb < ns) {
    const int64_t r0 = (int64_t)rows_big * 256;                          // the short tiles start below the tall ones
    g.A = (const bf16_t*)g.A + r0 * g.lda;
    g.C = OUT_F32 ? (void*)((float*)g.C + r0 * g.ldc) : (void*)((bf16_t*)g.C + r0 * g.ldc);
    g.M -= (int)r0;
    gemm_big_body<3, 2, true, B_KMAJ, OUT_F32, 2, 4, F16>(g, rows_small, tiles_n, g.K, nullptr, s0 + local - nb, 0, 0, 1, false);
  }
}

// estimated time of that launch: per XCD its eighth of the tall and of the short tiles, handed to its 32 CUs in order as they free up
// (tile times from gemm_plan's fitted model: K-steps x 1.45 + 7.6 us for 256 x 256, x 1.13 + 6.5 for 192 x 256)
static double mixed_time_us(int nbig, int nsmall, int nk) {
  const double t4 = nk * 1.45 + 7.6, t3 = nk * 1.13 + 6.5;
  double worst = 0.0;
  for (int x = 0; x < 8; ++x) {
    auto lo = [](int n, int x_) { const int q = n >> 3, r = n & 7; return x_ < r ? x_ * (q + 1) : r * (q + 1) + (x_ - r) * q; };
    const int nb = lo(nbig, x + 1) - lo(nbig, x), ns = lo(nsmall, x + 1) - lo(nsmall, x);
    double cu[32];
    for (int c = 0; c < 32; ++c) cu[c] = 0.0;
    for (int i = 0; i < nb + ns; ++i) {
      int best = 0;
      for (int c = 1; c < 32; ++c) best = cu[c] < cu[best] ? c : best;
      cu[best] += i < nb ? t4 : t3;
    }
    for (int c = 0; c < 32; ++c) worst = cu[c] > worst ? cu[c] : worst;
  }
  return worst;
}

template <bool BKM, bool OF, bool F16>
static void launch_big_mixed(const GemmArgs& g, int rows_big, int rows_small, hipStream_t st) {
  const int tiles_n = cdiv(g.N, 256), nbig = rows_big * tiles_n, nsmall = rows_small * tiles_n;
  const size_t lds = 4 * (size_t)256 * BK * sizeof(bf16_t);
  auto kern = gemm_big_mixed_kernel<BKM, OF, F16>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int per_xcd = cdiv(nbig, 8) + cdiv(nsmall, 8);              // (an XCD's list is at most this long; shorter lists leave idle workgroups)
  hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(512), lds, st, g, tiles_n, rows_big, rows_small);
}

// Grouped weight-gradient products (ofa_gemm_group_tn): up to GROUP_MAX independent  slabs_p[s] = A_p^T B_p over K-slice s
// in ONE launch of 256 x 256 eight-wave tiles.  A layer's weight gradients are each 9-36 such tiles: alone, a product has
// to be cut into 3-7 K-slices of 128 x 128 tiles to occupy the chip (short K loops, per-tile overheads every ~30 K-steps,
// a 3-7 slab reduce); together they fill the 256 CUs with ~2 slices each and K loops of ~100 steps.
// (GroupItem / GroupArgs: gemm_core.h)

template <bool F16, bool W4 = false>
__global__ __launch_bounds__(W4 ? 256 : 512) void gemm_group_tn_kernel(GroupArgs ga) {
  GemmArgs g;
  int t, ks;
  const GroupItem* itp = group_enter(ga, g, t, ks);
  if (!itp) return;
  const GroupItem& it = *itp;
  // (OUT_F32 = false: the non-slab epilogue of this instantiation is the 16-bit accumulate of a one-slice product)
  if constexpr (W4) gemm_big_body<4, 4, false, false, false, 2, 2, F16>(g, it.tiles_m, it.tiles_n, it.ksplit, it.ws, t, ks, 0, it.splits, it.out == nullptr);
  else gemm_big_body<4, 2, false, false, false, 2, 4, F16>(g, it.tiles_m, it.tiles_n, it.ksplit, it.ws, t, ks, 0, it.splits, it.out == nullptr);
}

template <bool OUT_F32, bool F16 = false>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, const float* __restrict__ ws, int splits) {
  const int64_t quads = (int64_t)g.M * ((g.N + 3) / 4);
  const int bz = blockIdx.y;
  const int64_t coff = batch_off(bz, g.batch_inner, g.strideC, g.strideC2);
  void* Cb = OUT_F32 ? (void*)((float*)g.C + coff) : (void*)((bf16_t*)g.C + coff);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / ((g.N + 3) / 4)), n = (int)(i % ((g.N + 3) / 4)) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 p = *reinterpret_cast<const float4*>(ws + (((int64_t)bz * splits + k) * g.M + m) * ((g.N + 3) & ~3) + n);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    epilogue_store<OUT_F32, F16>(g, Cb, m, n, s.x, s.y, s.z, s.w);
  }
}

bool gemm_mfma_supported(const GemmArgs& g) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return false;
  // Leading dimensions are 16-byte multiples; logical sizes may be ragged as long as the padded vector/quad stays
  // inside the row (ld): e.g. logits [rows, V=51265] stored with ld = 51272 (adaptor/text.py:129-142 output).
  const int64_t M8 = (g.M + 7) & ~7, N8 = (g.N + 7) & ~7, N4 = (g.N + 3) & ~3, K8 = (g.K + 7) & ~7;
  if ((g.lda & 7) || (g.ldb & 7) || (g.ldc & 3) || N4 > g.ldc) return false;
  if ((g.strideA & 7) || (g.strideB & 7) || (g.strideC & 3)) return false;
  if ((g.strideA2 & 7) || (g.strideB2 & 7) || (g.strideC2 & 3)) return false;
  if (!g.transA && (g.K & 7)) {              // k-major A with a ragged K: only against an m-major B (whose k rows are
    if (g.transB || !(g.flags & OFA_GEMM_A_KPAD_ZERO) || K8 > g.lda) return false;   // exact) and a zero row tail
  }
  if (g.transB && (g.K & 7)) return false;   // k-major B: vectors along k must be whole
  if (g.transA && M8 > g.lda) return false;  // m-major A: vectors along m
  if (!g.transB && N8 > g.ldb) return false; // m-major B: vectors along n
  if ((g.flags & OFA_GEMM_BIAS_COL) && (g.N & 3)) return false;
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15)) return false;
  if ((g.flags & OFA_GEMM_BIAS_COL) && ((uintptr_t)g.bias & 7)) return false;
  return true;
}

template <int WM, int WN, bool AK, bool BKM, bool OF, bool GL, bool F16 = false>
static void launch_cfg2(const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  const size_t lds = GL ? 2 * (size_t)(BM + BN) * BK * sizeof(bf16_t)
                        : 2 * (size_t)(TileGeom<BM, AK>::ELEMS + TileGeom<BN, BKM>::ELEMS) * sizeof(bf16_t);
  auto kern = gemm_mfma_kernel<WM, WN, AK, BKM, OF, GL, F16>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(tiles_m * tiles_n, splits, batch), block(WM * WN * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, st, g, tiles_m, tiles_n, ksplit, ws);
}

template <int WM, int WN, bool AK, bool BKM, bool OF, int S = 4, bool F16 = false, int TMW = 2, int TNW = 2>
static void launch_ring(const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  constexpr int BM = 32 * TMW * WM, BN = 32 * TNW * WN;
  const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  const size_t lds = S * (size_t)(BM + BN) * BK * sizeof(bf16_t);
  auto kern = gemm_ring_kernel<WM, WN, AK, BKM, OF, S, F16, TMW, TNW>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(tiles_m * tiles_n, splits, batch), block(WM * WN * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, st, g, tiles_m, tiles_n, ksplit, ws);
}

// small grids take the 4-stage ring (see gemm_ring_kernel)
static bool use_ring(const GemmArgs& g, int wm, int wn, int64_t blocks, int ksplit) {
  if ((g.K % BK) != 0 || (ksplit % BK) != 0 || (g.flags & OFA_GEMM_NO_LDS_DMA)) return false;
  const int64_t cap = (wm == 1 && wn == 1) ? 512 : 256;              // 64 KiB of LDS: two per CU; 96 / 128 KiB: one
  return blocks <= cap && ksplit >= 4 * BK;
}

template <int WM, int WN, bool AK, bool BKM, bool OF, bool F16 = false>
static void launch_cfg(const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  if (use_ring(g, WM, WN, (int64_t)cdiv(g.M, 64 * WM) * cdiv(g.N, 64 * WN) * splits * batch, ksplit)) {
    // the 64 x 64 / 64 x 128 workgroup tiles run as FOUR waves of 32 x 32 / 32 x 64 (see gemm_ring_kernel): in the replayed steps
    // cfg-2 12.18 -> 12.07 ms, cfg-5 (every product is a small grid) 135.6 -> 130.5 ms, same box (profiles/round4_ring_waves_ab.txt)
    if constexpr (WM == 1 && WN == 1) launch_ring<2, 2, AK, BKM, OF, 4, F16, 1, 1>(g, batch, splits, ksplit, ws, st);
    else if constexpr (WM == 1 && WN == 2) launch_ring<2, 2, AK, BKM, OF, 4, F16, 1, 2>(g, batch, splits, ksplit, ws, st);
    else launch_ring<WM, WN, AK, BKM, OF, 4, F16>(g, batch, splits, ksplit, ws, st);
    return;
  }
  // LDS-DMA staging needs full K tiles (no zero fill); anything else takes the register-staged loop
  if ((g.K % BK) == 0 && !(g.flags & OFA_GEMM_NO_LDS_DMA)) launch_cfg2<WM, WN, AK, BKM, OF, true, F16>(g, batch, splits, ksplit, ws, st);
  else launch_cfg2<WM, WN, AK, BKM, OF, false, F16>(g, batch, splits, ksplit, ws, st);
}

template <bool AK, bool BKM, bool OF, bool F16 = false>
static void launch_shape(const GemmArgs& g, int batch, int wm, int wn, int splits, int ksplit, float* ws,
                         hipStream_t st) {
  if (wm == 2 && wn == 2) launch_cfg<2, 2, AK, BKM, OF, F16>(g, batch, splits, ksplit, ws, st);
  else if (wm == 1 && wn == 2) launch_cfg<1, 2, AK, BKM, OF, F16>(g, batch, splits, ksplit, ws, st);
  else launch_cfg<1, 1, AK, BKM, OF, F16>(g, batch, splits, ksplit, ws, st);
}

template <int TM, int TN, bool AK, bool BKM, bool OF, int WGM = 2, int WGN = 2, bool F16 = false>
static void launch_big(const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  const size_t lds = 4 * (size_t)256 * BK * sizeof(bf16_t);     // 2 operands x 2 stages x 32 KiB
  auto kern = gemm_big_kernel<TM, TN, AK, BKM, OF, WGM, WGN, F16>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(tiles_m * tiles_n, splits, batch), block(64 * WGM * WGN);
  hipLaunchKernelGGL(kern, grid, block, lds, st, g, tiles_m, tiles_n, ksplit, ws);
}

// Which main loop runs a big tile: 0 = the compiler-scheduled lockstep loop (gemm_big_kernel), 23 = the ping-pong loop (gemm_pp.hip).
// Measured on the step's products (profiles/round5_gemm_pp_ab.txt, 7 interleaved rounds, bit-identical results): the two are level on the
// k-major / k-major forward products at K = 768 (the loop is bound by what a K-tile moves, not by who issues it:
// profiles/round5_gemm_pp_ablate.txt, round5_gemm_pp_timeline.txt); with an m-major operand -- fragments are two transposing reads each:
// twice the LDS read instructions -- and a long contraction the ping-pong loop wins: grouped weight gradients of an encoder layer
// 209 -> 182 us, input gradients at K = 2304 / 3072 / 9216: 49.2 -> 47.0, 60.9 -> 59.4, 178 -> 169 us.
static int pp_variant(bool trans_a, bool trans_b, int K) {
#ifdef OFA_DEBUG_SWITCHES
  const char* e = getenv("OFA_GEMM_PP");       // (read per call: tools/gemm_pp_ab.py flips it inside one process)
  if (e) return atoi(e);
#endif
  if (trans_a && !trans_b) return 23;
  if (!trans_a && !trans_b && K >= 2048) return 23;
  return 0;
}

template <bool AK, bool BKM, bool OF, bool F16 = false>
static void launch_big_shape(const GemmArgs& g, int batch, int tm, int splits, int ksplit, float* ws, hipStream_t st) {
  if (const int v = pp_variant(!AK, BKM, g.K)) {
    if (gemm_pp_launch(v, g, batch, tm, splits, ksplit, ws, st, F16)) return;
  }
  if constexpr (AK) {
    if (tm == 3) { launch_big<3, 2, AK, BKM, OF, 2, 4, F16>(g, batch, splits, ksplit, ws, st); return; }   // 96 x 64 per wave
  }
  launch_big<4, 2, AK, BKM, OF, 2, 4, F16>(g, batch, splits, ksplit, ws, st);                              // 128 x 64 per wave
}

struct GemmPlan { int wm, wn, big_tm, splits, ksplit, K, mixed_a, mixed_b; };   // mixed_a / mixed_b: row tiles of 256 / 192 (gemm_big_mixed_kernel), 0 / 0: not used

// tile / split-K plan of one product (shared by the launcher and ofa_gemm_splits, which tells a caller that defers the
// split-K reduce how many partial slabs the launch will write)
static GemmPlan gemm_plan(GemmArgs g, int batch, bool has_ws, int64_t ws_bytes) {
  // zero-padded contraction (vocabulary-logit gradients, lda padded to a multiple of 64): run the LDS-DMA loop over the
  // padded K; A's tail columns are zeros, B's rows past K are clamped reads
  if ((g.flags & OFA_GEMM_A_KPAD_ZERO) && !g.transA && !g.transB && (g.K % BK) != 0 && ((g.K + BK - 1) / BK) * BK <= g.lda)
    g.K = ((g.K + BK - 1) / BK) * BK;
  // weight gradients over a row count that is not a multiple of 64 (both operands m-major): the LDS-DMA loop over the rounded-up K,
  // A's missing rows read as zeros, B's clamped (a_krows / b_krows = the real count, set by the launcher)
  if (g.transA && !g.transB && (g.K % BK) != 0 && !(g.flags & OFA_GEMM_NO_LDS_DMA)) g.K = ((g.K + BK - 1) / BK) * BK;
  // tile choice: the biggest tile that, together with split-K (when a workspace is given and K is long), still puts
  // >= ~1.5 workgroups on every CU; 128x128 tiles halve the LDS traffic per flop of 64-wide ones.
  const int64_t t22 = (int64_t)cdiv(g.M, 128) * cdiv(g.N, 128) * batch;
  const int64_t t12 = (int64_t)cdiv(g.M, 64) * cdiv(g.N, 128) * batch;
  const int64_t t11 = (int64_t)cdiv(g.M, 64) * cdiv(g.N, 64) * batch;
  int maxs = has_ws ? g.K / 256 : 1;                 // every split keeps >= 4 K-tiles
  maxs = maxs < 1 ? 1 : (maxs > 32 ? 32 : maxs);
  // short contractions are not split: at K = 768 a split saves a few K-steps but costs a second (reduce) launch --
  // 2048 x 2304 x 768: 28.9 us split in two + reduce vs 16.2 us unsplit; from K = 2304 up the split wins (21.5 vs 24.7 us).
  // Round 3 (tools/gemm_split_check.py, the ResNet trunk's 1 x 1 / 3 x 3 products): at K = 1024 the split loses too (18432 x 256 x 1024:
  // 31.7 us in two slices + reduce vs 20.2 us as 64 x 128 tiles; 6272 x 256 x 1024: 22.4 vs 14.0 us as 64 x 64 tiles), and so does any
  // split of a product whose 128 x 128 tiles already fill a round of the 256 CUs (18432 x 256 x 2304: 46.6 vs 37.6 us)
  // Planner overrides for experiments (tools/gemm_tile_sweep.py, gemm_split_check.py, gemm_timeline.py, the forced-tile test) exist in
  // the DEBUG library only (make -C ofasys_amd/csrc debug -> libofasys_amd_dbg.so, -DOFA_DEBUG_SWITCHES); the shipped library has
  // one code path and reads no environment variable.
#ifdef OFA_DEBUG_SWITCHES
  static const int split_min_k = getenv("OFA_GEMM_SPLIT_MIN_K") ? atoi(getenv("OFA_GEMM_SPLIT_MIN_K")) : 2048;
  const int force_tile = getenv("OFA_GEMM_TILE") ? atoi(getenv("OFA_GEMM_TILE")) : 0;   // 22 / 12 / 11 / 44 (= 84) / 34 (= 83); read per call
  static const int64_t nosplit_t11 = getenv("OFA_GEMM_NOSPLIT_T11") ? atoll(getenv("OFA_GEMM_NOSPLIT_T11")) : 256;
#else
  constexpr int split_min_k = 2048, force_tile = 0;
  constexpr int64_t nosplit_t11 = 256;
#endif
  if (g.K < split_min_k || t22 >= 256) maxs = 1;
  // Round 4: a forward / input-gradient product whose 64 x 64 tiles alone put a four-wave workgroup on every compute unit is not split
  // either (the decoder-side K = 2304 / 3072 input gradients: 1536 x 768 x 3072 ran as 72 tiles x 5 slices + a reduce launch, 26 + 6.5 us;
  // as 288 ring-staged 64 x 64 tiles it needs no second launch)
  if (!g.transA && g.K <= 4096 && t11 >= nosplit_t11) maxs = 1;
  const int64_t want = 384;
  int wm, wn;
  int64_t tiles;
  if (t22 * maxs >= want && g.M > 64 && g.N >= 128) { wm = 2; wn = 2; tiles = t22; }
  else if (t12 * maxs >= want && g.N >= 128) { wm = 1; wn = 2; tiles = t12; }
  else { wm = 1; wn = 1; tiles = t11; }
  if (force_tile == 22) { wm = 2; wn = 2; tiles = t22; }
  else if (force_tile == 12) { wm = 1; wn = 2; tiles = t12; }
  else if (force_tile == 11) { wm = 1; wn = 1; tiles = t11; }
  // Big tiles: ONE 512-thread workgroup per CU, eight waves of 128 x 64 (256 x 256 tile) or 96 x 64 (192 x 256) -- two waves per
  // SIMD, so one wave's barrier / LDS waits are covered by the other's MFMAs: 1.42 us per K-step of 2048 MFMA-clocks against
  // 0.87 us per 1024 for a pair of 128 x 128 workgroups (tools/gemm_timeline.py, profiles/round2_gemm_timeline.txt).  The
  // price is per tile: first-tile flight, epilogue, store acknowledgement and re-dispatch cost 7.6 us with nothing else
  // on the CU to hide them (4.2 us per pair of 128 x 128 workgroups), and the grid is quantised in rounds of 256 tiles.
  // The choice is made on that model (microseconds; profiles/round2_gemm_eight_wave_sweep.txt is what it was fitted to):
  //     time = rounds x (K-steps x step + per-tile)      128 x 128: 512 slots, step 0.90 (1.0 NT), per-tile 4.2
  //                                                      256 x 256: 256 slots, step 1.45, per-tile 7.6
  //                                                      192 x 256: 256 slots, step 1.13, per-tile 6.5   (k-major A only)
  // e.g. 13312 x 768 x 3072: 95 / 77 / 61 -> 192 x 256 (measured 81 / 77-83 / 65); 13312 x 2304 x 768: 62 / 50 / 60 -> 256 x 256
  // (62 / 54 / 61); 13312 x 3072 x 768 (NN): 75 / 75 / 80 -> stays on 128 x 128 (76 / 76 / 84).  m-major A (weight gradients)
  // keeps the rule of round 1: one round of <= 256 tiles or K >= 4096, on the 256 x 256 tile.
  int big_tm = 0;                                           // 3: 192 x 256, 4: 256 x 256
  const bool big_ok = (g.K % BK) == 0 && !(g.flags & OFA_GEMM_NO_LDS_DMA) && g.N >= 256 && g.M >= 192;
  if (big_ok && force_tile != 22 && force_tile != 12 && force_tile != 11) {
    const int64_t t4 = (int64_t)cdiv(g.M, 256) * cdiv(g.N, 256) * batch, t3 = (int64_t)cdiv(g.M, 192) * cdiv(g.N, 256) * batch;
    if (force_tile == 44 || force_tile == 84) big_tm = 4;
    else if (force_tile == 34 || force_tile == 83) big_tm = g.transA ? 4 : 3;
    else if (!g.transA) {
      const int nk = g.K / BK;
      if (batch == 1 && nk >= 4 && t22 >= 512 && wm == 2 && wn == 2) {
        auto est = [&](int64_t t, int slots, double step, double per_tile) { return (double)cdiv(t, slots) * (nk * step + per_tile); };
        const double e2 = est(t22, 512, g.transB ? 1.0 : 0.90, 4.2);
        const double e4 = t4 >= 128 ? est(t4, 256, 1.45, 7.6) : 1e30, e3 = t3 >= 128 ? est(t3, 256, 1.13, 6.5) : 1e30;
        if (e4 <= e3 && e4 < 0.95 * e2) big_tm = 4;
        else if (e3 < e4 && e3 < 0.95 * e2) big_tm = 3;
      }
    } else {
      const double f4 = (double)t4 / (double)(cdiv(t4, 256) * 256) * ((double)g.M / (cdiv(g.M, 256) * 256)) * ((double)g.N / (cdiv(g.N, 256) * 256));
      // (round 6: also several well-filled rounds over a medium contraction -- the vocabulary projection's weight gradient, 51272 x 768
      //  over 1536 decoder rows, 603 tiles = 2.36 rounds: 156 us on 128 x 128 tiles, whose transposing fragment reads the time model
      //  above flatters, 131 us here)
      if (f4 >= 0.7 && (g.K >= 4096 || t4 <= 256 || g.K >= 1024)) big_tm = 4;
    }
    if (big_tm) {
      wm = wn = 0;
      tiles = big_tm == 4 ? t4 : t3;
    }
  }
  int splits = 1;
  // Round 6: a k-major-A product with a very long contraction and a handful of output tiles (the vocabulary projection's input gradient,
  // 1536 x 768 x 51328: 72 tiles of 128 x 128 in 6 K-slices ran at 15 % of peak, every tile re-reading its operand panels at 64 flop / B)
  // takes the 256 x 256 tile cut into as many K-slices as fill ONE round of the 256 CUs: twice the flops per operand byte, ~57 K tiles per
  // workgroup, fp32 slabs summed by the reduce launch (tools/gemm_bench.py, profiles/round6_gemm_microbench.txt).
#ifdef OFA_DEBUG_SWITCHES
  const bool big_split_on = !(getenv("OFA_GEMM_BIGSPLIT") && atoi(getenv("OFA_GEMM_BIGSPLIT")) == 0);
#else
  constexpr bool big_split_on = true;
#endif
  if (big_split_on && !big_tm && big_ok && !force_tile && !g.transA && batch == 1 && has_ws && g.K >= 8192) {
    const int64_t t4 = (int64_t)cdiv(g.M, 256) * cdiv(g.N, 256);
    int s = t4 <= 64 ? (int)(256 / t4) : 1;
    s = s > maxs ? maxs : s;
    while (s > 1 && (int64_t)s * g.M * ((g.N + 3) & ~3) * 4 > ws_bytes) --s;
    if (s >= 4) {
      big_tm = 4;
      wm = wn = 0;
      tiles = t4;
      splits = s;
    }
  }
  if (!big_tm && tiles < want && maxs > 1) {
    splits = (int)((want + tiles - 1) / tiles);
    if (splits > maxs) splits = maxs;
    while (splits > 1 && (int64_t)splits * batch * g.M * ((g.N + 3) & ~3) * 4 > ws_bytes) --splits;
  }
  int ksplit = g.K;
  if (splits > 1) {
    ksplit = cdiv(cdiv(g.K, splits), BK) * BK;
    splits = cdiv(g.K, ksplit);
  }
  // Round 6: two tile heights in one launch (gemm_big_mixed_kernel) where the plan above leaves a partly filled round: k-major A, one
  // K-slice, no batch, plain epilogue (the column statistics and the row bias index rows of the whole product).  The candidates
  // (a tall row tiles, the rest short) are simulated once per shape; the choice has to beat the plan's own estimate by 5 % (tools/gemm_mixed_bench.py, MI355X, replayed graphs:
  // 13312 x 3072 x 768 NT 76.5 -> 68.5 us, its NN twin 78.3 -> 68.6; 13312 x 9216 x 768 -- 4 % by the model -- measured 201 -> 192 alone but
  // nothing inside the step and 200 -> 225 in an eager back-to-back loop, where consecutive launches overlap their tails: not taken).
  int mixed_a = 0, mixed_b = 0;
#ifdef OFA_DEBUG_SWITCHES
  const int mixed_mode = getenv("OFA_GEMM_MIXED") ? atoi(getenv("OFA_GEMM_MIXED")) : -1;     // 0: never, 1: whenever eligible, -1: the model decides
#else
  constexpr int mixed_mode = -1;
#endif
  if (mixed_mode != 0 && big_ok && !force_tile && !g.transA && batch == 1 && splits == 1 && !g.colstat && !(g.flags & OFA_GEMM_BIAS_ROW) &&
      g.M >= 1024 && g.N >= 512 && g.K >= 4 * BK && pp_variant(false, g.transB != 0, g.K) == 0) {
    struct Key { int M, N, K, tb; };
    struct Val { int a, b; double t; };
    static std::mutex mu;
    static std::vector<std::pair<Key, Val>> cache;
    const int nk = g.K / BK, tiles_n = cdiv(g.N, 256);
    Val best{0, 0, 1e30};
    bool hit = false;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (auto& e : cache)
        if (e.first.M == g.M && e.first.N == g.N && e.first.K == g.K && e.first.tb == g.transB) { best = e.second; hit = true; break; }
    }
    if (!hit) {
      const int amax = g.M / 256;
      for (int a = 1; a <= amax; ++a) {
        const int rest = g.M - a * 256;
        if (rest <= 0) break;
        const int b = cdiv(rest, 192);
        const double t = mixed_time_us(a * tiles_n, b * tiles_n, nk);
        if (t < best.t) best = Val{a, b, t};
      }
      std::lock_guard<std::mutex> lk(mu);
      if (cache.size() < 256) cache.push_back({Key{g.M, g.N, g.K, g.transB}, best});
    }
    // the plan's own estimate, by the same model
    double cur;
    if (big_tm == 4) cur = (double)cdiv(tiles, 256) * (nk * 1.45 + 7.6);
    else if (big_tm == 3) cur = (double)cdiv(tiles, 256) * (nk * 1.13 + 6.5);
    else cur = (double)cdiv(tiles, (wm == 2 && wn == 2) ? 512 : 1024) * (nk * (g.transB ? 1.0 : 0.90) + 4.2);
    const bool modelled = (wm == 2 && wn == 2) || big_tm;               // (the plans the time model above covers)
    if (best.a > 0 && best.b > 0 && (mixed_mode == 1 || (modelled && best.t < 0.95 * cur))) {
      mixed_a = best.a;
      mixed_b = best.b;
    }
  }
  return GemmPlan{wm, wn, big_tm, splits, ksplit, g.K, mixed_a, mixed_b};
}

int gemm_mfma_splits(const GemmArgs& g, int batch, int64_t ws_bytes) { return gemm_plan(g, batch, ws_bytes > 0, ws_bytes).splits; }

// rows per column-statistics partial row of the kernel the plan selects (0: that kernel cannot produce them): the accumulator
// rows of one wave, 32 * TM (epilogue_lds)
static int plan_colstat_rows(const GemmArgs& g, const GemmPlan& pl, int batch) {
  if (pl.splits > 1 || batch != 1) return 0;
  if (pl.big_tm) return pl.big_tm * 32;
  GemmArgs gp = g;
  gp.K = pl.K;
  const int64_t blocks = (int64_t)cdiv(g.M, 64 * pl.wm) * cdiv(g.N, 64 * pl.wn) * pl.splits * batch;
  return (use_ring(gp, pl.wm, pl.wn, blocks, pl.ksplit) && pl.wm == 1) ? 32 : 64;      // (the four-wave ring forms: 32-row wave blocks)
}

int gemm_mfma_colstat_groups(const GemmArgs& g_in, void* ws, int64_t ws_bytes) {
  GemmArgs g = g_in;
  if ((g.flags & (OFA_GEMM_OUT_F32 | OFA_GEMM_ACCUM)) || (g.N & 7)) return 0;
  const GemmPlan pl = gemm_plan(g, 1, ws != nullptr, ws_bytes);
  const int rows = plan_colstat_rows(g, pl, 1);
  return rows ? cdiv(g.M, rows) : 0;
}

int gemm_mfma_launch(const GemmArgs& g_in, int batch, void* ws, int64_t ws_bytes, hipStream_t st, bool f16) {
  GemmArgs g = g_in;
  g.a_krows = g.b_krows = g.K;
  const GemmPlan pl = gemm_plan(g, batch, ws != nullptr, ws_bytes);
  if (g.colstat) g.colstat_rows = plan_colstat_rows(g, pl, batch);
  g.K = pl.K;
  const int wm = pl.wm, wn = pl.wn, big_tm = pl.big_tm, splits = pl.splits, ksplit = pl.ksplit;
  const bool ak = !g.transA, bk = g.transB != 0, of = (g.flags & OFA_GEMM_OUT_F32) != 0;
  if (pl.mixed_a > 0) {                                     // (k-major A, one K-slice, no batch: gemm_plan)
    if (bk) { if (of) { if (f16) launch_big_mixed<true, true, true>(g, pl.mixed_a, pl.mixed_b, st); else launch_big_mixed<true, true, false>(g, pl.mixed_a, pl.mixed_b, st); }
              else { if (f16) launch_big_mixed<true, false, true>(g, pl.mixed_a, pl.mixed_b, st); else launch_big_mixed<true, false, false>(g, pl.mixed_a, pl.mixed_b, st); } }
    else { if (of) { if (f16) launch_big_mixed<false, true, true>(g, pl.mixed_a, pl.mixed_b, st); else launch_big_mixed<false, true, false>(g, pl.mixed_a, pl.mixed_b, st); }
           else { if (f16) launch_big_mixed<false, false, true>(g, pl.mixed_a, pl.mixed_b, st); else launch_big_mixed<false, false, false>(g, pl.mixed_a, pl.mixed_b, st); } }
    return check_launch("gemm_big_mixed");
  }
#define GEMM_DISPATCH(AK, BKM, OF)                                                                      \
  do {                                                                                                  \
    if (f16) {                                                                                          \
      if (big_tm) launch_big_shape<AK, BKM, OF, true>(g, batch, big_tm, splits, ksplit, (float*)ws, st);  \
      else launch_shape<AK, BKM, OF, true>(g, batch, wm, wn, splits, ksplit, (float*)ws, st);           \
    } else if (big_tm) launch_big_shape<AK, BKM, OF>(g, batch, big_tm, splits, ksplit, (float*)ws, st); \
    else launch_shape<AK, BKM, OF>(g, batch, wm, wn, splits, ksplit, (float*)ws, st);                   \
  } while (0)
  if (ak && bk) { if (of) GEMM_DISPATCH(true, true, true); else GEMM_DISPATCH(true, true, false); }
  else if (ak && !bk) { if (of) GEMM_DISPATCH(true, false, true); else GEMM_DISPATCH(true, false, false); }
  else if (!ak && !bk) { if (of) GEMM_DISPATCH(false, false, true); else GEMM_DISPATCH(false, false, false); }
  else { if (of) GEMM_DISPATCH(false, true, true); else GEMM_DISPATCH(false, true, false); }
#undef GEMM_DISPATCH
  int rc = check_launch("gemm_mfma");
  if (rc) return rc;
  if (splits > 1 && !(g.flags & OFA_GEMM_DEFER_REDUCE)) {
    const int64_t quads = (int64_t)g.M * ((g.N + 3) / 4);
    dim3 grid((unsigned)((quads + 255) / 256 > 2048 ? 2048 : (quads + 255) / 256), batch), block(256);
    if (of && f16) hipLaunchKernelGGL((splitk_reduce_kernel<true, true>), grid, block, 0, st, g, (const float*)ws, splits);
    else if (of) hipLaunchKernelGGL(splitk_reduce_kernel<true>, grid, block, 0, st, g, (const float*)ws, splits);
    else if (f16) hipLaunchKernelGGL((splitk_reduce_kernel<false, true>), grid, block, 0, st, g, (const float*)ws, splits);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, grid, block, 0, st, g, (const float*)ws, splits);
    rc = check_launch("gemm_splitk_reduce");
  }
  return rc;
}

}  // namespace ofa

using namespace ofa;

extern "C" int ofa_gemm_splits(int M, int N, int K, int transA, int transB, int batch, int flags, int dtype, int64_t ws_bytes) {
  if ((dtype != OFA_BF16 && dtype != OFA_F16) || (flags & OFA_GEMM_FORCE_SIMPLE) || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 1;
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.transA = transA; g.transB = transB; g.flags = flags;
  g.lda = transA ? M : K; g.ldb = transB ? K : N; g.ldc = N;
  return gemm_mfma_splits(g, batch, ws_bytes);
}

// ---- grouped weight-gradient products
static bool group_item_ok(const ofa_gemm_group_item& it) {
  if (it.m <= 0 || it.n <= 0 || it.k <= 0 || !it.a || !it.b) return false;
  if ((it.m & 7) || (it.n & 7)) return false;                         // 16-byte vectors along m and n (k: any row count, see ofa_zero16)
  if (it.lda < it.m || it.ldb < it.n || (it.lda & 7) || (it.ldb & 7)) return false;
  if (((uintptr_t)it.a & 15) || ((uintptr_t)it.b & 15)) return false;
  return true;
}

// K-slice length shared by the group (every product is cut into ceil(k / length) slices; a slice keeps >= 4 K tiles): the candidate with the
// shortest estimated time -- the finish time of the busiest XCD (an eighth of the flattened workgroup list, contiguous: group_enter), its
// workgroups handed to its 32 CUs in order as they free up, plus what the slabs cost afterwards.  Constants from the cfg-2 traces
// (profiles/round5_*): 1.5 us per K tile of a 256 x 256 workgroup, 6 us of prologue / epilogue / re-dispatch per workgroup (163 us for 104
// K tiles, 317 us for 208), the fold launch moving (4 B per slab + old and new 16-bit gradient) per element at 4.1 TB/s (20.5 us for an encoder
// layer's two-slab products, 26.8 us for a decoder layer's).  (Rounds 3-4: the shortest length with at most 256 workgroups in the group --
// it never weighed the slabs against an idle sixth of the chip, and two layers' products in one launch need none.)
static double group_time_us(const ofa_gemm_group_item* items, int n, int len) {
  int steps[GROUP_MAX], count[GROUP_MAX], total = 0;
  double slab_bytes = 0.0;
  for (int p = 0; p < n; ++p) {
    int sp = cdiv(items[p].k, len);
    sp = sp > 32 ? 32 : sp;
    const int ksplit = cdiv(cdiv(items[p].k, BK), sp) * BK;
    sp = cdiv(items[p].k, ksplit);
    steps[p] = ksplit / BK;
    count[p] = cdiv(items[p].m, 256) * cdiv(items[p].n, 256) * sp;
    total += count[p];
    if (sp > 1) slab_bytes += ((double)sp * 4.0 + 4.0) * items[p].m * items[p].n;
  }
  double worst = 0.0;
  for (int x = 0; x < 8; ++x) {                                             // xcd_remap: XCD x runs ids [lo, hi) of the flattened list
    const int q = total >> 3, r = total & 7;
    const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, hi = lo + q + (x < r ? 1 : 0);
    double cu[32];
    for (int c = 0; c < 32; ++c) cu[c] = 0.0;
    int id = 0;
    for (int p = 0; p < n; ++p)
      for (int w = 0; w < count[p]; ++w, ++id) {
        if (id < lo || id >= hi) continue;
        int best = 0;
        for (int c = 1; c < 32; ++c) best = cu[c] < cu[best] ? c : best;
        cu[best] += 1.5 * steps[p] + 6.0;
      }
    for (int c = 0; c < 32; ++c) worst = cu[c] > worst ? cu[c] : worst;
  }
  return worst + (slab_bytes > 0.0 ? 2.0 + slab_bytes / 4.1e6 : 0.0);
}

// (memoised on the group's shapes: the simulation below costs ~10 ms of host time per call, 7 calls per eager cfg-2 step -- a replayed
//  graph never sees it, an eager step was host-bound on it; found by tools/gemm_bench.py timing the grouped launch at 5.3 ms in round 6)
static int group_plan_len(const ofa_gemm_group_item* items, int n);
static void group_plan(ofa_gemm_group_item* items, int n) {
  struct Key { int n; int mnk[GROUP_MAX][3]; };
  static std::mutex mu;
  static std::vector<std::pair<Key, int>> cache;
  Key key{};
  key.n = n;
  for (int p = 0; p < n; ++p) { key.mnk[p][0] = items[p].m; key.mnk[p][1] = items[p].n; key.mnk[p][2] = items[p].k; }
  int len = -1;
  {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : cache)
      if (!memcmp(&e.first, &key, sizeof(Key))) { len = e.second; break; }
  }
  if (len < 0) {
    len = group_plan_len(items, n);
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() < 512) cache.push_back({key, len});
  }
  for (int p = 0; p < n; ++p) {
    int sp = cdiv(items[p].k, len);
    sp = sp > 32 ? 32 : sp;
    const int ksplit = cdiv(cdiv(items[p].k, BK), sp) * BK;
    items[p].splits = cdiv(items[p].k, ksplit);
  }
}

static int group_plan_len(const ofa_gemm_group_item* items, int n) {
  int kmax = 0;
  for (int p = 0; p < n; ++p) kmax = items[p].k > kmax ? items[p].k : kmax;
  const int tiles = cdiv(kmax, BK);
  int len = tiles * BK;
  double best = group_time_us(items, n, len);
  for (int sp = 2; sp <= 32 && cdiv(tiles, sp) >= 4; ++sp) {               // the lengths at which the longest product gains a slice
    const int l = cdiv(tiles, sp) * BK;
    const double t = group_time_us(items, n, l);
    if (t < best * 0.97) { best = t; len = l; }                             // (a tie goes to fewer slices: less slab traffic)
  }
  return len;
}

extern "C" int ofa_gemm_group_plan(ofa_gemm_group_item* items, int n, int dtype) {
  OFA_REQUIRE(dtype == OFA_BF16 || dtype == OFA_F16, OFA_ERR_INVALID, "gemm_group: 16-bit operands only (dtype %d)", dtype);
  OFA_REQUIRE(items && n >= 1 && n <= GROUP_MAX, OFA_ERR_INVALID, "gemm_group: 1..%d products per launch (got %d)", GROUP_MAX, n);
  for (int p = 0; p < n; ++p)
    OFA_REQUIRE(group_item_ok(items[p]), OFA_ERR_INVALID,
                "gemm_group: product %d (m=%d n=%d k=%d lda=%lld ldb=%lld) needs m, n, lda, ldb %% 8 == 0 and 16-byte aligned operands",
                p, items[p].m, items[p].n, items[p].k, (long long)items[p].lda, (long long)items[p].ldb);
  group_plan(items, n);
  return 0;
}

extern "C" int ofa_gemm_group_tn(const ofa_gemm_group_item* items, int n, int dtype, void* stream) {
  OFA_REQUIRE(dtype == OFA_BF16 || dtype == OFA_F16, OFA_ERR_INVALID, "gemm_group: 16-bit operands only (dtype %d)", dtype);
  OFA_REQUIRE(items && n >= 1 && n <= GROUP_MAX, OFA_ERR_INVALID, "gemm_group: 1..%d products per launch (got %d)", GROUP_MAX, n);
  GroupArgs ga;
  int first = 0;
  for (int p = 0; p < n; ++p) {
    const ofa_gemm_group_item& it = items[p];
    const bool direct = it.out != nullptr && it.splits == 1;
    OFA_REQUIRE(group_item_ok(it) && (direct || (it.slabs && !((uintptr_t)it.slabs & 15))), OFA_ERR_INVALID, "gemm_group: product %d is not eligible", p);
    OFA_REQUIRE(!direct || (!((uintptr_t)it.out & 15) && it.ldo >= it.n && !(it.ldo & 7)), OFA_ERR_INVALID,
                "gemm_group: product %d: out must be 16-byte aligned with a row stride that is a multiple of 8 elements", p);
    OFA_REQUIRE(it.splits >= 1 && it.splits <= 32, OFA_ERR_INVALID, "gemm_group: product %d: splits %d (run ofa_gemm_group_plan)", p, it.splits);
    GroupItem& d = ga.it[p];
    d.A = it.a; d.B = it.b; d.ws = it.slabs; d.lda = it.lda; d.ldb = it.ldb;
    d.out = direct ? it.out : nullptr; d.ldo = it.ldo; d.alpha = it.out_alpha;
    d.M = it.m; d.N = it.n; d.K = cdiv(it.k, BK) * BK; d.krows = it.k;
    d.ksplit = cdiv(cdiv(it.k, BK), it.splits) * BK;
    d.splits = it.splits;
    OFA_REQUIRE(cdiv(it.k, d.ksplit) == it.splits, OFA_ERR_INVALID, "gemm_group: product %d: %d slices leave an empty one", p, it.splits);
    d.tiles_m = cdiv(it.m, 256); d.tiles_n = cdiv(it.n, 256);
    d.first = first;
    first += d.tiles_m * d.tiles_n * d.splits;
  }
  for (int p = n; p < GROUP_MAX; ++p) ga.it[p] = ga.it[0];
  ga.n = n; ga.total = first;
  const size_t lds = 4 * (size_t)256 * BK * sizeof(bf16_t);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_group_tn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_group_tn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipStream_t st = (hipStream_t)stream;
#ifdef OFA_DEBUG_SWITCHES
  static const bool w4 = getenv("OFA_GROUP_W4") && atoi(getenv("OFA_GROUP_W4"));     // experiment: four waves of 128 x 128
  if (w4 && dtype != OFA_F16) {
    static bool attr4 = false;
    if (!attr4) {
      (void)hipFuncSetAttribute((const void*)gemm_group_tn_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr4 = true;
    }
    hipLaunchKernelGGL((gemm_group_tn_kernel<false, true>), dim3(ga.total), dim3(256), lds, st, ga);
    return check_launch("gemm_group_tn_w4");
  }
#endif
  if (const int v = pp_variant(true, false, 0)) {
    if (gemm_group_pp_launch(v, ga, dtype == OFA_F16, st)) return check_launch("gemm_group_tn_pp");
  }
  if (dtype == OFA_F16) hipLaunchKernelGGL(gemm_group_tn_kernel<true>, dim3(ga.total), dim3(512), lds, st, ga);
  else hipLaunchKernelGGL(gemm_group_tn_kernel<false>, dim3(ga.total), dim3(512), lds, st, ga);
  return check_launch("gemm_group_tn");
}

extern "C" int ofa_gemm(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int transA,
                        int transB, int64_t lda, int64_t ldb, int64_t ldc, int batch, int64_t strideA, int64_t strideB,
                        int64_t strideC, int batch_inner, int64_t strideA2, int64_t strideB2, int64_t strideC2,
                        float alpha, int flags, int dtype, void* ws, int64_t ws_bytes, void* stream) {
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "gemm: bad dtype %d", dtype);
  OFA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, OFA_ERR_INVALID, "gemm: negative size");
  if (M == 0 || N == 0 || batch == 0) return 0;
  OFA_REQUIRE(A && B && C, OFA_ERR_INVALID, "gemm: null pointer");
  OFA_REQUIRE(!(flags & (OFA_GEMM_BIAS_COL | OFA_GEMM_BIAS_ROW)) || bias, OFA_ERR_INVALID, "gemm: bias flag without bias");
  OFA_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, OFA_ERR_INVALID,
              "gemm: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)", (long long)lda, (long long)ldb,
              (long long)ldc);
  OFA_REQUIRE(!(dtype == OFA_F32 && (flags & OFA_GEMM_OUT_F32)), OFA_ERR_INVALID, "gemm: OUT_F32 is for 16-bit inputs");
  if (batch_inner <= 0 || batch_inner >= batch) { batch_inner = batch; strideA2 = strideB2 = strideC2 = 0; }
  GemmArgs g{A, B, C, bias, M, N, K, transA, transB, lda, ldb, ldc, strideA, strideB, strideC, alpha, flags,
             batch_inner, strideA2, strideB2, strideC2};
  hipStream_t st = (hipStream_t)stream;
  if (dtype != OFA_F32 && !(flags & OFA_GEMM_FORCE_SIMPLE) && K > 0 && gemm_mfma_supported(g))
    return gemm_mfma_launch(g, batch, ws, ws_bytes, st, dtype == OFA_F16);
  return gemm_simple_launch(g, batch, dtype, st);
}

extern "C" int ofa_gemm_colstat(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, int transA, int transB,
                                int64_t lda, int64_t ldb, int64_t ldc, float alpha, int flags, int dtype, void* ws, int64_t ws_bytes,
                                double* partial, int max_groups, int* groups, void* stream) {
  OFA_REQUIRE(groups, OFA_ERR_INVALID, "gemm_colstat: groups must not be NULL");
  *groups = 0;
  OFA_REQUIRE(OFA_DT_OK(dtype), OFA_ERR_INVALID, "gemm_colstat: bad dtype %d", dtype);
  OFA_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, OFA_ERR_INVALID, "gemm_colstat: bad argument");
  OFA_REQUIRE(!(flags & (OFA_GEMM_BIAS_COL | OFA_GEMM_BIAS_ROW)) || bias, OFA_ERR_INVALID, "gemm_colstat: bias flag without bias");
  OFA_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, OFA_ERR_INVALID, "gemm_colstat: leading dimension too small");
  GemmArgs g{A, B, C, bias, M, N, K, transA, transB, lda, ldb, ldc, 0, 0, 0, alpha, flags, 1, 0, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  if (dtype != OFA_F32 && !(flags & OFA_GEMM_FORCE_SIMPLE) && gemm_mfma_supported(g)) {
    const int ng = partial ? gemm_mfma_colstat_groups(g, ws, ws_bytes) : 0;
    if (ng > 0 && ng <= max_groups) {
      g.colstat = partial;
      *groups = ng;
    }
    return gemm_mfma_launch(g, 1, ws, ws_bytes, st, dtype == OFA_F16);
  }
  return gemm_simple_launch(g, 1, dtype, st);
}
