// Ping-pong main loop for the 256 x 256 x 64 (192 x 256 x 64) bf16 / fp16 MFMA GEMM tile -- gfx950, one 512-thread workgroup per CU.
//
// Same contract, operand layouts (NT / NN / TN without HBM transposes), LDS image (LDS-DMA with the source-side chunk swizzle),
// fragment reads and epilogue as gemm_big_kernel (gemm_mfma.hip); what differs is WHEN things are issued.  The compiler-scheduled
// loop there runs its eight waves in lockstep: every wave weaves its own fragment reads and LDS-DMA pieces between its own MFMAs and
// all of them meet at one stage barrier per K-tile (vmcnt(0) in front of it) -- 78 % of the matrix pipe inside the loop, 1309 TFLOP/s
// at 8192^3 against 1583 for the vendor's hand-placed kernel (profiles/round4_gemm_microbench.txt).  Here the two waves of a SIMD
// take TURNS (cdna_hip_programming.md section 5, "8-phase" schedule; MI355X_MICROARCH.md "Two waves per SIMD"):
//
//   * the workgroup is two GROUPS of four waves (one wave of each group per SIMD); group g owns rows [g*HM, (g+1)*HM) of the tile
//     (HM = 128 or 96), wave w of a group the 64 columns [64 w, 64 w + 64): 128 (96) x 64 accumulators per wave as before;
//   * a K-tile is P = 4 / NKS PHASES of NKS k-slices (NKS = 2: 16 MFMAs per phase).  A phase is a LOAD segment (the phase's
//     fragment reads, then this wave's share of LDS-DMA pieces) and an MFMA segment (wait for the fragments, s_setprio 1, the
//     phase's MFMAs back to back, s_setprio 0), each closed by a bare s_barrier; group 1 starts one barrier late, so in every
//     barrier interval one wave of a SIMD multiplies while its partner loads: matrix beside memory, never matrix beside matrix;
//   * LDS: 160 KiB.  A: 2 slots x (2 groups x 16 KiB) -- a group's half is DMA'd, read and retired by that group alone.
//     B: 3 slots x 32 KiB, shared.  Prefetch distance: A one K-tile, B two; nothing in the loop waits vmcnt(0):
//       issue order per wave   ... a(t+1) [L(t,first)]   b(t+2) [L(t,last)]   a(t+2)   b(t+3) ...
//       end of L(t,last):  vmcnt(NVA + NVB)  -> b(t+1) has landed  (group 0 reads it one interval before group 1 would wait for it)
//       end of M(t,last):  vmcnt(NVB)        -> a(t+1) has landed
//     RAW: a buffer is read one barrier after the wait that retires its DMA (the reader of the other group: two).  WAR: a(t+1)
//     overwrites A(t-1), whose reads every wave of the group completed (lgkmcnt(0)) before the barrier closing M(t-1,last);
//     b(t+2) overwrites B(t-1): group 1's reads of it completed at the start of ITS M(t-1,last), two barriers before group 0
//     issues b(t+2) in L(t,last).
// Roofline: MFMA-bound; algorithmic flops = 2*M*N*K.  Reference call sites as gemm_mfma.hip (multihead_attention.py:199-217,346;
// transformer_layer.py:194,202).
#include "gemm_core.h"

namespace ofa {

constexpr uint32_t PP_A0 = 0u;             // A slots at 0 / 32 KiB (toggle: xor 0x8000); group half at + grp * 16 KiB
constexpr uint32_t PP_B0 = 65536u;         // B slots at 64 / 96 / 128 KiB
constexpr uint32_t PP_SLOT = 32768u;
constexpr int PP_LDS = 163840;

#define PP_SB __builtin_amdgcn_sched_barrier(0)

// makes a fragment opaque at this point of the stream: nothing that consumes it is scheduled above (section 5.7, item 3)
__device__ __forceinline__ void pp_pin(u64x2& d) { asm volatile("" : "+v"(d)); }

// Segment timing probe (measurement build only: make -C ofasys_amd/csrc timeline -> libofasys_amd_tl.so, tools/gemm_pp_timeline.py).  Waves 0
// and 4 (one of each group) sum, over the phases of their K loop, the shader clocks between: load-segment start -> its last issue ->
// barrier passed -> fragments landed -> MFMA segment issued (+ vmcnt) -> next load-segment start; s_memtime results come back on
// lgkmcnt and are only read behind the loop's own lgkmcnt(0), so the probe adds no wait of its own.
#ifdef OFA_PP_TIMELINE
#define PP_T(x) x = __builtin_amdgcn_s_memtime()
#else
#define PP_T(x) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void pp_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// (t, ks, bz): output tile, K-slice and batch index of this workgroup; nsplit > 1 or to_ws: the raw fp32 tile goes to slab ks of ws
// ABL (measurement builds of tools/gemm_pp_ab.py only, WRONG results): 1 no fragment reads, 2 no LDS-DMA, 4 no MFMAs inside the loop
// NDL: LDS-DMA pieces of a phase issued in its LOAD segment; the others go between the MFMAs of its MFMA segment
template <int TM, bool A_KMAJ, bool B_KMAJ, bool OUT_F32, bool F16, int NKS, bool STAGGER, bool PRIO, int ABL = 0, int NDL = 4>
__device__ __forceinline__ void gemm_pp_body(const GemmArgs& g, int tiles_m, int tiles_n, int ksplit, float* __restrict__ ws, int t,
                                             int ks, int bz, int nsplit, bool to_ws) {
  constexpr int TN = 2, HM = 32 * TM, BM = 2 * HM, BN = 256;
  static_assert(A_KMAJ || HM == 128, "an m-major A half is 128 wide (swizzle)");
  static_assert(NKS == 1 || NKS == 2 || NKS == 4, "k-slices per phase");
  constexpr int P = 4 / NKS;
  constexpr int NVA = HM * 8 / 256, NVB = BN * 8 / 512;     // LDS-DMA pieces per wave and K-tile: own A half (4 or 3), shared B (4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave_u >> 2, wn = wave_u & 3;
  const int tid_g = tid & 255;
  constexpr int GM = 8;
  const int gsz = GM * tiles_n;
  const int gid = t / gsz, first_m = gid * GM;
  const int rows_in_group = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int tm = first_m + (t % gsz) % rows_in_group, tn = (t % gsz) / rows_in_group;
  const int m0 = tm * BM, n0 = tn * BN;
  const int mg = m0 + grp * HM;                                 // first row of this group's half
  const bf16_t* A = (const bf16_t*)g.A + batch_off(bz, g.batch_inner, g.strideA, g.strideA2);
  const bf16_t* B = (const bf16_t*)g.B + batch_off(bz, g.batch_inner, g.strideB, g.strideB2);
  const int kbeg = ks * ksplit;
  const int kend = (kbeg + ksplit < g.K) ? kbeg + ksplit : g.K;
  const int nk = (kend - kbeg) / BK;                            // launcher guarantees whole K tiles

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bf16_t* pa[NVA];
  const bf16_t* pb[NVB];
  glds_ptrs<HM, A_KMAJ, 256, NVA, true>(pa, A, g.lda, mg, g.M, kbeg, tid_g, g.a_krows);
  glds_ptrs<BN, B_KMAJ, 512, NVB>(pb, B, g.ldb, n0, g.N, kbeg, tid, g.b_krows);
  const int64_t stepA = A_KMAJ ? BK : (int64_t)BK * g.lda, stepB = B_KMAJ ? BK : (int64_t)BK * g.ldb;
  int ka_next = kbeg, kb_next = kbeg;
  // this wave's pieces of K-tile (ka_next / kb_next) into the A half at byte offset `slot` / the B slot at `slot`
  // piece i of this wave's share of K-tile (ka_next / kb_next) into the A half at byte offset `slot` / the B slot at `slot`
  auto dma_a_piece = [&](auto ic, uint32_t slot) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i == 0) {
      if (!A_KMAJ && ka_next + BK > g.a_krows)                  // ragged contraction tail: A's missing k rows read as zeros
        glds_ptrs<HM, A_KMAJ, 256, NVA, true>(pa, A, g.lda, mg, g.M, ka_next, tid_g, g.a_krows);
    }
    unsigned char* d = smem_raw + slot + (uint32_t)grp * 16384u + (uint32_t)(wave_u & 3) * 1024u;
    __builtin_amdgcn_global_load_lds((gvoid_t*)pa[i], (lvoid_t*)(d + i * 4096), 16, 0, 0);
    pa[i] += stepA;
    if constexpr (i == NVA - 1) ka_next += BK;
  };
  auto dma_b_piece = [&](auto ic, uint32_t slot) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i == 0) {
      if (!B_KMAJ && kb_next + BK > g.b_krows)                  // zero-padded contraction tail: clamp B's k rows
        glds_ptrs<BN, B_KMAJ, 512, NVB>(pb, B, g.ldb, n0, g.N, kb_next, tid, g.b_krows);
    }
    unsigned char* d = smem_raw + slot + (uint32_t)wave_u * 1024u;
    __builtin_amdgcn_global_load_lds((gvoid_t*)pb[i], (lvoid_t*)(d + i * 8192), 16, 0, 0);
    pb[i] += stepB;
    if constexpr (i == NVB - 1) kb_next += BK;
  };
  auto dma_a = [&](uint32_t slot) { static_for<0, NVA>([&](auto ic) { dma_a_piece(ic, slot); }); };
  auto dma_b = [&](uint32_t slot) { static_for<0, NVB>([&](auto ic) { dma_b_piece(ic, slot); }); };
  constexpr int LA = NDL < NVA ? NDL : NVA, LB = NDL < NVB ? NDL : NVB;       // pieces issued in the load segment

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  BigAddr<HM, A_KMAJ> fax;
  BigAddr<BN, B_KMAJ> faw;
  fax.init(lds0 + PP_A0 + (uint32_t)grp * 16384u, 0, lane);
  faw.init(lds0 + PP_B0, wn * TN * 32, lane);

  u64x2 xa[NKS][TM], wb[NKS][TN];
#ifdef OFA_PP_TIMELINE
  unsigned long long tl_p0 = 0, tl_p1 = 0, tl_p2 = 0, tl_p3 = 0, tl_p4 = 0, tl_c0 = 0, tl_c1 = 0, tl_c2 = 0;
  unsigned long long tl_acc0 = 0, tl_acc1 = 0, tl_acc2 = 0, tl_acc3 = 0, tl_acc4 = 0, tl_n = 0;
  const unsigned long long tl_begin = __builtin_amdgcn_s_memtime(), tl_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
  if (nk > 0) {
    // prologue, in the loop's issue order: b(0), a(0), b(1)
    dma_b(PP_B0);
    dma_a(PP_A0);
    if (nk > 1) {
      dma_b(PP_B0 + PP_SLOT);
      pp_vmcnt<NVB>();
    } else {
      pp_vmcnt<0>();
    }
    PP_SB;
    __builtin_amdgcn_s_barrier();
    if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier interval behind group 0
    if constexpr ((ABL & 1) != 0) {
      static_for<0, NKS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        static_for<0, TM>([&](auto ic) { big_frag<HM, A_KMAJ, s, decltype(ic)::value, 0>(xa[s][decltype(ic)::value], fax); });
        static_for<0, TN>([&](auto jc) { big_frag<BN, B_KMAJ, s, decltype(jc)::value, 0>(wb[s][decltype(jc)::value], faw); });
      });
    }
    uint32_t a_wr = PP_A0 + PP_SLOT;                            // slot of a(t+1)
    uint32_t b_wr = PP_B0 + 2 * PP_SLOT;                        // slot of b(t+2)
    int b_rd = 0;                                               // slot index of B(t)
    for (int kt = 0; kt < nk; ++kt) {
      const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
      static_for<0, P>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        // ---- load segment: the phase's fragments, then this wave's DMA pieces
        PP_T(tl_c0);
        if constexpr (!(ABL & 1)) {
          static_for<0, NKS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, kk = p * NKS + s;
            static_for<0, TM>([&](auto ic) { big_frag<HM, A_KMAJ, kk, decltype(ic)::value, 0>(xa[s][decltype(ic)::value], fax); });
            static_for<0, TN>([&](auto jc) { big_frag<BN, B_KMAJ, kk, decltype(jc)::value, 0>(wb[s][decltype(jc)::value], faw); });
          });
        }
        PP_SB;
        if constexpr (p == 0 && !(ABL & 2)) {
          if (more1) static_for<0, LA>([&](auto ic) { dma_a_piece(ic, a_wr); });
        }
        if constexpr (p == P - 1) {
          if (more2) {
            if constexpr (!(ABL & 2)) static_for<0, LB>([&](auto ic) { dma_b_piece(ic, b_wr); });
            pp_vmcnt<NVA + LB>();                               // b(kt+1) has landed (a(kt+1) and the first pieces of b(kt+2) may be in flight)
          } else {
            pp_vmcnt<NVA>();                                    // no b(kt+2): only a(kt+1) may be in flight
          }
        }
        PP_T(tl_c1);
        PP_SB;
        __builtin_amdgcn_s_barrier();
        PP_SB;
        PP_T(tl_c2);
        // ---- MFMA segment
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<0, NKS>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          static_for<0, TM>([&](auto ic) { pp_pin(xa[s][decltype(ic)::value]); });
          static_for<0, TN>([&](auto jc) { pp_pin(wb[s][decltype(jc)::value]); });
        });
        PP_SB;
#ifdef OFA_PP_TIMELINE
        if (tl_p0) {
          tl_acc0 += tl_p1 - tl_p0; tl_acc1 += tl_p2 - tl_p1; tl_acc2 += tl_p3 - tl_p2; tl_acc3 += tl_p4 - tl_p3; tl_acc4 += tl_c0 - tl_p4;
          ++tl_n;
        }
        tl_p0 = tl_c0; tl_p1 = tl_c1; tl_p2 = tl_c2;
        PP_T(tl_p3);
        PP_SB;
#endif
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        static_for<0, NKS * TM * TN>([&](auto tc) {
          constexpr int n = decltype(tc)::value, NM = NKS * TM * TN;
          constexpr int s = n / (TM * TN), i = (n % (TM * TN)) / TN, j = n % TN;
          if constexpr (!(ABL & 4)) acc[i][j] = mfma16<F16>(wb[s][j], xa[s][i], acc[i][j]);
          // the pieces the load segment left over, evenly spaced between the MFMAs (a's before b's: the counted waits assume the order)
          if constexpr (!(ABL & 2)) {
            if constexpr (p == 0 && LA < NVA) {
              constexpr int R = NVA - LA;
              static_for<0, R>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (n == (r + 1) * (p == P - 1 ? NM / 2 : NM) / (R + 1) - 1) {
                  PP_SB;
                  if (more1) dma_a_piece(std::integral_constant<int, LA + r>{}, a_wr);
                  PP_SB;
                }
              });
            }
            if constexpr (p == P - 1 && LB < NVB) {
              constexpr int R = NVB - LB, N0 = (P == 1 && LA < NVA) ? NM / 2 : 0;
              static_for<0, R>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (n == N0 + (r + 1) * (NM - N0) / (R + 1) - 1) {
                  PP_SB;
                  if (more2) dma_b_piece(std::integral_constant<int, LB + r>{}, b_wr);
                  PP_SB;
                }
              });
            }
          }
        });
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        PP_SB;
        if constexpr (p == P - 1) {
          if (more2) pp_vmcnt<NVB>();                           // a(kt+1) has landed (b(kt+2) may be in flight)
          else pp_vmcnt<0>();
        }
        PP_T(tl_p4);
        PP_SB;
        __builtin_amdgcn_s_barrier();
        PP_SB;
      });
      // next K-tile: A slot toggles, B slot rotates
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fax.a[i] ^= PP_SLOT;
        faw.a[i] += PP_SLOT;
      }
      if (++b_rd == 3) {
        b_rd = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) faw.a[i] -= 3 * PP_SLOT;
      }
      a_wr ^= PP_SLOT;
      b_wr = (b_wr == PP_B0 + 2 * PP_SLOT) ? PP_B0 : b_wr + PP_SLOT;
    }
    if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();      // pairs with group 1's last barrier: every fragment read is complete
  }
  PP_SB;
#ifdef OFA_PP_TIMELINE
  if ((tid & 255) == 0 && ws) {
    unsigned long long* o = (unsigned long long*)ws + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 2 + grp) * 16;
    o[0] = tl_acc0; o[1] = tl_acc1; o[2] = tl_acc2; o[3] = tl_acc3; o[4] = tl_acc4; o[5] = tl_n;
    o[6] = __builtin_amdgcn_s_memtime() - tl_begin; o[7] = __builtin_amdgcn_s_memrealtime() - tl_rbegin;
    o[8] = (unsigned long long)nk;
  }
#endif
  {
    const bool split = to_ws || nsplit > 1;
    constexpr int REGION = 16384;                               // per wave: 8 x 16 KiB of the (now idle) stages
    unsigned char* wl = smem_raw + wave_u * REGION;
    const int m_w = m0 + grp * HM, n_w = n0 + wn * TN * 32;
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
}

template <int TM, bool A_KMAJ, bool B_KMAJ, bool OUT_F32, bool F16, int NKS, bool STAGGER, bool PRIO, int ABL = 0, int NDL = 4>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs g, int tiles_m, int tiles_n, int ksplit, float* __restrict__ ws) {
  int t, ks;
  tile_and_slice(tiles_m * tiles_n, t, ks);
  gemm_pp_body<TM, A_KMAJ, B_KMAJ, OUT_F32, F16, NKS, STAGGER, PRIO, ABL, NDL>(g, tiles_m, tiles_n, ksplit, ws, t, ks, (int)blockIdx.z,
                                                                        (int)gridDim.y, false);
}

// 160 KiB of dynamic LDS needs the attribute on every DEVICE the kernel is launched on (the flag is a bit per device, not one per
// process), and a refused attribute must not leave a launch that cannot start: false -> the planner's lockstep kernel.
static bool pp_lds_attr(const void* kern, uint64_t& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (done >> dev & 1) return true;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  done |= 1ull << dev;
  return true;
}

template <int TM, bool AK, bool BKM, bool OF, bool F16, int NKS, bool STAGGER, bool PRIO, int ABL = 0, int NDL = 4>
static bool launch_pp(const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  constexpr int BM = 64 * TM, BN = 256;
  const int tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  auto kern = gemm_pp_kernel<TM, AK, BKM, OF, F16, NKS, STAGGER, PRIO, ABL, NDL>;
  static uint64_t attr_done = 0;   // per instantiation, one bit per device
  if (!pp_lds_attr((const void*)kern, attr_done)) return false;    // (the caller falls back to the lockstep loop)
  dim3 grid(tiles_m * tiles_n, splits, batch), block(512);
  hipLaunchKernelGGL(kern, grid, block, PP_LDS, st, g, tiles_m, tiles_n, ksplit, ws);
  return true;
}

// variant: 23 = the shipped form (two k-slices per phase, staggered groups, s_setprio around the MFMA segment, THREE of a phase's four LDS-DMA
// pieces issued in its load segment and the fourth between the MFMAs: profiles/round5_gemm_pp_ab.txt).  Debug library only (tools/gemm_pp_ab.py,
// gemm_pp_ablate.py): NKS * 10 + {1: staggered + s_setprio, 0: staggered, no priority, 2: lockstep (no stagger, no priority)}, all pieces in
// the load segment; + 1000 + 100 * NDL: NDL pieces in the load segment; + 100 * ABL: timing-only ablations.
template <int TM, bool AK, bool BKM, bool OF, bool F16>
static bool launch_pp_variant(int variant, const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  switch (variant) {
    case 23: return launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 0, 3>(g, batch, splits, ksplit, ws, st);    // the shipped form
#ifdef OFA_DEBUG_SWITCHES
    case 21: launch_pp<TM, AK, BKM, OF, F16, 2, true, true>(g, batch, splits, ksplit, ws, st); return true;
    case 20: if constexpr (!F16) { launch_pp<TM, AK, BKM, OF, F16, 2, true, false>(g, batch, splits, ksplit, ws, st); return true; } return false;
    case 22: if constexpr (!F16) { launch_pp<TM, AK, BKM, OF, F16, 2, false, false>(g, batch, splits, ksplit, ws, st); return true; } return false;
    case 11: if constexpr (!F16) { launch_pp<TM, AK, BKM, OF, F16, 1, true, true>(g, batch, splits, ksplit, ws, st); return true; } return false;
    case 1021: case 1121: case 1221: case 1321:                                  // NDL sweep of variant 21
      if constexpr (!F16) {
        switch ((variant - 1000) / 100) {
          case 0: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 0, 0>(g, batch, splits, ksplit, ws, st); return true;
          case 1: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 0, 1>(g, batch, splits, ksplit, ws, st); return true;
          case 2: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 0, 2>(g, batch, splits, ksplit, ws, st); return true;
          case 3: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 0, 3>(g, batch, splits, ksplit, ws, st); return true;
        }
      }
      return false;
    case 121: case 221: case 321: case 421: case 521: case 621: case 721:        // ablations of variant 21 (NT, 256 x 256 only)
      if constexpr (TM == 4 && AK && BKM && !F16) {
        switch (variant / 100) {
          case 1: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 1>(g, batch, splits, ksplit, ws, st); return true;
          case 2: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 2>(g, batch, splits, ksplit, ws, st); return true;
          case 3: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 3>(g, batch, splits, ksplit, ws, st); return true;
          case 4: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 4>(g, batch, splits, ksplit, ws, st); return true;
          case 5: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 5>(g, batch, splits, ksplit, ws, st); return true;
          case 6: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 6>(g, batch, splits, ksplit, ws, st); return true;
          case 7: launch_pp<TM, AK, BKM, OF, F16, 2, true, true, 7>(g, batch, splits, ksplit, ws, st); return true;
        }
      }
      return false;
#endif
    default: return false;
  }
}

// Same tile / split-K / batch contract as launch_big_shape (gemm_mfma.hip); false: not a shape or variant this loop is built for (fp32
// outputs and the k-major-B / m-major-A layout stay on the lockstep loop)
bool gemm_pp_launch(int variant, const GemmArgs& g, int batch, int tm, int splits, int ksplit, float* ws, hipStream_t st, bool f16) {
  const bool ak = !g.transA, bk = g.transB != 0, of = (g.flags & OFA_GEMM_OUT_F32) != 0;
  if (of || (!ak && bk)) return false;
  if (!ak) tm = 4;
#define PP_DISPATCH(TMV, AK, BKM) \
  (f16 ? launch_pp_variant<TMV, AK, BKM, false, true>(variant, g, batch, splits, ksplit, ws, st) \
       : launch_pp_variant<TMV, AK, BKM, false, false>(variant, g, batch, splits, ksplit, ws, st))
#ifdef OFA_DEBUG_SWITCHES
  if (ak && bk) return tm == 3 ? PP_DISPATCH(3, true, true) : PP_DISPATCH(4, true, true);     // (NT: measured level with the lockstep loop; experiments only)
#else
  if (ak && bk) return false;
#endif
  if (ak && !bk) return tm == 3 ? PP_DISPATCH(3, true, false) : PP_DISPATCH(4, true, false);
  return PP_DISPATCH(4, false, false);
#undef PP_DISPATCH
}

// Grouped weight gradients (ofa_gemm_group_tn) on the same loop: both operands m-major, 256 x 256 tiles
template <bool F16, int NKS, bool STAGGER, bool PRIO, int NDL = 4>
__global__ __launch_bounds__(512) void gemm_group_tn_pp_kernel(GroupArgs ga) {
  GemmArgs g;
  int t, ks;
  const GroupItem* itp = group_enter(ga, g, t, ks);
  if (!itp) return;
  const GroupItem& it = *itp;
  gemm_pp_body<4, false, false, false, F16, NKS, STAGGER, PRIO, 0, NDL>(g, it.tiles_m, it.tiles_n, it.ksplit, it.ws, t, ks, 0, it.splits, it.out == nullptr);
}

template <bool F16, int NKS, bool STAGGER, bool PRIO, int NDL = 4>
static bool launch_group_pp(const GroupArgs& ga, hipStream_t st) {
  auto kern = gemm_group_tn_pp_kernel<F16, NKS, STAGGER, PRIO, NDL>;
  static uint64_t attr_done = 0;   // per instantiation, one bit per device
  if (!pp_lds_attr((const void*)kern, attr_done)) return false;    // (the caller launches gemm_group_tn_kernel)
  hipLaunchKernelGGL(kern, dim3(ga.total), dim3(512), PP_LDS, st, ga);
  return true;
}

bool gemm_group_pp_launch(int variant, const GroupArgs& ga, bool f16, hipStream_t st) {
  switch (variant) {
    case 23:                                                                   // the shipped form
      return f16 ? launch_group_pp<true, 2, true, true, 3>(ga, st) : launch_group_pp<false, 2, true, true, 3>(ga, st);
#ifdef OFA_DEBUG_SWITCHES
    case 21: if (f16) return false; launch_group_pp<false, 2, true, true>(ga, st); return true;
    case 20: if (f16) return false; launch_group_pp<false, 2, true, false>(ga, st); return true;
    case 22: if (f16) return false; launch_group_pp<false, 2, false, false>(ga, st); return true;
    case 11: if (f16) return false; launch_group_pp<false, 1, true, true>(ga, st); return true;
    case 1321: if (f16) return false; launch_group_pp<false, 2, true, true, 3>(ga, st); return true;   // (= 23)
    case 1221: if (f16) return false; launch_group_pp<false, 2, true, true, 2>(ga, st); return true;
#endif
    default: return false;
  }
}

}  // namespace ofa
