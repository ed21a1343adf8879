// EXPERIMENT RECORD (round 2) -- NOT part of the product build (not in ofasys_amd/csrc/Makefile).
// The "three-slot" GEMM kernel: 128 x 128 tile, 64 x 64 per wave, K-step 32, 3-stage LDS-DMA ring, 119-144 registers and 48 KiB of LDS
// -> three (by registers four) workgroups per CU.  Drops into ofasys_amd/csrc/gemm_mfma.hip in front of splitk_reduce_kernel and is
// selected with OFA_GEMM_TILE=33.  Result (profiles/round2_gemm_tri_experiment.txt): numerically identical to the 128 x 128
// kernel; NT products 8-27 % slower, NN / TN products within +-4 %.  A third resident workgroup does not buy the tail back.
// ---------------------------------------------------------------------------------------------------------------
// Three-slot kernel: the 128 x 128 workgroup tile (64 x 64 per wave) at K-step 32 with a 3-stage LDS-DMA ring: 48 KiB of LDS and
// <= 170 registers, so THREE workgroups share a CU.  Why: a workgroup's tail -- accumulators -> LDS -> row segments -> stores,
// then the stores' acknowledgement before the wave may retire -- holds its slot for 3-4 us without issuing an MFMA.  With two
// slots per CU and ~8 us of K loop per tile at K = 768 that tail is the "30% epilogue" measured in round 1 (46.9 us without the
// epilogue vs 64-67 with, 14336 x 2304 x 768); the main loop itself is identical in speed across the 128 x 128, 256 x 128 and ring
// variants (round-2 experiments), so the lever is a third slot that computes while the other two drain.
// LDS images (LDS-DMA, lane-linear; swizzle on the source address and again on the read):
//   k-major [R][32]: 64-byte rows, chunk c (4 per row) of row r lives at c ^ ((r>>2)&3) -> the 16 rows of a ds_read_b128 lane
//                    group cover all 16 16-byte slots of the 256-byte bank row;
//   m-major [32][R]: as in the kernels above (swz<R, false>), read with ds_read_b64_tr_b16.
constexpr int WBK = 32;
__device__ __forceinline__ int swz32(int row, int c) { return c ^ ((row >> 2) & 3); }

template <int R, bool KMAJ, int NT, int NV>
__device__ __forceinline__ void wide_ptrs(const bf16_t* (&ptr)[NV], const bf16_t* __restrict__ base, int64_t ld, int r0, int rmax,
                                          int k0, int tid, int krows = 0x7fffffff) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int gidx = tid + i * NT;
    if (KMAJ) {
      const int r = gidx >> 2, c = swz32(r, gidx & 3);
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
      kr = kr < krows ? kr : krows - 1;
      ptr[i] = base + (int64_t)kr * ld + col;
    }
  }
}

template <int R, bool KMAJ> struct WideAddr {
  uint32_t a[KMAJ ? 2 : 1];          // k-major: one per k-slice (the swizzle depends on the chunk); m-major: base
  __device__ __forceinline__ void init(uint32_t tile0, int rbase, int lane) {
    if (KMAJ) {
      const int row = rbase + (lane & 31), hi = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) a[kk] = tile0 + (uint32_t)(row * WBK + swz32(row, kk * 2 + hi) * 8) * 2u;
    } else {
      const int g = lane >> 4, q = lane & 15;
      const int k = (g >> 1) * 8 + (q >> 2);
      const int col = rbase + (g & 1) * 16 + 4 * (q & 3);
      a[0] = tile0 + (uint32_t)(k * R + swz<R, false>(k, col >> 3) * 8 + (col & 7)) * 2u;
    }
  }
};

template <int R, bool KMAJ, int KK, int BUFOFF>
__device__ __forceinline__ void wide_frag(u64x2& d, const WideAddr<R, KMAJ>& fa) {
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

template <bool A_KMAJ, bool B_KMAJ, bool OUT_F32>
__global__ __launch_bounds__(256, 3) void gemm_tri_kernel(GemmArgs g, int tiles_m, int tiles_n, int ksplit,
                                                         float* __restrict__ ws) {
  constexpr int BM = 128, BN = 128, NT = 256, TM = 2, TN = 2;
  constexpr int NVA = BM * WBK / 8 / NT, NVB = BN * WBK / 8 / NT;      // 2 + 2 16-byte pieces per thread and stage
  constexpr int EA = BM * WBK, EB = BN * WBK;                          // elements per stage and operand
  constexpr int STAGE = (EA + EB) * 2;                                 // 16 KiB: [A 8 KiB | B 8 KiB]
  constexpr int NPIECE = NVA + NVB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
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
  const int nk = (kend - kbeg) / WBK;                     // launcher guarantees whole K-steps

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
  wide_ptrs<BM, A_KMAJ, NT, NVA>(pa, A, g.lda, m0, g.M, kbeg, tid);
  wide_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, kbeg, tid, g.b_krows);
  const int64_t stepA = A_KMAJ ? WBK : (int64_t)WBK * g.lda, stepB = B_KMAJ ? WBK : (int64_t)WBK * g.ldb;
  int knext = kbeg;
  auto dma = [&](int st) {             // one K-step of both operands into stage `st` (NPIECE LDS-DMA instructions per thread)
    if (!B_KMAJ && knext + WBK > g.b_krows)               // zero-padded contraction tail: clamp B's k rows
      wide_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
    glds_issue<NT, NVA>(pa, stepA, lds + st * (STAGE / 2), wave_u);
    glds_issue<NT, NVB>(pb, stepB, lds + st * (STAGE / 2) + EA, wave_u);
    knext += WBK;
  };
  // K-steps kt+1 and kt+2 travel while kt multiplies; the stage barrier is the bare s_barrier behind a COUNTED vmcnt (only the
  // older of the two tiles in flight must have landed -- __syncthreads() would drain both with its implicit vmcnt(0))
  if (nk > 0) dma(0);
  if (nk > 1) dma(1);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  WideAddr<BM, A_KMAJ> fax[TM];
  WideAddr<BN, B_KMAJ> faw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fax[i].init(lds0, wm * 64 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < TN; ++j) faw[j].init(lds0 + EA * 2, wn * 64 + j * 32, lane);
#define T_ISSUE(KK, X, W, OFF)                                                                           \
  static_for<0, TM>([&](auto ic) { wide_frag<BM, A_KMAJ, KK, OFF>(X[decltype(ic)::value], fax[decltype(ic)::value]); }); \
  static_for<0, TN>([&](auto jc) { wide_frag<BN, B_KMAJ, KK, OFF>(W[decltype(jc)::value], faw[decltype(jc)::value]); })
#define T_WAIT(X, W) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(X[1]), "+v"(W[0]), "+v"(W[1]))
#define T_MMA(X, W)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                        \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, W[j]),                            \
                                                          __builtin_bit_cast(bf16x8, X[i]), acc[i][j], 0, 0, 0)
#define T_KSTEP(CUR, NXT2)                                                                              \
  {                                                                                                     \
    const bool more2 = kt + 2 < nk;                                                                     \
    if (more2) dma(NXT2);              /* its previous contents (K-step kt-1) were retired by the last barrier */ \
    u64x2 x0[TM], w0[TN], x1[TM], w1[TN];                                                               \
    T_ISSUE(0, x0, w0, CUR * STAGE);                                                                    \
    T_WAIT(x0, w0);                                                                                     \
    T_ISSUE(1, x1, w1, CUR * STAGE);   /* the second k-slice's LDS reads fly under the first slice's MFMAs */ \
    T_MMA(x0, w0);                                                                                      \
    T_WAIT(x1, w1);                                                                                     \
    T_MMA(x1, w1);                                                                                      \
    if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");   /* kt+1 landed; kt+2 may still fly */ \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
    __builtin_amdgcn_s_barrier();      /* ... for every wave; and everybody is done reading stage CUR */ \
    ++kt;                                                                                               \
  }
  int kt = 0;
  while (kt + 2 < nk) {
    T_KSTEP(0, 2);
    T_KSTEP(1, 0);
    T_KSTEP(2, 1);
  }
  if (kt < nk) T_KSTEP(0, 2);
  if (kt < nk) T_KSTEP(1, 0);
#undef T_KSTEP
#undef T_ISSUE
#undef T_WAIT
#undef T_MMA

  // epilogue through the wave's 12 KiB slice of the (idle) stages; the last stage barrier above already ordered every read.
  // The tile origin and lane id are laundered: everything the epilogue derives from them (store pointers, swizzled staging
  // offsets, bias addresses) would otherwise be hoisted above the K loop and cost registers through it.
  {
    const bool split = gridDim.y > 1;
    constexpr int REGION = 3 * STAGE / 4;
    int em0 = m0, en0 = n0, elane = lane, ewave = wave;
    asm volatile("" : "+s"(em0), "+s"(en0));
    asm volatile("" : "+v"(elane), "+v"(ewave));
    unsigned char* wl = smem_raw + ewave * REGION;
    const int m_w = em0 + (ewave >> 1) * 64, n_w = en0 + (ewave & 1) * 64;
    if (split) {
      const int64_t n4 = (g.N + 3) & ~3;
      float* wsb = ws + ((int64_t)bz * gridDim.y + ks) * g.M * n4;
      epilogue_lds<TM, TN, true, true>(g, acc, wl, REGION, wsb, n4, m_w, n_w, elane);
    } else {
      const int64_t coff = batch_off(bz, g.batch_inner, g.strideC, g.strideC2);
      void* Cb = OUT_F32 ? (void*)((float*)g.C + coff) : (void*)((bf16_t*)g.C + coff);
      epilogue_lds<TM, TN, OUT_F32, false>(g, acc, wl, REGION, Cb, g.ldc, m_w, n_w, elane);
    }
  }
}


template <bool AK, bool BKM, bool OF>
static void launch_tri(const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  const int tiles_m = cdiv(g.M, 128), tiles_n = cdiv(g.N, 128);
  const size_t lds = 3 * (size_t)(128 + 128) * WBK * sizeof(bf16_t);   // three stages of 16 KiB: three workgroups per CU
  auto kern = gemm_tri_kernel<AK, BKM, OF>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(tiles_m * tiles_n, splits, batch), block(256);
  hipLaunchKernelGGL(kern, grid, block, lds, st, g, tiles_m, tiles_n, ksplit, ws);
}

