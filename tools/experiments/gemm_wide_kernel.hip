// EXPERIMENT RECORD (round 2) -- NOT part of the product build (not in ofasys_amd/csrc/Makefile).
// The 256 x 128 "wide-wave" GEMM kernel (128 x 64 per wave, K-step 32, 3-stage LDS-DMA ring, optional persistent tile loop with the
// next tile's prefetch issued in front of the epilogue) as it was measured in profiles/round2_gemm_wide_experiment.txt.  It drops
// into ofasys_amd/csrc/gemm_mfma.hip in front of splitk_reduce_kernel (it uses that file's swz<>, glds_issue, xcd_remap,
// epilogue_lds-with-hook and static_for) and is selected with OFA_GEMM_TILE=42.  Result: numerically identical to the 128 x 128
// kernel, same speed non-persistent, 18 % slower persistent (DESIGN.md sections 5a and 8).
// ---------------------------------------------------------------------------------------------------------------
// Wide-wave kernel: 256 x 128 workgroup tile, 4 waves (2 x 2), each owning 128 x 64 = 4 x 2 MFMA tiles (128 fp32 accumulator
// registers), K-step 32, two LDS stages of 24 KiB -> 48 KiB per workgroup, TWO workgroups per CU.  Why: the 128 x 128
// kernel above reads 4 fragments per 4 MFMAs (16 KiB per wave per 16 MFMAs) -- 128 B/clk/CU, the LDS peak, before the LDS-DMA
// writes are counted -- so it is LDS-bound at ~40% MFMA utilisation; the 256 x 256 kernel (128 x 128 per wave) halves that
// traffic but runs one workgroup per CU, which leaves its output burst exposed on K = 768 products.  A 128 x 64 wave tile
// reads 6 fragments per 8 MFMAs (12 KiB per 16 MFMAs, -25%) and stages 24 instead of 32 KiB per 16 MFMAs (-25%), and at
// K-step 32 two workgroups still fit a CU, so one's epilogue / stage barrier hides under the other's MFMAs.
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

// PERSISTENT form (PERSIST, launched with min(tiles, 2 x CUs) workgroups, no split-K): a workgroup walks tiles blockIdx.x,
// + gridDim.x, ...  Why: a workgroup's tail -- accumulators -> LDS -> row segments -> stores, then the stores' round trip before
// the wave may retire -- keeps its CU slot for 3-4 us without issuing a single MFMA; with two slots per CU and ~8 us of K loop
// per tile at K = 768 that is the "30% epilogue" of DESIGN.md section 5 (the instructions themselves are ~300).  Here the next
// tile's first two K-steps are sent for (LDS-DMA into stages 0 and 1) BEFORE this tile's epilogue, which stages through stage 2,
// and the K loop resumes behind a COUNTED wait that leaves the epilogue's own stores in flight: vmcnt retires in issue order
// (loads and stores share the counter on gfx950), so "at most NSTORE + 6 outstanding" means the older stage-0 DMA has landed.
template <bool A_KMAJ, bool B_KMAJ, bool OUT_F32, bool PERSIST>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(GemmArgs g, int tiles_m, int tiles_n, int ksplit,
                                                          float* __restrict__ ws) {
  constexpr int BM = 256, BN = 128, NT = 256, TM = 4, TN = 2;
  constexpr int NVA = BM * WBK / 8 / NT, NVB = BN * WBK / 8 / NT;      // 4 and 2 16-byte pieces per thread and stage
  constexpr int EA = BM * WBK, EB = BN * WBK;                          // elements per stage and operand
  constexpr int STAGE = (EA + EB) * 2;                                 // 24 KiB: [A 16 KiB | B 8 KiB]
  // store instructions per lane of an interior tile's epilogue (epilogue_lds fast path): 128 rows / rows-per-instruction
  constexpr int NSTORE = OUT_F32 ? 128 / (64 / (TN * 32 * 4 / 16)) : 128 / (64 / (TN * 32 * 2 / 16));
  static_assert(NSTORE + 6 <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = tiles_m * tiles_n;
  const int bz = blockIdx.z, ks = blockIdx.y;
  const bf16_t* A = (const bf16_t*)g.A + batch_off(bz, g.batch_inner, g.strideA, g.strideA2);
  const bf16_t* B = (const bf16_t*)g.B + batch_off(bz, g.batch_inner, g.strideB, g.strideB2);
  const int kbeg = ks * ksplit;
  const int kend = (kbeg + ksplit < g.K) ? kbeg + ksplit : g.K;
  const int nk = (kend - kbeg) / WBK;                     // launcher guarantees whole K-steps
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int64_t stepA = A_KMAJ ? WBK : (int64_t)WBK * g.lda, stepB = B_KMAJ ? WBK : (int64_t)WBK * g.ldb;

  const bf16_t* pa[NVA];
  const bf16_t* pb[NVB];
  int m0 = 0, n0 = 0, knext = kbeg;
  // tile id (position in launch order, XCD-contiguous remap) -> tile origin and DMA source pointers
  auto enter_tile = [&](int id) {
    const int t = xcd_remap(id, ntiles);
    constexpr int GM = 4;                                 // 4 x 256 rows: the same A panel height as the 128-row kernels' GM = 8
    const int gsz = GM * tiles_n;
    const int gid = t / gsz, first_m = gid * GM;
    const int rows_in_group = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int tm = first_m + (t % gsz) % rows_in_group, tn = (t % gsz) / rows_in_group;
    m0 = tm * BM;
    n0 = tn * BN;
    wide_ptrs<BM, A_KMAJ, NT, NVA>(pa, A, g.lda, m0, g.M, kbeg, tid);
    wide_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, kbeg, tid, g.b_krows);
    knext = kbeg;
  };
  // one K-step of both operands into stage `st` (6 LDS-DMA instructions per thread)
  auto dma = [&](int st) {
    if (!B_KMAJ && knext + WBK > g.b_krows)               // zero-padded contraction tail: clamp B's k rows
      wide_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
    glds_issue<NT, NVA>(pa, stepA, lds + st * (STAGE / 2), wave_u);
    glds_issue<NT, NVB>(pb, stepB, lds + st * (STAGE / 2) + EA, wave_u);
    knext += WBK;
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  WideAddr<BM, A_KMAJ> fax[TM];
  WideAddr<BN, B_KMAJ> faw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fax[i].init(lds0, wm * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < TN; ++j) faw[j].init(lds0 + EA * 2, wn * 64 + j * 32, lane);

  int tile = blockIdx.x;
  enter_tile(tile);
  // THREE stages: K-steps kt+1 and kt+2 are travelling while kt multiplies.  A K-step is only ~0.25 us of MFMAs per
  // workgroup, far less than the global -> LDS latency, so with one step of look-ahead (the double-buffered kernels above) every
  // step waits for memory; the stage barrier here is the bare s_barrier behind `vmcnt(6)` -- only the OLDER of the two tiles in
  // flight must have landed (`__syncthreads()` would drain both with its implicit vmcnt(0)).
  if (nk > 0) dma(0);
  if (nk > 1) dma(1);
  bool stores_behind = false;          // this tile's prologue DMAs were issued in front of the previous epilogue's stores
  for (;;) {
    const int cur_m0 = m0, cur_n0 = n0;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // stage 0 has landed (queue, oldest first: dma0, [dma1], [previous epilogue's stores])
    if (PERSIST && stores_behind && nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE + 6) : "memory");
    else if (nk > 1 && !stores_behind) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#define W_ISSUE(KK, X, W, OFF)                                                                           \
  static_for<0, TM>([&](auto ic) { wide_frag<BM, A_KMAJ, KK, OFF>(X[decltype(ic)::value], fax[decltype(ic)::value]); }); \
  static_for<0, TN>([&](auto jc) { wide_frag<BN, B_KMAJ, KK, OFF>(W[decltype(jc)::value], faw[decltype(jc)::value]); })
#define W_WAIT(X, W) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(W[0]), "+v"(W[1]))
#define W_MMA(X, W)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                        \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, W[j]),                            \
                                                          __builtin_bit_cast(bf16x8, X[i]), acc[i][j], 0, 0, 0)
    // K-step kt on stage CUR (compile-time LDS offsets); refills stage NXT2 = (CUR + 2) % 3 with K-step kt + 2
#define W_KSTEP(CUR, NXT2)                                                                              \
  {                                                                                                     \
    const bool more2 = kt + 2 < nk;                                                                     \
    if (more2) dma(NXT2);              /* its previous contents (K-step kt-1) were retired by the last barrier */ \
    u64x2 x0[TM], w0[TN], x1[TM], w1[TN];                                                               \
    W_ISSUE(0, x0, w0, CUR * STAGE);                                                                    \
    W_WAIT(x0, w0);                                                                                     \
    W_ISSUE(1, x1, w1, CUR * STAGE);   /* the second k-slice's LDS reads fly under the first slice's MFMAs */ \
    W_MMA(x0, w0);                                                                                      \
    W_WAIT(x1, w1);                                                                                     \
    W_MMA(x1, w1);                                                                                      \
    /* K-step kt+1 has landed; kt+2 may still fly (and, right behind a tile switch, the last epilogue's stores, which */ \
    /* sit between the two in the queue) */                                                             \
    if (PERSIST && stores_behind && kt == 0 && more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE + 6) : "memory"); \
    else if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                    \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
    __builtin_amdgcn_s_barrier();      /* ... for every wave; and everybody is done reading stage CUR */ \
    ++kt;                                                                                               \
  }
    int kt = 0;
    while (kt + 2 < nk) {
      W_KSTEP(0, 2);
      W_KSTEP(1, 0);
      W_KSTEP(2, 1);
    }
    if (kt < nk) W_KSTEP(0, 2);
    if (kt < nk) W_KSTEP(1, 0);
#undef W_KSTEP
#undef W_ISSUE
#undef W_WAIT
#undef W_MMA
    // every DMA of this tile has landed and every wave is past its last fragment read (the final stage barrier)
    const int next = tile + (int)gridDim.x;
    const bool more_tiles = PERSIST && next < ntiles;
    const bool interior = cur_m0 + BM <= g.M && cur_n0 + BN <= g.N && !(g.flags & OFA_GEMM_ACCUM) &&
                          (OUT_F32 || (g.ldc & 7) == 0);
    auto prefetch = [&]() {
      if (more_tiles) {                  // the next tile's first two K-steps go out IN FRONT of this tile's stores
        enter_tile(next);
        if (nk > 0) dma(0);
        if (nk > 1) dma(1);
      }
    };
    const bool split = gridDim.y > 1;
    // launder the tile origin and the lane id: everything the epilogue derives from them (store pointers, swizzled staging
    // offsets, bias addresses) would otherwise be hoisted above the K loop and spilled around it -- and every reload in the
    // epilogue comes with a compiler-made vmcnt(0) that waits for the stores just issued
    int em0 = cur_m0, en0 = cur_n0, elane = lane, ewave = wave;
    asm volatile("" : "+s"(em0), "+s"(en0));
    asm volatile("" : "+v"(elane), "+v"(ewave));
    const int m_w = em0 + (ewave >> 1) * 128, n_w = en0 + (ewave & 1) * 64;
    if (!PERSIST && split) {             // (never persistent)
      constexpr int REGION = 3 * STAGE / 4;            // 18 KiB per wave
      const int64_t n4 = (g.N + 3) & ~3;
      float* wsb = ws + ((int64_t)bz * gridDim.y + ks) * g.M * n4;
      epilogue_lds<TM, TN, true, true>(g, acc, smem_raw + ewave * REGION, REGION, wsb, n4, m_w, n_w, elane);
    } else {
      // persistent: stage through stage 2 only (6 KiB per wave) -- stages 0 and 1 are being refilled
      constexpr int REGION = PERSIST ? STAGE / 4 : 3 * STAGE / 4;
      static_assert(!PERSIST || !OUT_F32, "the fp32 epilogue needs 8 KiB per 32-row tile");
      unsigned char* wl = smem_raw + (PERSIST ? 2 * STAGE : 0) + ewave * REGION;
      const int64_t coff = batch_off(bz, g.batch_inner, g.strideC, g.strideC2);
      void* Cb = OUT_F32 ? (void*)((float*)g.C + coff) : (void*)((bf16_t*)g.C + coff);
      epilogue_lds<TM, TN, OUT_F32, false>(g, acc, wl, REGION, Cb, g.ldc, m_w, n_w, elane, prefetch);
    }
    if (!more_tiles) break;
    tile = next;
    stores_behind = interior;            // an edge / accumulating tile stores through the guarded path: drain everything
    if (!interior) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}


template <bool AK, bool BKM, bool OF>
static void launch_wide(const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  const int tiles_m = cdiv(g.M, 256), tiles_n = cdiv(g.N, 128);
  const size_t lds = 3 * (size_t)(256 + 128) * WBK * sizeof(bf16_t);   // three stages of 24 KiB: two workgroups per CU
  static const int persist_env = getenv("OFA_GEMM_PERSIST") ? atoi(getenv("OFA_GEMM_PERSIST")) : 1;   // experiments: 0 = one tile per workgroup
  if constexpr (!OF) {
    if (splits == 1 && persist_env) {
      auto kern = gemm_wide_kernel<AK, BKM, OF, true>;
      static bool attr_done = false;
      if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
      }
      const int tiles = tiles_m * tiles_n;
      dim3 grid(tiles < 512 ? tiles : 512, 1, batch), block(256);       // two resident workgroups per CU walk the tiles
      hipLaunchKernelGGL(kern, grid, block, lds, st, g, tiles_m, tiles_n, ksplit, ws);
      return;
    }
  }
  auto kern = gemm_wide_kernel<AK, BKM, OF, false>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(tiles_m * tiles_n, splits, batch), block(256);
  hipLaunchKernelGGL(kern, grid, block, lds, st, g, tiles_m, tiles_n, ksplit, ws);
}

