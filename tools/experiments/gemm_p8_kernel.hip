// EXPERIMENT RECORD (round 2) -- NOT part of the product build (not in ofasys_amd/csrc/Makefile).
// The persistent form of the eight-wave 256 x 256 GEMM kernel.  Drops into ofasys_amd/csrc/gemm_mfma.hip in front of
// splitk_reduce_kernel (kernel) / launch_big_shape (launcher); selected there with `if (tm == 9) launch_p8<AK, BKM>(g, ws, st)`.
// Result (profiles/round2_gemm_timeline.txt section D, profiles/round2_gemm_eight_wave_sweep.txt): numerically identical to the
// launch-per-tile kernel; 13312 x 3072 x 768: 73.8 vs 76.5-79.4 us, x 2304: 51.8 vs 53.0-54.0 us.  The first-tile flight, the store
// acknowledgement and the re-dispatch (5 us per tile) are gone, but the epilogue grew from 2.5 to ~4 us (four passes through a 4 KiB
// bounce slice, 20-68 bytes of spills whose reloads wait vmcnt(0)) and all eight waves of the CU sit in it together.  What it would
// take: the bounce in the retired stage (delay the next tile's step-1 DMA behind the epilogue), DMA pointers as 32-bit offsets.
// (A second version did move the bounce into the retired stage -- two 64-row epilogue passes, the refill sent for behind a barrier
//  after the epilogue: correct, but hipcc then spilled 168-216 bytes per lane INTO the K loop and it ran 2x slower (150 vs 76 us).
//  The register budget, not the schedule, is what a persistent form of this kernel has to be designed around.)
// ---------------------------------------------------------------------------------------------------------------
// Persistent eight-wave kernel: the 256 x 256 tile of gemm_big_kernel<4, 2, .., 2, 4> (eight waves, 128 x 64 each, two per
// SIMD), but ONE workgroup per CU walks a list of tiles and the operand stream never stops at a tile boundary.
// Why (tools/gemm_timeline.py, 13312 x 3072 x 768): a tile lives 24 us of which 18 are K loop; the rest is the first
// tile's flight (2.6 us), the epilogue (2.5), the wait for the stores' acknowledgement before the wave may retire (0.6)
// and the dispatch of the next 512-thread / 128 KiB workgroup (1.9).  Here K-steps 0 and 1 of the NEXT tile are sent
// for during the last two K-steps of this one (the global->LDS stream is indexed by a running step number, not by tile),
// the epilogue bounces through a private 4 KiB slice of the 32 KiB of LDS the stages leave free -- so it never touches
// a buffer the stream is writing -- and its stores drain under the next tile's first K-step.
// bf16 output, no split-K, batch 1 (anything else takes the launch-per-tile kernels).
template <bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmArgs g, int tiles_m, int tiles_n, float* __restrict__ ws) {
  constexpr int TM = 4, TN = 2, WGN = 4, BM = 256, BN = 256, NT = 512;
  constexpr int NVA = BM * 8 / NT, NVB = BN * 8 / NT;
  constexpr int STAGE = 256 * BK;
  constexpr int BOUNCE = 4096;                              // per wave: 32 rows x 128 B
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sA0 = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* sA1 = sA0 + STAGE;
  bf16_t* sB0 = sA1 + STAGE;
  bf16_t* sB1 = sB0 + STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int ntiles = tiles_m * tiles_n;
#ifdef OFA_GEMM_TIMELINE
  unsigned long long* tl = (unsigned long long*)ws + 32 * blockIdx.x;
  int tlk = 8;
  if (threadIdx.x == 0) {
    tl[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    tl[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    tl[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  OFA_TL(0);
  // this workgroup's tiles: workgroups are dispatched round-robin over the 8 XCDs; XCD x owns a contiguous run of tile ids and
  // its `per` workgroups walk it side by side, so at any moment an XCD's L2 holds the operand panels of neighbouring tiles
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int run0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int run_n = q8 + (xcd < r8 ? 1 : 0);
  const int n_my = run_n > local ? (run_n - local + per - 1) / per : 0;
  if (n_my == 0) return;
  constexpr int GM = 8;
  auto origin = [&](int j, int& m0, int& n0) {
    const int t = run0 + local + j * per;
    const int gsz = GM * tiles_n;
    const int gid = t / gsz, first_m = gid * GM;
    const int rows_in_group = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    m0 = (first_m + (t % gsz) % rows_in_group) * BM;
    n0 = ((t % gsz) / rows_in_group) * BN;
  };
  const bf16_t* A = (const bf16_t*)g.A;
  const bf16_t* B = (const bf16_t*)g.B;
  const int nk = g.K / BK;                                  // launcher guarantees whole K tiles, nk >= 1

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int64_t stepA = A_KMAJ ? BK : (int64_t)BK * g.lda, stepB = B_KMAJ ? BK : (int64_t)BK * g.ldb;
  // the operand stream: (dj, dk) = tile of my list and K-step the NEXT DMA fetches
  const bf16_t* pa[NVA];
  const bf16_t* pb[NVB];
  int dj = 0, dk = 0, dn0 = 0;
  auto rebase = [&]() {
    int m0;
    origin(dj, m0, dn0);
    glds_ptrs<BM, A_KMAJ, NT, NVA>(pa, A, g.lda, m0, g.M, 0, tid);
    glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, dn0, g.N, 0, tid, g.b_krows);
  };
  auto advance = [&]() {
    if (++dk == nk) {
      dk = 0;
      if (++dj < n_my) rebase();
    }
  };
  auto clamp_tail = [&]() {
    if (!B_KMAJ && dk * BK + BK > g.b_krows)                // zero-padded contraction tail: clamp B's k rows
      glds_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, dn0, g.N, dk * BK, tid, g.b_krows);
  };
  rebase();

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  BigAddr<BM, A_KMAJ> fax;
  BigAddr<BN, B_KMAJ> faw;
  fax.init(lds0, wm * TM * 32, lane);
  faw.init(lds0 + 2 * STAGE * 2, wn * TN * 32, lane);
  f32x16 acc[TM][TN];
  u64x2 xa[2][TM], wb[2][TN];
#define P8_ISSUE(KK, SET)                                                                                         \
  static_for<0, TM>([&](auto ic) { big_frag<BM, A_KMAJ, KK, decltype(ic)::value, 0>(xa[SET][decltype(ic)::value], fax); }); \
  static_for<0, TN>([&](auto ic) { big_frag<BN, B_KMAJ, KK, decltype(ic)::value, 0>(wb[SET][decltype(ic)::value], faw); })
#define P8_WAIT(SET)                                                                                                 \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[SET][0]), "+v"(xa[SET][1]), "+v"(xa[SET][2]), "+v"(xa[SET][3]),      \
               "+v"(wb[SET][0]), "+v"(wb[SET][1]))
#define P8_SB __builtin_amdgcn_sched_barrier(0)
#define P8_SLICE(SET, KKN)                                                                                          \
  static_for<0, TM * TN>([&](auto tc) {                                                                             \
    constexpr int t = decltype(tc)::value, i = t / TN, j = t % TN;                                                  \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb[SET][j]),                     \
                                                        __builtin_bit_cast(bf16x8, xa[SET][i]), acc[i][j], 0, 0, 0); \
    constexpr int r = big_read_after(t, TM + TN);                                                                   \
    if constexpr (r >= 0 && r < TM) big_frag<BM, A_KMAJ, KKN, (r < TM ? r : 0), 0>(xa[1 - (SET)][r < TM ? r : 0], fax); \
    if constexpr (r >= TM) big_frag<BN, B_KMAJ, KKN, (r >= TM ? r - TM : 0), 0>(wb[1 - (SET)][r >= TM ? r - TM : 0], faw); \
    P8_SB;                                                                                                          \
  })
  // stream steps 0 and 1
  clamp_tail();
  glds_issue<NT, NVA>(pa, stepA, sA0, wave_u);
  glds_issue<NT, NVB>(pb, stepB, sB0, wave_u);
  advance();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  OFA_TL(1);
  if (dj < n_my) {
    clamp_tail();
    glds_issue<NT, NVA>(pa, stepA, sA1, wave_u);
    glds_issue<NT, NVB>(pb, stepB, sB1, wave_u);
    advance();
  }
  bf16_t* curA = sA0;                  // the stage being multiplied; its buffers are refilled with stream step s+2
  bf16_t* curB = sB0;
  for (int j = 0; j < n_my; ++j) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    P8_ISSUE(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      P8_WAIT(0); P8_SB; P8_SLICE(0, 1);
      P8_WAIT(1); P8_SB; P8_SLICE(1, 2);
      P8_WAIT(0); P8_SB; P8_SLICE(0, 3);
      P8_WAIT(1);
      const bool last_step = (j + 1 == n_my) && (kt + 1 == nk);
      if (!last_step) {
        // the next stream step has landed (explicit vmcnt: hipcc does not see the DMA as a load) and every wave is done
        // reading this stage
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      const bool more = dj < n_my;        // refill the retired stage with stream step s+2 ...
      if (more) clamp_tail();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fax.a[i] ^= (uint32_t)(STAGE * 2);
        faw.a[i] ^= (uint32_t)(STAGE * 2);
      }
      P8_SB;
      // ... one LDS-DMA piece per MFMA gap; the first-slice fragments of the next K-step are read in the same gaps (at a
      // tile's last step they belong to the next tile and are read again behind the epilogue; after the very last step
      // they are stale, waited for and dropped)
      static_for<0, TM * TN>([&](auto tc) {
        constexpr int t = decltype(tc)::value, i = t / TN, jj = t % TN;
        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb[1][jj]),
                                                             __builtin_bit_cast(bf16x8, xa[1][i]), acc[i][jj], 0, 0, 0);
        constexpr int r = big_read_after(t, TM + TN);
        if constexpr (r >= 0 && r < TM) big_frag<BM, A_KMAJ, 0, (r < TM ? r : 0), 0>(xa[0][r < TM ? r : 0], fax);
        if constexpr (r >= TM) big_frag<BN, B_KMAJ, 0, (r >= TM ? r - TM : 0), 0>(wb[0][r >= TM ? r - TM : 0], faw);
        if (more) {
          static_assert(NVA + NVB == TM * TN, "one DMA piece per MFMA gap");
          if constexpr (t < NVA) {
            __builtin_amdgcn_global_load_lds((gvoid_t*)pa[t], (lvoid_t*)(curA + (wave_u * 64 + t * NT) * 8), 16, 0, 0);
            pa[t] += stepA;
          } else {
            __builtin_amdgcn_global_load_lds((gvoid_t*)pb[t - NVA], (lvoid_t*)(curB + (wave_u * 64 + (t - NVA) * NT) * 8), 16, 0, 0);
            pb[t - NVA] += stepB;
          }
        }
        P8_SB;
      });
      if (more) advance();
      curA = (bf16_t*)((uintptr_t)curA ^ (uintptr_t)(STAGE * 2));
      curB = (bf16_t*)((uintptr_t)curB ^ (uintptr_t)(STAGE * 2));
    }
    P8_WAIT(0);                        // retire the look-ahead reads: the epilogue needs the registers
    P8_SB;
#ifdef OFA_GEMM_TIMELINE
    OFA_TL(tlk); ++tlk;
#endif
    {
      // (the tile index and lane id are laundered: otherwise hipcc computes the epilogue's ~50 address registers in front of
      //  the K loop and spills them around it)
      int je = j, lane_e = lane, wm_e = wm, wn_e = wn;
      asm volatile("" : "+s"(je));
      asm volatile("" : "+v"(lane_e), "+v"(wm_e), "+v"(wn_e));
      int m0, n0;
      origin(je, m0, n0);
      epilogue_lds<TM, TN, false, false>(g, acc, smem_raw + 4 * STAGE * 2 + (lane_e >> 6) * 0 + wave_u * BOUNCE, BOUNCE, g.C, g.ldc,
                                         m0 + wm_e * TM * 32, n0 + wn_e * TN * 32, lane_e);
    }
#ifdef OFA_GEMM_TIMELINE
    OFA_TL(tlk); ++tlk;
#endif
  }
#undef P8_SLICE
#undef P8_SB
#undef P8_WAIT
#undef P8_ISSUE
#ifdef OFA_GEMM_TIMELINE
  OFA_TL(2);
  OFA_TL(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OFA_TL(4);
  if (threadIdx.x == 0) tl[31] = __builtin_amdgcn_s_memrealtime();
#endif
}


// ---- launcher
template <bool AK, bool BKM>
static void launch_p8(const GemmArgs& g, float* ws, hipStream_t st) {
  const int tiles_m = cdiv(g.M, 256), tiles_n = cdiv(g.N, 256);
  const int ntiles = tiles_m * tiles_n;
  const size_t lds = 4 * (size_t)256 * BK * sizeof(bf16_t) + 8 * 4096;   // 2 operands x 2 stages x 32 KiB + 8 bounce slices
  auto kern = gemm_p8_kernel<AK, BKM>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  static const int cus = [] {
    int dev = 0, n = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? (n & ~7) : 8;
  }();
  const int wgs = ntiles >= cus ? cus : ((ntiles + 7) & ~7);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, st, g, tiles_m, tiles_n, ws);
}

