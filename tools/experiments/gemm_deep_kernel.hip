// EXPERIMENT RECORD (round 2) -- NOT part of the product build.  To reproduce: #include this file in ofasys_amd/csrc/gemm_mfma.hip in
// front of splitk_reduce_kernel, call launch_deep<AK, BKM, OF, 4 | 5> from launch_shape for a forced tile id (it was 33 / 35) and
// let gemm_plan map that id to the 128 x 128 tile count.
// RESULT (profiles/round2_gemm_deep_ring.txt): numerically identical, and 10-25 % SLOWER than the product 128 x 128 kernel on every
// shape of the train step, m-major (weight-gradient) operands included: with 4 or 5 stages the hypothesis below is refuted --
// prefetch depth is not what holds the K-step at 0.87 us.  What the variants have in common is ~0.4 us per stage barrier
// interval on top of the MFMA time (0.56 / 0.87 / 1.42 us per step for 512 / 1024 / 2048 MFMA-clocks per SIMD).
// The "deep ring": the 128 x 128 tile (64 x 64 per wave, two workgroups per CU) at K-step 32 with NS stages of 16 KiB, so that the
// operand stream runs NS - 1 K-steps (= (NS-1)/2 of the product kernel's 64-wide steps) ahead of the MFMAs in the SAME LDS budget.
// Why: tools/gemm_timeline.py puts the product kernel's K-step at 0.87 us for the pair of workgroups against 0.49 us of MFMA work,
// and every variant with a prefetch distance of ONE 64-wide step in time (128 x 128 double-buffered, 256 x 128 / 128 x 128 at
// K-step 32 with three stages) lands on the same time: the hypothesis is K-step = LDS-DMA round trip / prefetch distance.
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


template <bool A_KMAJ, bool B_KMAJ, bool OUT_F32, int NS>
__global__ __launch_bounds__(256, 2) void gemm_deep_kernel(GemmArgs g, int tiles_m, int tiles_n, int ksplit, float* __restrict__ ws) {
  constexpr int BM = 128, BN = 128, NT = 256;
  constexpr int NVA = BM * WBK / 8 / NT, NVB = BN * WBK / 8 / NT;      // 2 + 2 16-byte pieces per thread and stage
  constexpr int EA = BM * WBK, EB = BN * WBK;                          // elements per stage and operand
  constexpr uint32_t STAGE = (EA + EB) * 2;                            // 16 KiB: [A 8 KiB | B 8 KiB]
  constexpr int NPIECE = NVA + NVB;
  static_assert(NPIECE == 4, "one DMA piece per MFMA gap of the tail slice");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = tiles_m * tiles_n;
  OFA_TL_BEGIN;
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

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bf16_t* pa[NVA];
  const bf16_t* pb[NVB];
  wide_ptrs<BM, A_KMAJ, NT, NVA>(pa, A, g.lda, m0, g.M, kbeg, tid);
  wide_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, kbeg, tid, g.b_krows);
  const int64_t stepA = A_KMAJ ? WBK : (int64_t)WBK * g.lda, stepB = B_KMAJ ? WBK : (int64_t)WBK * g.ldb;
  int knext = kbeg;
  // prologue: tiles 0 .. NS-1 fill the ring
  int issued = 0;
  for (; issued < NS && issued < nk; ++issued) {
    if (!B_KMAJ && knext + WBK > g.b_krows) wide_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
    glds_issue<NT, NVA>(pa, stepA, lds + issued * (STAGE / 2), wave_u);
    glds_issue<NT, NVB>(pb, stepB, lds + issued * (STAGE / 2) + EA, wave_u);
    knext += WBK;
  }
  // tile 0 has landed when at most (issued - 1) tiles are outstanding
  if (issued >= NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * NPIECE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  OFA_TL(1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  WideAddr<BM, A_KMAJ> fax[2];
  WideAddr<BN, B_KMAJ> faw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) fax[i].init(lds0, wm * 64 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) faw[j].init(lds0 + EA * 2, wn * 64 + j * 32, lane);
  constexpr int NAA = A_KMAJ ? 2 : 1, NAB = B_KMAJ ? 2 : 1;
#define D_SB __builtin_amdgcn_sched_barrier(0)
#define D_WAIT(X, W) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(X[1]), "+v"(W[0]), "+v"(W[1]))
#define D_MF(X, W, I, J)                                                                                       \
  acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, W[J]), __builtin_bit_cast(bf16x8, X[I]), \
                                                      acc[I][J], 0, 0, 0)
  u64x2 x0[2], w0[2], x1[2], w1[2];
  if (nk > 0) {
    wide_frag<BM, A_KMAJ, 0, 0>(x0[0], fax[0]);
    wide_frag<BM, A_KMAJ, 0, 0>(x0[1], fax[1]);
    wide_frag<BN, B_KMAJ, 0, 0>(w0[0], faw[0]);
    wide_frag<BN, B_KMAJ, 0, 0>(w0[1], faw[1]);
  }
  int stage = 0;                       // the stage being multiplied
  for (int kt = 0; kt < nk; ++kt) {
    D_WAIT(x0, w0); D_SB;
    D_MF(x0, w0, 0, 0); wide_frag<BM, A_KMAJ, 1, 0>(x1[0], fax[0]); D_SB;
    D_MF(x0, w0, 0, 1); wide_frag<BM, A_KMAJ, 1, 0>(x1[1], fax[1]); D_SB;
    D_MF(x0, w0, 1, 0); wide_frag<BN, B_KMAJ, 1, 0>(w1[0], faw[0]); D_SB;
    D_MF(x0, w0, 1, 1); wide_frag<BN, B_KMAJ, 1, 0>(w1[1], faw[1]); D_SB;
    D_WAIT(x1, w1);                    // every fragment of this stage is in registers
    if (kt + 1 < nk) {
      // tile kt+1 has landed once at most the tiles behind it are outstanding (loads retire in issue order)
      const int behind = issued - kt - 2;                              // block-uniform
      if (behind >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NPIECE) : "memory");
      else if (NS >= 4 && behind == NS - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * NPIECE) : "memory");
      else if (NS >= 5 && behind == NS - 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 4) * NPIECE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      D_SB;
      __builtin_amdgcn_s_barrier();    // ... for every wave; and every wave is done reading this stage
      D_SB;
    }
    const bool more = issued < nk;     // refill the retired stage with tile kt + NS
    bf16_t* dst = lds + stage * (STAGE / 2);
    if (more && !B_KMAJ && knext + WBK > g.b_krows) wide_ptrs<BN, B_KMAJ, NT, NVB>(pb, B, g.ldb, n0, g.N, knext, tid, g.b_krows);
    // next stage: advance the fragment addresses around the ring
    const bool wrap = stage == NS - 1;
    const uint32_t delta = wrap ? (uint32_t)(0u - (uint32_t)(NS - 1) * STAGE) : STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < NAA; ++q) fax[i].a[q] += delta;
#pragma unroll
      for (int q = 0; q < NAB; ++q) faw[i].a[q] += delta;
    }
    stage = wrap ? 0 : stage + 1;
    D_SB;
    D_MF(x1, w1, 0, 0); wide_frag<BM, A_KMAJ, 0, 0>(x0[0], fax[0]);
    if (more) { __builtin_amdgcn_global_load_lds((gvoid_t*)pa[0], (lvoid_t*)(dst + (wave_u * 64 + 0 * NT) * 8), 16, 0, 0); pa[0] += stepA; }
    D_SB;
    D_MF(x1, w1, 0, 1); wide_frag<BM, A_KMAJ, 0, 0>(x0[1], fax[1]);
    if (more) { __builtin_amdgcn_global_load_lds((gvoid_t*)pa[1], (lvoid_t*)(dst + (wave_u * 64 + 1 * NT) * 8), 16, 0, 0); pa[1] += stepA; }
    D_SB;
    D_MF(x1, w1, 1, 0); wide_frag<BN, B_KMAJ, 0, 0>(w0[0], faw[0]);
    if (more) { __builtin_amdgcn_global_load_lds((gvoid_t*)pb[0], (lvoid_t*)(dst + EA + (wave_u * 64 + 0 * NT) * 8), 16, 0, 0); pb[0] += stepB; }
    D_SB;
    D_MF(x1, w1, 1, 1); wide_frag<BN, B_KMAJ, 0, 0>(w0[1], faw[1]);
    if (more) { __builtin_amdgcn_global_load_lds((gvoid_t*)pb[1], (lvoid_t*)(dst + EA + (wave_u * 64 + 1 * NT) * 8), 16, 0, 0); pb[1] += stepB; }
    D_SB;
    if (more) { knext += WBK; ++issued; }
    if ((kt & 1) == 1) OFA_TL_STEP;
  }
  if (nk > 0) D_WAIT(x0, w0);          // retire the look-ahead reads
#undef D_MF
#undef D_WAIT
#undef D_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  OFA_TL(2);
  {
    const bool split = gridDim.y > 1;
    constexpr int REGION = (int)(NS * STAGE / 4) & ~1023;
    int em0 = m0, en0 = n0, elane = lane, ewave = wave;
    asm volatile("" : "+s"(em0), "+s"(en0));
    asm volatile("" : "+v"(elane), "+v"(ewave));
    unsigned char* wl = smem_raw + ewave * REGION;
    const int m_w = em0 + (ewave >> 1) * 64, n_w = en0 + (ewave & 1) * 64;
    if (split) {
      const int64_t n4 = (g.N + 3) & ~3;
      float* wsb = ws + ((int64_t)bz * gridDim.y + ks) * g.M * n4;
      epilogue_lds<2, 2, true, true>(g, acc, wl, REGION, wsb, n4, m_w, n_w, elane);
    } else {
      const int64_t coff = batch_off(bz, g.batch_inner, g.strideC, g.strideC2);
      void* Cb = OUT_F32 ? (void*)((float*)g.C + coff) : (void*)((bf16_t*)g.C + coff);
      epilogue_lds<2, 2, OUT_F32, false>(g, acc, wl, REGION, Cb, g.ldc, m_w, n_w, elane);
    }
  }
  OFA_TL_END;
}

template <bool AK, bool BKM, bool OF, int NS>
static void launch_deep(const GemmArgs& g, int batch, int splits, int ksplit, float* ws, hipStream_t st) {
  const int tiles_m = cdiv(g.M, 128), tiles_n = cdiv(g.N, 128);
  const size_t lds = NS * (size_t)(128 + 128) * WBK * sizeof(bf16_t);
  auto kern = gemm_deep_kernel<AK, BKM, OF, NS>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid(tiles_m * tiles_n, splits, batch), block(256);
  hipLaunchKernelGGL(kern, grid, block, lds, st, g, tiles_m, tiles_n, ksplit, ws);
}
