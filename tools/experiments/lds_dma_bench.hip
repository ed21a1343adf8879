// Microbenchmark (round 5): how fast can ONE compute unit move L2-resident operand tiles into LDS, by path and by number of issuing waves?
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, 64-bit per-lane addresses)           -- what the GEMM kernels use
//   mode 1  global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging)
//   mode 2  buffer_load_dwordx4 ... lds (LDS-DMA through a buffer descriptor, 32-bit per-lane offsets)
//   mode 3  half of the pieces by mode 0, half by mode 1 (do the two paths add up?)
// One 512-thread workgroup per CU, W of its 8 waves issue (the others exit); every issuing wave moves `pieces` 1 KiB pieces per round
// (rows of 128 B, 8 rows per piece, like a k-major GEMM tile) from a 2 MiB window that stays in every XCD's L2, keeping <= 2 rounds in flight.
// Prints bytes per shader clock per CU (s_memtime of wave 0) and the chip-wide rate.
//   hipcc --offload-arch=gfx950 -O3 -o _build/lds_dma_bench lds_dma_bench.hip && _build/lds_dma_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int PIECES>
__global__ __launch_bounds__(512) void bench(const uint4* __restrict__ src, int waves, int rounds, unsigned long long* out, uint32_t window16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= waves) return;
  // per-lane element offset (16-byte units): 8 lanes cover one 128-byte row, rows 4 KiB apart (a [rows][K] operand with a long K)
  uint32_t off = ((uint32_t)blockIdx.x * 4099u + (uint32_t)wave * 257u) * 8u;
  const uint32_t lane_off = (uint32_t)(lane >> 3) * 256u + (uint32_t)(lane & 7);
  unsigned char* my = smem + wave * (PIECES * 2 * 1024);          // 2 rounds of pieces per wave
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(window16 * 16u), 0x00020000);
  uint4 regs[PIECES];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    unsigned char* dst = my + (r & 1) * (PIECES * 1024);
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const uint32_t e = (off + (uint32_t)p * 2048u + lane_off) % window16;
      const bool dma = MODE == 0 || (MODE == 3 && (p & 1) == 0);
      if (MODE == 2) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lvoid_t*)(dst + p * 1024), 16, (int)(e * 16u), 0, 0, 0);
      } else if (dma) {
        __builtin_amdgcn_global_load_lds((gvoid_t*)(src + e), (lvoid_t*)(dst + p * 1024), 16, 0, 0);
      } else {
        regs[p] = src[e];
      }
    }
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p)
        if (MODE == 1 || (p & 1)) *reinterpret_cast<uint4*>(dst + p * 1024 + lane * 16) = regs[p];
    }
    off += PIECES * 2048u + 64u;
    // keep at most one older round in flight (counted wait; the DMA rounds are never drained to zero)
    if (MODE == 0 || MODE == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIECES) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (smem[tid] == 0x5a && rounds < 0) out[0] = 0;                // keep the LDS image alive
}

template <int MODE, int PIECES>
static void run(const char* name, const uint4* src, unsigned long long* out, uint32_t window16, int nblk) {
  for (int waves : {1, 2, 4, 8}) {
    const int rounds = 2000;
    size_t lds = 8 * PIECES * 2 * 1024;
    hipFuncSetAttribute((const void*)bench<MODE, PIECES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<MODE, PIECES><<<nblk, 512, lds>>>(src, waves, 50, out, window16);
    hipEventRecord(e0);
    bench<MODE, PIECES><<<nblk, 512, lds>>>(src, waves, rounds, out, window16);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), out, nblk * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto c : h) cyc += (double)c;
    cyc /= nblk;
    const double bytes = (double)waves * rounds * PIECES * 1024.0;
    printf("%-28s pieces/round %d  waves %d: %6.1f B/clk/CU (%.0f clk per piece and CU)   chip %.2f TB/s   %.1f us\n", name, PIECES, waves, bytes / cyc,
           cyc / (bytes / 1024.0), bytes * nblk / (ms * 1e-3) / 1e12, ms * 1e3);
  }
}

int main() {
  const uint32_t window16 = (2u << 20) / 16;
  uint4* src;
  unsigned long long* out;
  hipMalloc(&src, (size_t)window16 * 16 + 65536);
  hipMemset(src, 1, (size_t)window16 * 16 + 65536);
  hipMalloc(&out, 4096 * 8);
  const int nblk = 256;
  run<0, 8>("global_load_lds x4", src, out, window16, nblk);
  run<0, 4>("global_load_lds x4", src, out, window16, nblk);
  run<2, 8>("buffer_load lds x4", src, out, window16, nblk);
  run<1, 8>("global_load + ds_write_b128", src, out, window16, nblk);
  run<3, 8>("half DMA, half registers", src, out, window16, nblk);
  return 0;
}
