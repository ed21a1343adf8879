// Batched "fold the fp32 partial rows" pass.
//
// LayerNorm backward (dgamma / dbeta / fused bias column sums), bias-gradient column sums and split-K weight-gradient
// GEMMs all end with the same tiny step: out[c] (+)= alpha * sum_s part[s*stride + c].  Each used to be its own launch
// -- ~320 launches of 5-9 us per train step (2.1 ms of an 18 ms step), every one of them at the launch-latency floor.
// Their results are gradients that nobody reads before the end of backward, so the producers can leave their partials
// in place and ONE launch folds up to 56 of them (job descriptors travel by value in the kernel arguments; block ->
// (job, 128-column chunk) through a prefix table).  Summation order is fixed (slot order), so results are bit-identical
// to the per-call reduce kernels.  HBM-bound; algorithmic bytes = sum over jobs of nslots*cols*4.
#include "common.h"

namespace ofa {

constexpr int FOLD_MAX_JOBS = 56;

struct FoldJobs {
  const float* part[FOLD_MAX_JOBS];
  void* out[FOLD_MAX_JOBS];
  int64_t cols[FOLD_MAX_JOBS];
  int64_t stride[FOLD_MAX_JOBS];
  float alpha[FOLD_MAX_JOBS];
  int nslots[FOLD_MAX_JOBS];
  unsigned block0[FOLD_MAX_JOBS + 1];   // first block of each job
  unsigned char flags[FOLD_MAX_JOBS];   // bit 0: accumulate, bit 1: out is fp32 (else bf16)
  int njobs;
};

__device__ __forceinline__ void fold_store(const FoldJobs& J, int j, int64_t c, int64_t cols, const float (&t)[4]) {
  const float alpha = J.alpha[j];
  const bool acc = J.flags[j] & 1, f32 = J.flags[j] & 2, f16 = J.flags[j] & 8;
  if (c + 4 <= cols && (cols & 3) == 0) {                       // whole quad, aligned (out + c is 8 / 16-byte aligned)
    if (f32) {
      float4* o = reinterpret_cast<float4*>((float*)J.out[j] + c);
      float4 v = make_float4(t[0] * alpha, t[1] * alpha, t[2] * alpha, t[3] * alpha);
      if (acc) { const float4 u = *o; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
      *o = v;
    } else if (f16) {
      uint2* o = reinterpret_cast<uint2*>((f16_t*)J.out[j] + c);
      float v[4] = {t[0] * alpha, t[1] * alpha, t[2] * alpha, t[3] * alpha};
      if (acc) {
        const uint2 u = *o;
        const f16x2_t a = __builtin_bit_cast(f16x2_t, u.x), b = __builtin_bit_cast(f16x2_t, u.y);
        v[0] += (float)a[0]; v[1] += (float)a[1]; v[2] += (float)b[0]; v[3] += (float)b[1];
      }
      uint2 w;
      w.x = pack_f16x2(v[0], v[1]);
      w.y = pack_f16x2(v[2], v[3]);
      *o = w;
    } else {
      uint2* o = reinterpret_cast<uint2*>((bf16_t*)J.out[j] + c);
      float v[4] = {t[0] * alpha, t[1] * alpha, t[2] * alpha, t[3] * alpha};
      if (acc) {
        const uint2 u = *o;
        v[0] += __uint_as_float(u.x << 16); v[1] += __uint_as_float(u.x & 0xffff0000u);
        v[2] += __uint_as_float(u.y << 16); v[3] += __uint_as_float(u.y & 0xffff0000u);
      }
      uint2 w;
      w.x = pack_bf16x2(v[0], v[1]);
      w.y = pack_bf16x2(v[2], v[3]);
      *o = w;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (c + e >= cols) break;
    const float v = t[e] * alpha;
    if (f32) {
      float* o = (float*)J.out[j] + c + e;
      *o = acc ? *o + v : v;
    } else if (f16) {
      f16_t* o = (f16_t*)J.out[j] + c + e;
      *o = (f16_t)(acc ? (float)*o + v : v);
    } else {
      bf16_t* o = (bf16_t*)J.out[j] + c + e;
      *o = f2bf(acc ? bf2f(*o) + v : v);
    }
  }
}

// Two shapes of job: TALL (many partial rows, few columns: LayerNorm 256 x 768, column sums 128 x 768) -- a block is
// 32 column-quads x 8 slot-lanes over 128 columns; WIDE (split-K slabs: 3..16 rows of 10^5..10^6 columns) -- a block is
// 256 column-quads over 1024 columns, each thread walking the slots itself.  flags bit 2 selects WIDE.
__global__ __launch_bounds__(256) void fold_batched_kernel(FoldJobs J) {
  __shared__ float4 red[8][32];
  int j = 0;
  while (j + 1 < J.njobs && blockIdx.x >= J.block0[j + 1]) ++j;       // uniform scan over <= 56 entries
  const int64_t chunk = blockIdx.x - J.block0[j];
  const int64_t cols = J.cols[j], stride = J.stride[j];
  const float* __restrict__ p = J.part[j];
  const int ns = J.nslots[j];
  const bool vec = (cols & 3) == 0 && (stride & 3) == 0;
  if (J.flags[j] & 4) {
    const int64_t c = chunk * 1024 + threadIdx.x * 4;
    if (c >= cols) return;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      // four independent 16-byte loads in flight per thread (the slot count is a run-time value: without the manual
      // unroll every load waited for the previous add); the additions keep the slot order
      const float* r = p + c;
      int s = 0;
      for (; s + 4 <= ns; s += 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(r + (int64_t)s * stride);
        const float4 v1 = *reinterpret_cast<const float4*>(r + (int64_t)(s + 1) * stride);
        const float4 v2 = *reinterpret_cast<const float4*>(r + (int64_t)(s + 2) * stride);
        const float4 v3 = *reinterpret_cast<const float4*>(r + (int64_t)(s + 3) * stride);
        t[0] = (((t[0] + v0.x) + v1.x) + v2.x) + v3.x;
        t[1] = (((t[1] + v0.y) + v1.y) + v2.y) + v3.y;
        t[2] = (((t[2] + v0.z) + v1.z) + v2.z) + v3.z;
        t[3] = (((t[3] + v0.w) + v1.w) + v2.w) + v3.w;
      }
      // the 1..3 remaining slots (ALL of them for the 2-3 K-slice slabs of a grouped weight-gradient launch): loads issued
      // together (clamped to the last slot, dropped when beyond it), additions in slot order
      if (s < ns) {
        const int rem = ns - s;
        const float4 v0 = *reinterpret_cast<const float4*>(r + (int64_t)s * stride);
        const float4 v1 = *reinterpret_cast<const float4*>(r + (int64_t)(rem > 1 ? s + 1 : s) * stride);
        const float4 v2 = *reinterpret_cast<const float4*>(r + (int64_t)(rem > 2 ? s + 2 : s) * stride);
        t[0] += v0.x; t[1] += v0.y; t[2] += v0.z; t[3] += v0.w;
        if (rem > 1) { t[0] += v1.x; t[1] += v1.y; t[2] += v1.z; t[3] += v1.w; }
        if (rem > 2) { t[0] += v2.x; t[1] += v2.y; t[2] += v2.z; t[3] += v2.w; }
      }
    } else {
      for (int s = 0; s < ns; ++s)
        for (int e = 0; e < 4 && c + e < cols; ++e) t[e] += p[(int64_t)s * stride + c + e];
    }
    fold_store(J, j, c, cols, t);
    return;
  }
  const int q = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t c = chunk * 128 + q * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols) {
    if (vec) {
      const float* r = p + c;
      int s = sl;
      for (; s + 24 < ns; s += 32) {                      // slots s, s+8, s+16, s+24: four loads in flight
        const float4 v0 = *reinterpret_cast<const float4*>(r + (int64_t)s * stride);
        const float4 v1 = *reinterpret_cast<const float4*>(r + (int64_t)(s + 8) * stride);
        const float4 v2 = *reinterpret_cast<const float4*>(r + (int64_t)(s + 16) * stride);
        const float4 v3 = *reinterpret_cast<const float4*>(r + (int64_t)(s + 24) * stride);
        a.x = (((a.x + v0.x) + v1.x) + v2.x) + v3.x;
        a.y = (((a.y + v0.y) + v1.y) + v2.y) + v3.y;
        a.z = (((a.z + v0.z) + v1.z) + v2.z) + v3.z;
        a.w = (((a.w + v0.w) + v1.w) + v2.w) + v3.w;
      }
      for (; s < ns; s += 8) {
        const float4 v = *reinterpret_cast<const float4*>(r + (int64_t)s * stride);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    } else {
      for (int s = sl; s < ns; s += 8) {
        const float* r = p + (int64_t)s * stride + c;
        a.x += r[0];
        if (c + 1 < cols) a.y += r[1];
        if (c + 2 < cols) a.z += r[2];
        if (c + 3 < cols) a.w += r[3];
      }
    }
  }
  red[sl][q] = a;
  __syncthreads();
  if (sl == 0 && c < cols) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 8; ++r) {            // fixed order: lane r holds slots r, r+8, ...
      t[0] += red[r][q].x; t[1] += red[r][q].y; t[2] += red[r][q].z; t[3] += red[r][q].w;
    }
    fold_store(J, j, c, cols, t);
  }
}

// ---- batched device-to-device copy: the static inputs of a replayed step graph (a batch is ~30-80 small tensors: one ~3.5 us copy
// launch each otherwise).  Job descriptors travel by value; a block copies one 32 KiB chunk of one job.
constexpr int COPY_MAX_JOBS = 96;
constexpr int COPY_CHUNK = 32768;
struct CopyJobs {
  const unsigned char* src[COPY_MAX_JOBS];
  unsigned char* dst[COPY_MAX_JOBS];
  int64_t bytes[COPY_MAX_JOBS];
  unsigned block0[COPY_MAX_JOBS + 1];
  int njobs;
};

__global__ __launch_bounds__(256) void copy_batched_kernel(CopyJobs J) {
  int lo = 0, hi = J.njobs - 1;                             // the job of this block: the last one whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (J.block0[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int64_t off = (int64_t)(blockIdx.x - J.block0[lo]) * COPY_CHUNK;
  const int64_t left = J.bytes[lo] - off;
  const int n = (int)(left < COPY_CHUNK ? left : COPY_CHUNK);
  const unsigned char* s = J.src[lo] + off;
  unsigned char* d = J.dst[lo] + off;
  if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
    const int nv = n >> 4;
    for (int i = threadIdx.x; i < nv; i += 256) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    for (int i = (nv << 4) + threadIdx.x; i < n; i += 256) d[i] = s[i];
  } else {
    for (int i = threadIdx.x; i < n; i += 256) d[i] = s[i];
  }
}

}  // namespace ofa
using namespace ofa;

extern "C" int ofa_copy_batched(const ofa_copy_job* jobs, int njobs, void* stream) {
  OFA_REQUIRE(njobs >= 0 && (njobs == 0 || jobs), OFA_ERR_INVALID, "copy_batched: bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < njobs; base += COPY_MAX_JOBS) {
    CopyJobs J;
    const int n = njobs - base < COPY_MAX_JOBS ? njobs - base : COPY_MAX_JOBS;
    int m = 0;
    unsigned blocks = 0;
    for (int i = 0; i < n; ++i) {
      const ofa_copy_job& b = jobs[base + i];
      OFA_REQUIRE(b.bytes >= 0 && (b.bytes == 0 || (b.src && b.dst)), OFA_ERR_INVALID, "copy_batched: bad job %d", base + i);
      if (b.bytes == 0) continue;
      J.src[m] = (const unsigned char*)b.src; J.dst[m] = (unsigned char*)b.dst; J.bytes[m] = b.bytes;
      J.block0[m] = blocks;
      blocks += (unsigned)((b.bytes + COPY_CHUNK - 1) / COPY_CHUNK);
      ++m;
    }
    if (m == 0) continue;
    J.block0[m] = blocks;
    J.njobs = m;
    hipLaunchKernelGGL(copy_batched_kernel, dim3(blocks), dim3(256), 0, st, J);
    int rc = check_launch("copy_batched");
    if (rc) return rc;
  }
  return 0;
}

extern "C" int ofa_fold_batched(const ofa_fold_job* jobs, int njobs, void* stream) {
  OFA_REQUIRE(njobs >= 0 && (njobs == 0 || jobs), OFA_ERR_INVALID, "fold_batched: bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < njobs; base += FOLD_MAX_JOBS) {
    FoldJobs J;
    const int n = njobs - base < FOLD_MAX_JOBS ? njobs - base : FOLD_MAX_JOBS;
    J.njobs = n;
    unsigned blocks = 0;
    for (int i = 0; i < n; ++i) {
      const ofa_fold_job& b = jobs[base + i];
      OFA_REQUIRE(b.part && b.out && b.cols > 0 && b.nslots > 0 && b.stride >= b.cols, OFA_ERR_INVALID, "fold_batched: bad job %d", base + i);
      OFA_REQUIRE(OFA_DT_OK(b.out_dtype), OFA_ERR_INVALID, "fold_batched: bad out dtype %d", b.out_dtype);
      J.part[i] = b.part; J.out[i] = b.out; J.cols[i] = b.cols; J.stride[i] = b.stride; J.alpha[i] = b.alpha;
      J.nslots[i] = b.nslots;
      const bool wide = b.nslots <= 16;
      J.flags[i] = (unsigned char)((b.accumulate ? 1 : 0) | (b.out_dtype == OFA_F32 ? 2 : 0) | (wide ? 4 : 0) | (b.out_dtype == OFA_F16 ? 8 : 0));
      J.block0[i] = blocks;
      blocks += (unsigned)(wide ? (b.cols + 1023) / 1024 : (b.cols + 127) / 128);
    }
    J.block0[n] = blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(fold_batched_kernel, dim3(blocks), dim3(256), 0, st, J);
    int rc = check_launch("fold_batched");
    if (rc) return rc;
  }
  return 0;
}
