// Standalone microbenchmark: how fast can the Adam pass (28 B per bf16 parameter: fp32 master / m / v read + written, bf16 gradient
// read, bf16 model copy written) stream on an MI355X, as a function of grid size, elements per thread and load / store flavour?
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/_build/adam_stream_bench tools/experiments/adam_stream_bench.hip
//   tools/experiments/_build/adam_stream_bench [n_params]
// The arithmetic is ofa::adam_kernel's (csrc/loss_optim.hip); only the memory schedule varies.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  auto r = [](float f) { uint32_t u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
  return r(a) | (r(b) << 16);
}

template <int NT> __device__ __forceinline__ f4v ld4(const float* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return *reinterpret_cast<const f4v*>(p);
}
template <int NT> __device__ __forceinline__ void st4(float* p, f4v v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(p));
  else *reinterpret_cast<f4v*>(p) = v;
}

// Q quads (4 parameters each) per thread and trip, all loads of the trip issued before the first update; NTL / NTS: nontemporal
// loads / stores on the fp32 state (the bf16 model copy is read by the next forward: always a plain store)
template <int Q, int NTL, int NTS>
__global__ __launch_bounds__(256) void adam_var(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                const uint16_t* __restrict__ grad, uint16_t* __restrict__ model, int64_t n,
                                                float gmul, float step_size, float beta1, float beta2, float eps) {
  const int64_t nq = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x; q0 < nq; q0 += stride * Q) {
    f4v p4[Q], m4[Q], v4[Q];
    u2v g2[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int64_t q = q0 + k * stride;
      if (q < nq) {
        const int64_t i = q << 2;
        p4[k] = ld4<NTL>(master + i);
        m4[k] = ld4<NTL>(m + i);
        v4[k] = ld4<NTL>(v + i);
        g2[k] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const u2v*>(grad + i)) : *reinterpret_cast<const u2v*>(grad + i);
      }
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int64_t q = q0 + k * stride;
      if (q < nq) {
        const int64_t i = q << 2;
        float g[4] = {__uint_as_float(g2[k].x << 16), __uint_as_float(g2[k].x & 0xffff0000u), __uint_as_float(g2[k].y << 16),
                      __uint_as_float(g2[k].y & 0xffff0000u)};
        float* pp = reinterpret_cast<float*>(&p4[k]);
        float* mm = reinterpret_cast<float*>(&m4[k]);
        float* vv = reinterpret_cast<float*>(&v4[k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ge = g[e] * gmul;
          mm[e] = mm[e] * beta1 + (1.f - beta1) * ge;
          vv[e] = vv[e] * beta2 + (1.f - beta2) * ge * ge;
          pp[e] -= step_size * mm[e] / (sqrtf(vv[e]) + eps);
        }
        st4<NTS>(m + i, m4[k]);
        st4<NTS>(v + i, v4[k]);
        st4<NTS>(master + i, p4[k]);
        u2v o;
        o.x = pack_bf16(pp[0], pp[1]);
        o.y = pack_bf16(pp[2], pp[3]);
        *reinterpret_cast<u2v*>(model + i) = o;
      }
    }
  }
}

// contiguous chunk per block instead of a grid-stride walk: block b owns quads [b * per, (b + 1) * per)
template <int Q, int NTS>
__global__ __launch_bounds__(256) void adam_chunk(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                  const uint16_t* __restrict__ grad, uint16_t* __restrict__ model, int64_t n,
                                                  float gmul, float step_size, float beta1, float beta2, float eps) {
  const int64_t nq = n >> 2;
  const int64_t per = (nq + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < nq ? lo + per : nq;
  for (int64_t q0 = lo + threadIdx.x; q0 < hi; q0 += 256 * Q) {
    f4v p4[Q], m4[Q], v4[Q];
    u2v g2[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int64_t q = q0 + k * 256;
      if (q < hi) {
        const int64_t i = q << 2;
        p4[k] = *reinterpret_cast<const f4v*>(master + i);
        m4[k] = *reinterpret_cast<const f4v*>(m + i);
        v4[k] = *reinterpret_cast<const f4v*>(v + i);
        g2[k] = *reinterpret_cast<const u2v*>(grad + i);
      }
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int64_t q = q0 + k * 256;
      if (q < hi) {
        const int64_t i = q << 2;
        float g[4] = {__uint_as_float(g2[k].x << 16), __uint_as_float(g2[k].x & 0xffff0000u), __uint_as_float(g2[k].y << 16),
                      __uint_as_float(g2[k].y & 0xffff0000u)};
        float* pp = reinterpret_cast<float*>(&p4[k]);
        float* mm = reinterpret_cast<float*>(&m4[k]);
        float* vv = reinterpret_cast<float*>(&v4[k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ge = g[e] * gmul;
          mm[e] = mm[e] * beta1 + (1.f - beta1) * ge;
          vv[e] = vv[e] * beta2 + (1.f - beta2) * ge * ge;
          pp[e] -= step_size * mm[e] / (sqrtf(vv[e]) + eps);
        }
        st4<NTS>(m + i, m4[k]);
        st4<NTS>(v + i, v4[k]);
        st4<NTS>(master + i, p4[k]);
        u2v o;
        o.x = pack_bf16(pp[0], pp[1]);
        o.y = pack_bf16(pp[2], pp[3]);
        *reinterpret_cast<u2v*>(model + i) = o;
      }
    }
  }
}

// reference points: a float4 copy of the same byte count (14 B read + 14 B written per parameter)
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, int64_t nv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) b[i] = a[i];
}

template <typename F> static float time_us(F launch, int reps = 10) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 141600000;
  float *master, *m, *v;
  uint16_t *grad, *model;
  CK(hipMalloc(&master, n * 4));
  CK(hipMalloc(&m, n * 4));
  CK(hipMalloc(&v, n * 4));
  CK(hipMalloc(&grad, n * 2));
  CK(hipMalloc(&model, n * 2));
  CK(hipMemset(master, 0, n * 4));
  CK(hipMemset(m, 0, n * 4));
  CK(hipMemset(v, 0, n * 4));
  CK(hipMemset(grad, 0, n * 2));
  CK(hipMemset(model, 0, n * 2));
  const double bytes = 28.0 * (double)n;
  printf("# Adam stream microbenchmark, n = %lld parameters, %.2f GB moved per pass (28 B / parameter)\n", (long long)n, bytes / 1e9);
  {
    const int64_t nv = (int64_t)(14.0 * n / 16.0);
    float4 *a, *b;
    CK(hipMalloc(&a, nv * 16));
    CK(hipMalloc(&b, nv * 16));
    CK(hipMemset(a, 0, nv * 16));
    for (int nb : {2048, 8192, 32768}) {
      const float us = time_us([&] { hipLaunchKernelGGL(copy_kernel, dim3(nb), dim3(256), 0, 0, a, b, nv); });
      printf("copy  same bytes            grid %6d : %8.1f us  %6.2f TB/s\n", nb, us, bytes / us / 1e6);
    }
    CK(hipFree(a));
    CK(hipFree(b));
  }
#define RUN(NAME, KERN)                                                                                             \
  for (int nb : {1024, 2048, 4096, 8192, 16384, 65536}) {                                                          \
    const float us = time_us([&] {                                                                                  \
      hipLaunchKernelGGL(KERN, dim3(nb), dim3(256), 0, 0, master, m, v, grad, model, n, 1.0f, 1e-4f, 0.9f, 0.999f, 1e-8f); \
    });                                                                                                             \
    printf("%-26s grid %6d : %8.1f us  %6.2f TB/s\n", NAME, nb, us, bytes / us / 1e6);                              \
  }
  RUN("stride Q=1 (shipped)", (adam_var<1, 0, 0>));
  RUN("stride Q=2", (adam_var<2, 0, 0>));
  RUN("stride Q=4", (adam_var<4, 0, 0>));
  RUN("stride Q=2 nt-store", (adam_var<2, 0, 1>));
  RUN("stride Q=2 nt-load+store", (adam_var<2, 1, 1>));
  RUN("stride Q=1 nt-store", (adam_var<1, 0, 1>));
  RUN("chunk  Q=2", (adam_chunk<2, 0>));
  RUN("chunk  Q=4", (adam_chunk<4, 0>));
  RUN("chunk  Q=2 nt-store", (adam_chunk<2, 1>));
  CK(hipDeviceSynchronize());
  return 0;
}
