// Design-space microbenchmark for the CDS SpMV (7 bands, 3-D grid): not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "../setintersectionprojection.jl_amd/csrc/kernels_cds.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct alignas(16) F4 { float v[4]; };
struct Offs { long long o[7]; };

__device__ __forceinline__ F4 ld(const float* p) { return *reinterpret_cast<const F4*>(p); }
__device__ __forceinline__ F4 ldnt(const float* p) {
  typedef float vf4 __attribute__((ext_vector_type(4)));
  F4 r;
  vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
  r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  return r;
}
__device__ __forceinline__ void st(float* p, const F4& x) { *reinterpret_cast<F4*>(p) = x; }
__device__ __forceinline__ void stnt(float* p, const F4& x) {
  typedef float vf4 __attribute__((ext_vector_type(4)));
  vf4 t = {x.v[0], x.v[1], x.v[2], x.v[3]};
  __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p));
}

template <int NT, int REMAP>
__global__ __launch_bounds__(256) void k_spmv(long long N, long long ldq, const float* __restrict__ R, Offs a,
                                              const float* __restrict__ x, float* __restrict__ y, int gridstride) {
  const long long nvec = N / 4;
  long long nb = gridDim.x;
  long long bid = blockIdx.x;
  if (REMAP) {   // XCD-aware: blocks b, b+8, ... share an XCD -> give each XCD a contiguous chunk of the sweep
    const long long per = nb / 8;
    bid = (bid % 8) * per + bid / 8;
  }
  for (long long vi = bid * 256 + threadIdx.x; vi < nvec; vi += nb * 256) {
    const long long r = vi * 4;
    float acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const long long o = a.o[b];
      const F4 rv = NT ? ldnt(R + b * ldq + r) : ld(R + b * ldq + r);
      const long long c = r + o;
      if ((o % 4) == 0 && c >= 0 && c + 4 <= N) {
        const F4 xv = ld(x + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = acc[k] + rv.v[k] * xv.v[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const long long cc = c + k;
          if (cc >= 0 && cc < N) acc[k] = acc[k] + rv.v[k] * x[cc];
        }
      }
    }
    F4 o4;
#pragma unroll
    for (int k = 0; k < 4; ++k) o4.v[k] = acc[k];
    if (NT) stnt(y + r, o4); else st(y + r, o4);
    if (!gridstride) break;
  }
}

// branch-free variant: x carries a zero halo of max|off| on both sides, every band is one unconditional
// (possibly 4-byte-aligned) dwordx4 load of x; out-of-range rows are masked by a select.
typedef float vf4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float vf4a __attribute__((ext_vector_type(4)));
template <int NTS>
__global__ __launch_bounds__(256) void k_spmv_halo(long long N, long long ldq, const float* __restrict__ R, Offs a,
                                                   const float* __restrict__ x, float* __restrict__ y) {
  const long long nvec = N / 4;
  for (long long vi = (long long)blockIdx.x * 256 + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * 256) {
    const long long r = vi * 4;
    vf4a rv[7]; vf4u xv[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      rv[b] = __builtin_nontemporal_load(reinterpret_cast<const vf4a*>(R + b * ldq + r));
      xv[b] = *reinterpret_cast<const vf4u*>(x + r + a.o[b]);
    }
    float acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const long long c = r + a.o[b];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float t = acc[k] + rv[b][k] * xv[b][k];
        acc[k] = (c + k >= 0 && c + k < N) ? t : acc[k];
      }
    }
    vf4a o4 = {acc[0], acc[1], acc[2], acc[3]};
    if (NTS) __builtin_nontemporal_store(o4, reinterpret_cast<vf4a*>(y + r));
    else *reinterpret_cast<vf4a*>(y + r) = o4;
  }
}

// pure streaming reference: y = sum of 7 bands * x (no shifted reads)
__global__ __launch_bounds__(256) void k_stream(long long N, long long ldq, const float* __restrict__ R,
                                                const float* __restrict__ x, float* __restrict__ y) {
  const long long nvec = N / 4;
  for (long long vi = (long long)blockIdx.x * 256 + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * 256) {
    const long long r = vi * 4;
    const F4 xv = ld(x + r);
    float acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const F4 rv = ld(R + b * ldq + r);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = acc[k] + rv.v[k] * xv.v[k];
    }
    F4 o4;
#pragma unroll
    for (int k = 0; k < 4; ++k) o4.v[k] = acc[k];
    st(y + r, o4);
  }
}

__global__ void k_copy(long long n4, const float4* __restrict__ a, float4* __restrict__ b) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) b[i] = a[i];
}

template <typename F>
double timeit(F f, int reps = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 256;
  const long long N = (long long)n * n * n;
  Offs a; long long offs[7] = {0, -1, 1, -n, n, -(long long)n * n, (long long)n * n};
  for (int i = 0; i < 7; ++i) a.o[i] = offs[i];
  const double bytes = 9.0 * N * 4;
  for (long long pad : {0ll}) {
    const long long ldq = N + pad;
    float *R, *x, *y;
    CK(hipMalloc(&R, ldq * 7 * 4)); CK(hipMalloc(&x, N * 4)); CK(hipMalloc(&y, N * 4));
    {   // random (non-zero) data: zero operands inflate clocks and bandwidth
      std::vector<float> h((size_t)ldq * 7);
      unsigned s = 12345u;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
      CK(hipMemcpy(R, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(x, h.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    }
    printf("n=%d pad=%lld\n", n, pad);
    // copy 3N floats: source R[0,3N) -> destination R[4*ldq, 4*ldq+3N)  (inside the 7*ldq allocation)
    double t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, 3 * N / 4, (const float4*)R, (float4*)(R + 4 * ldq)); });
    printf("  copy 3N floats (r+w)       : %.1f us  %.0f GB/s\n", t * 1e3, 2.0 * 3 * N * 4 / t / 1e6); fflush(stdout);
    for (int nb : {1024, 2048, 4096}) {
      t = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(nb), dim3(256), 0, 0, N, ldq, R, x, y); });
      printf("  stream nb=%d              : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
    }
    for (int nb : {1024, 2048, 4096, 8192}) {
      t = timeit([&] { hipLaunchKernelGGL((k_spmv<0, 0>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, x, y, 1); });
      printf("  spmv gs nb=%d             : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
      t = timeit([&] { hipLaunchKernelGGL((k_spmv<0, 1>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, x, y, 1); });
      printf("  spmv gs nb=%d remap       : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
      t = timeit([&] { hipLaunchKernelGGL((k_spmv<1, 0>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, x, y, 1); });
      printf("  spmv gs nb=%d nt          : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
      t = timeit([&] { hipLaunchKernelGGL((k_spmv<1, 1>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, x, y, 1); });
      printf("  spmv gs nb=%d nt remap    : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
    }
    {
      const long long H = (long long)n * n;      // halo = max |offset|
      float* xh; CK(hipMalloc(&xh, (N + 2 * H) * 4)); CK(hipMemset(xh, 0, (N + 2 * H) * 4));
      CK(hipMemcpy(xh + H, x, N * 4, hipMemcpyDeviceToDevice));
      for (int nb : {1024, 2048, 4096}) {
        t = timeit([&] { hipLaunchKernelGGL((k_spmv_halo<1>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, xh + H, y); });
        printf("  spmv HALO nt nb=%d         : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
        t = timeit([&] { hipLaunchKernelGGL((k_spmv_halo<0>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, xh + H, y); });
        printf("  spmv HALO nt-load nb=%d    : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
      }
      CK(hipFree(xh));
    }
    {
      sipx::CdsArgs ca; ca.d = 7; for (int i = 0; i < 7; ++i) ca.off[i] = a.o[i];
      sipx::Grid gg; gg.N = N;
      t = timeit([&] { sipx::K<float>::spmv(0, gg, N, R, ca, x, y); });
      printf("  ENGINE k_cds MODE0         : %.1f us  %.0f GB/s\n", t * 1e3, bytes / t / 1e6); fflush(stdout);
    }
    {
      const int nb = (int)(N / 4 / 256);
      t = timeit([&] { hipLaunchKernelGGL((k_spmv<0, 0>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, x, y, 0); });
      printf("  spmv one-shot nb=%d       : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
      t = timeit([&] { hipLaunchKernelGGL((k_spmv<0, 1>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, x, y, 0); });
      printf("  spmv one-shot remap        : %.1f us  %.0f GB/s\n", t * 1e3, bytes / t / 1e6); fflush(stdout);
      t = timeit([&] { hipLaunchKernelGGL((k_spmv<1, 0>), dim3(nb), dim3(256), 0, 0, N, ldq, R, a, x, y, 0); });
      printf("  spmv one-shot nt           : %.1f us  %.0f GB/s\n", t * 1e3, bytes / t / 1e6); fflush(stdout);
    }
    CK(hipFree(R)); CK(hipFree(x)); CK(hipFree(y));
  }
  return 0;
}
