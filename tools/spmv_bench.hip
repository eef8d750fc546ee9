// Design-space microbenchmark for the CDS SpMV (7 bands, 3-D grid): not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <cstring>
#include <type_traits>
#include "../setintersectionprojection.jl_amd/csrc/kernels_cds.hip"

namespace sipx {      // (defined in engine.cpp for the library; the stand-alone tool records nothing)
const LaunchObserver*& launch_observer() {
  static thread_local const LaunchObserver* obs = nullptr;
  return obs;
}
}  // namespace sipx

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

// ---- z-marching variant of the product kernel (round 3; what north_star calls "LDS-staged diagonal bands") ----------------
// Q symmetric bit for bit: only the bands 0, +1, +n1, +n1n2 are stored / read (as the engine's k_cds does).  A workgroup owns
// a tile of 4 LX x TY grid points of a plane and walks z: x of the planes k-1, k, k+1 and the +n1n2 band of plane k-1 stay in
// registers, the +-1 neighbours come from the lanes next door (shuffles), the +-n1 neighbours and the +n1 band of the row
// above go through LDS (one barrier per plane) -- so every band value and every x crosses the fabric ONCE, plus the rows
// in front of / behind a tile (2 / TY of x, 1 / TY of one band) and one plane per chunk.  Products are added in the engine's
// Q_offsets order 0, -1, +1, -n1, +n1, -n1n2, +n1n2 with its masks (column inside [0, N)), so y is bit-identical to k_cds.
template <int NTH>
__global__ __launch_bounds__(NTH) void k_spmv_march(long long n1, long long n2, long long n3, long long ldq, const float* __restrict__ R,
                                                    const float* __restrict__ x, float* __restrict__ y, int lgLX, int tiles_x, int tiles_y,
                                                    int zchunk, long long items) {
  __shared__ float sx[2][4][NTH], sr[2][4][NTH];
  const int tid = threadIdx.x, LX = 1 << lgLX, tx = tid & (LX - 1), ty = tid >> lgLX, TY = NTH >> lgLX;
  const long long st1 = n1, st2 = n1 * n2, N = st2 * n3;
  const float* __restrict__ R0 = R;
  const float* __restrict__ R1 = R + ldq;
  const float* __restrict__ R2 = R + 2 * ldq;
  const float* __restrict__ R3 = R + 3 * ldq;
  const long long tiles = (long long)tiles_x * tiles_y;
  for (long long item = blockIdx.x; item < items; item += gridDim.x) {
    const long long zc = item / tiles, tile = item - zc * tiles;
    const int tile_y = (int)(tile / tiles_x), tile_x = (int)(tile - (long long)tile_y * tiles_x);
    const long long i0 = ((long long)tile_x * LX + tx) * 4, j = (long long)tile_y * TY + ty;
    const bool active = i0 < n1 && j < n2;
    const long long k0 = zc * zchunk, k1 = (k0 + zchunk < n3) ? k0 + zchunk : n3;
    const unsigned go = active ? (unsigned)(i0 + st1 * j) : 0u;
    __syncthreads();
    vf4a xm = {0, 0, 0, 0}, x0 = {0, 0, 0, 0}, rzm = {0, 0, 0, 0};
    if (active) {
      xm = *reinterpret_cast<const vf4a*>(x + st2 * (k0 - 1) + go);            // (x carries a halo of a plane on both sides)
      x0 = *reinterpret_cast<const vf4a*>(x + st2 * k0 + go);
      if (k0 > 0) rzm = *reinterpret_cast<const vf4a*>(R3 + st2 * (k0 - 1) + go);
    }
    for (long long kz = k0; kz < k1; ++kz) {
      const int par = (int)(kz & 1);
      const long long pz = st2 * kz;
      vf4a xp = {0, 0, 0, 0}, r0 = xp, r1 = xp, r2 = xp, r3 = xp;
      if (active) {
        xp = *reinterpret_cast<const vf4a*>(x + pz + st2 + go);
        r0 = __builtin_nontemporal_load(reinterpret_cast<const vf4a*>(R0 + pz + go));
        r1 = __builtin_nontemporal_load(reinterpret_cast<const vf4a*>(R1 + pz + go));
        r2 = __builtin_nontemporal_load(reinterpret_cast<const vf4a*>(R2 + pz + go));
        r3 = __builtin_nontemporal_load(reinterpret_cast<const vf4a*>(R3 + pz + go));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { sx[par][k][tid] = x0[k]; sr[par][k][tid] = r2[k]; }
      // the points next door along x: lanes, or (tile edge / wave edge) one element from memory
      float xl = __shfl_up(x0[3], 1, 64), xr = __shfl_down(x0[0], 1, 64), rl = __shfl_up(r1[3], 1, 64);
      const long long r = pz + go;               // row of element 0
      if (active) {
        if (tx == 0 || (tid & 63) == 0) { xl = x[r - 1]; rl = r > 0 ? R1[r - 1] : 0.f; }
        if (tx == LX - 1 || (tid & 63) == 63) xr = x[r + 4];
      }
      __syncthreads();
      vf4a xu = {0, 0, 0, 0}, xd = xu, ru = xu;    // x of the row above (j - 1) / below (j + 1), +n1 band of the row above
      if (active) {
        if (ty > 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { xu[k] = sx[par][k][tid - LX]; ru[k] = sr[par][k][tid - LX]; }
        } else {
          xu = *reinterpret_cast<const vf4a*>(x + r - st1);
          if (r - st1 >= 0) ru = *reinterpret_cast<const vf4a*>(R2 + r - st1);
        }
        if (ty < TY - 1 && j + 1 < n2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) xd[k] = sx[par][k][tid + LX];
        } else {
          xd = *reinterpret_cast<const vf4a*>(x + r + st1);
        }
        vf4a o4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const long long rr = r + k;
          const float xm1 = k == 0 ? xl : x0[k - 1], xp1 = k == 3 ? xr : x0[k + 1];
          const float rm1 = k == 0 ? rl : r1[k - 1];
          float acc = 0.f;
          acc = acc + r0[k] * x0[k];                                             // offset 0
          { const float t = acc + rm1 * xm1; acc = (rr - 1 >= 0) ? t : acc; }    // -1: the +1 band one row back
          { const float t = acc + r1[k] * xp1; acc = (rr + 1 < N) ? t : acc; }   // +1
          { const float t = acc + ru[k] * xu[k]; acc = (rr - st1 >= 0) ? t : acc; }      // -n1
          { const float t = acc + r2[k] * xd[k]; acc = (rr + st1 < N) ? t : acc; }       // +n1
          { const float t = acc + rzm[k] * xm[k]; acc = (rr - st2 >= 0) ? t : acc; }     // -n1n2
          { const float t = acc + r3[k] * xp[k]; acc = (rr + st2 < N) ? t : acc; }       // +n1n2
          o4[k] = acc;
        }
        __builtin_nontemporal_store(o4, reinterpret_cast<vf4a*>(y + r));
      }
      xm = x0; x0 = xp; rzm = r3;
    }
  }
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
    if (!getenv("SPMV_QUICK"))
    for (int nb : {1024, 2048, 4096}) {
      t = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(nb), dim3(256), 0, 0, N, ldq, R, x, y); });
      printf("  stream nb=%d              : %.1f us  %.0f GB/s\n", nb, t * 1e3, bytes / t / 1e6); fflush(stdout);
    }
    if (!getenv("SPMV_QUICK"))
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
      // ---- the product kernel as the engine runs it (symmetric-partner read of 4 stored bands) against the z-march ----
      const long long H = (long long)n * n;
      float *xh, *y2;
      CK(hipMalloc(&xh, (N + 2 * H) * 4)); CK(hipMemset(xh, 0, (N + 2 * H) * 4)); CK(hipMalloc(&y2, N * 4));
      CK(hipMemcpy(xh + H, x, N * 4, hipMemcpyDeviceToDevice));
      // bands in the engine's order 0, -1, +1, -n, +n, -n^2, +n^2; make Q symmetric bit for bit: band(-o)[r] = band(+o)[r - o]
      std::vector<float> hq((size_t)ldq * 7);
      CK(hipMemcpy(hq.data(), R, hq.size() * 4, hipMemcpyDeviceToHost));
      const long long so[3] = {1, n, (long long)n * n};
      for (int q = 0; q < 3; ++q)
        for (long long r = 0; r < N; ++r) {
          // structurally zero couplings as in A'A of the difference operators: no coupling across the end of a line / plane
          const bool edge = q == 0 ? (r % n == n - 1) : (q == 1 ? ((r / n) % n == n - 1) : (r / ((long long)n * n) == n - 1));
          if (edge) hq[(size_t)(2 + 2 * q) * ldq + r] = 0.f;
        }
      for (int q = 0; q < 3; ++q)
        for (long long r = 0; r < N; ++r) hq[(size_t)(1 + 2 * q) * ldq + r] = r - so[q] >= 0 ? hq[(size_t)(2 + 2 * q) * ldq + r - so[q]] : 0.f;
      CK(hipMemcpy(R, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
      sipx::CdsArgs ca; ca.d = 7; for (int i = 0; i < 7; ++i) ca.off[i] = a.o[i];
      ca.sym = 1;
      for (int i = 0; i < 7; ++i) ca.partner[i] = i;
      ca.partner[1] = 2; ca.partner[3] = 4; ca.partner[5] = 6;
      sipx::Grid gg; gg.N = N;
      const double moved = 6.0 * N * 4;
      t = timeit([&] { sipx::K<float>::spmv(0, gg, N, R, ca, xh + H, y); });
      printf("  ENGINE k_cds sym (4 bands)  : %.1f us  %.0f GB/s by (d+2)Nw, %.0f GB/s by the 6 N w that must move\n", t * 1e3, bytes / t / 1e6, moved / t / 1e6);
      // the same four bands packed for the march kernel: 0, +1, +n, +n^2
      float* R4; CK(hipMalloc(&R4, (size_t)ldq * 4 * 4));
      const int src[4] = {0, 2, 4, 6};
      for (int q = 0; q < 4; ++q) CK(hipMemcpy(R4 + (size_t)q * ldq, R + (size_t)src[q] * ldq, (size_t)N * 4, hipMemcpyDeviceToDevice));
      auto march = [&](auto nth, int zchunk_planes) {
        constexpr int NTH = decltype(nth)::value;
        const long long nvx = n / 4;
        int lg = 0; while ((1 << lg) < nvx && lg < 6) ++lg;
        const int LX = 1 << lg, TY = NTH / LX;
        const int tiles_x = (int)((nvx + LX - 1) / LX), tiles_y = (n + TY - 1) / TY;
        const long long tiles = (long long)tiles_x * tiles_y, nch = (n + zchunk_planes - 1) / zchunk_planes, items = tiles * nch;
        const int grid = (int)(items < 2048 ? items : 2048);
        hipLaunchKernelGGL((k_spmv_march<NTH>), dim3(grid), dim3(NTH), 0, 0, (long long)n, (long long)n, (long long)n, ldq, R4, xh + H, y2, lg,
                           tiles_x, tiles_y, zchunk_planes, items);
      };
      std::vector<float> ya(N), yb(N);
      CK(hipMemcpy(ya.data(), y, N * 4, hipMemcpyDeviceToHost));
      for (int zc : {16, 32, 64, 128}) {
        t = timeit([&] { march(std::integral_constant<int, 256>{}, zc); });
        printf("  MARCH 256 thr, %3d planes   : %.1f us  %.0f GB/s by the 6 N w that must move\n", zc, t * 1e3, moved / t / 1e6);
        t = timeit([&] { march(std::integral_constant<int, 512>{}, zc); });
        printf("  MARCH 512 thr, %3d planes   : %.1f us  %.0f GB/s\n", zc, t * 1e3, moved / t / 1e6);
        t = timeit([&] { march(std::integral_constant<int, 1024>{}, zc); });
        printf("  MARCH 1024 thr, %3d planes  : %.1f us  %.0f GB/s\n", zc, t * 1e3, moved / t / 1e6);
        fflush(stdout);
      }
      march(std::integral_constant<int, 512>{}, 32);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(yb.data(), y2, N * 4, hipMemcpyDeviceToHost));
      long long bad = 0;
      for (long long r = 0; r < N; ++r) bad += memcmp(&ya[r], &yb[r], 4) != 0;
      printf("  MARCH vs ENGINE k_cds sym: %lld of %lld entries differ (bit comparison)\n", bad, N);
      if (getenv("SPMV_ONLY")) {      // one kernel only, for the PMC passes: SPMV_ONLY=engine | march
        const bool eng = std::string(getenv("SPMV_ONLY")) == "engine";
        for (int i = 0; i < 20; ++i) { if (eng) sipx::K<float>::spmv(0, gg, N, R, ca, xh + H, y); else march(std::integral_constant<int, 512>{}, 32); }
        CK(hipDeviceSynchronize());
      }
      CK(hipFree(xh)); CK(hipFree(y2)); CK(hipFree(R4));
    }
    if (!getenv("SPMV_QUICK")) {
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
