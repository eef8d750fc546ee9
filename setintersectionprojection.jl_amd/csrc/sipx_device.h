// Device-side helpers shared by the kernel translation units (gfx950, wave64).
#pragma once
#include "sipx_common.h"

namespace sipx {

template <typename T, int V>
struct alignas(sizeof(T) * V) Vec {
  T v[V];
};

template <typename T, int V>
__device__ __forceinline__ Vec<T, V> ldv(const T* p) {
  return *reinterpret_cast<const Vec<T, V>*>(p);
}
template <typename T, int V>
__device__ __forceinline__ void stv(T* p, const Vec<T, V>& x) {
  *reinterpret_cast<Vec<T, V>*>(p) = x;
}
// Non-temporal (streaming) vector load: for data read exactly once per pass, e.g. the CDS bands --
// keeps them from evicting the re-used vectors out of L2 / Infinity Cache (measured +8-10% on cds_spmv).
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> ldv_nt(const T* p) {
  typedef T vt __attribute__((ext_vector_type(V)));
  Vec<T, V> r;
  if constexpr (V == 1) {
    r.v[0] = __builtin_nontemporal_load(p);
  } else if constexpr (sizeof(T) * V == 16) {
    const vt t = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
#pragma unroll
    for (int k = 0; k < V; ++k) r.v[k] = t[k];
  } else {   // 32-byte vectors: two 16-byte halves
    typedef T vh __attribute__((ext_vector_type(V / 2)));
    const vh a = __builtin_nontemporal_load(reinterpret_cast<const vh*>(p));
    const vh b = __builtin_nontemporal_load(reinterpret_cast<const vh*>(p) + 1);
#pragma unroll
    for (int k = 0; k < V / 2; ++k) {
      r.v[k] = a[k];
      r.v[V / 2 + k] = b[k];
    }
  }
  return r;
}
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> zerov() {
  Vec<T, V> z;
#pragma unroll
  for (int k = 0; k < V; ++k) z.v[k] = T(0);
  return z;
}

// 64-lane butterfly-free reduction (fixed order => deterministic).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_down(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}

// Block partials: partials[(slot0+k)*NB + blockIdx.x] = sum over the block of acc[k].
template <int K>
__device__ __forceinline__ void block_reduce_store(double (&acc)[K], double* __restrict__ partials, int slot0) {
  __shared__ double sm[K][BLOCK / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double v = wave_sum(acc[k]);
    if (lane == 0) sm[k][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) s += sm[threadIdx.x][i];
    partials[(long long)(slot0 + threadIdx.x) * NB + blockIdx.x] = s;
  }
}

// Sum of the NB partials of one slot by one 256-thread block (fixed order).
__device__ __forceinline__ double block_sum_partials(const double* __restrict__ p) {
  __shared__ double sm[BLOCK / 64];
  double v = 0;
  for (int i = threadIdx.x; i < NB; i += BLOCK) v += p[i];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int i = 0; i < BLOCK / 64; ++i) s += sm[i];
  return s;
}

// Per-block max of a non-negative value -> maxpart[blockIdx.x].
template <typename T>
__device__ __forceinline__ void block_max_store(T vmax, T* __restrict__ maxpart) {
  __shared__ T smax[BLOCK / 64];
  vmax = wave_max<T>(vmax);
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = vmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    T m = smax[0];
    for (int i = 1; i < BLOCK / 64; ++i) m = smax[i] > m ? smax[i] : m;
    maxpart[blockIdx.x] = m;
  }
}

// One magnitude into the first-pass reductions of the two-pass projectors:
// slot 0 ||v||_1, 1 ||v||_2^2, 2 nnz, 3.. S_k = sum(|v| > t_k), 3+L1_K.. C_k = count(|v| > t_k).
template <typename T>
__device__ __forceinline__ void probe_acc(T av, T x, const double (&t)[L1_K], double (&acc)[PREP_SLOTS]) {
  const double a = (double)av;
  acc[0] += a;
  acc[1] += (double)x * (double)x;
  acc[2] += av > T(0) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < L1_K; ++k) {
    const bool on = a > t[k];
    acc[3 + k] += on ? a : 0.0;
    acc[3 + L1_K + k] += on ? 1.0 : 0.0;
  }
}

template <typename T>
__device__ __forceinline__ T eps_of();
template <>
__device__ __forceinline__ float eps_of<float>() { return 1.1920928955078125e-07f; }
template <>
__device__ __forceinline__ double eps_of<double>() { return 2.220446049250313e-16; }

// Julia max/min: NaN-propagating.
__device__ __forceinline__ double jl_max(double a, double b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }
__device__ __forceinline__ double jl_min(double a, double b) { return (a != a || b != b) ? (a + b) : (a < b ? a : b); }

struct Coord {
  int i, j, k;
};
// Coordinates of linear index g (column-major, dim 1 fastest).
__device__ __forceinline__ Coord coords(const Grid& G, long long g) {
  Coord c;
  if (G.N < (1ll << 31)) {
    unsigned u = (unsigned)g, n1 = (unsigned)G.n[0], n2 = (unsigned)G.n[1];
    unsigned jk = u / n1;
    c.i = (int)(u - jk * n1);
    unsigned k = jk / n2;
    c.j = (int)(jk - k * n2);
    c.k = (int)k;
  } else {
    long long jk = g / G.n[0];
    c.i = (int)(g - jk * G.n[0]);
    long long k = jk / G.n[1];
    c.j = (int)(jk - k * G.n[1]);
    c.k = (int)k;
  }
  return c;
}
__device__ __forceinline__ int coord_of(const Coord& c, int dir) { return dir == 0 ? c.i : (dir == 1 ? c.j : c.k); }

// Forward difference along `dir` at the V consecutive points g..g+V-1 (same line):
// s = (-ih)*x[g] + ih*x[g+stride], the two products of a CSC row in column order
// (reference get_discrete_Grad.jl:22-23,58-60 + SparseArrays mul!).  Points on the last
// hyper-plane along `dir` are pads of the padded layout: valid=false, s=0.
template <typename T, int V>
__device__ __forceinline__ void fwd_dir(const Grid& G, const T* __restrict__ x, const Vec<T, V>& xc, long long g,
                                        const Coord& c, int dir, T ih, T (&s)[V], bool (&valid)[V]) {
  const T nih = -ih;
  if (dir == 0) {
    T xn[V];
#pragma unroll
    for (int k = 0; k < V - 1; ++k) xn[k] = xc.v[k + 1];
    xn[V - 1] = (c.i + V < G.n[0]) ? x[g + V] : T(0);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      valid[k] = (c.i + k < G.n[0] - 1);
      s[k] = valid[k] ? (nih * xc.v[k] + ih * xn[k]) : T(0);
    }
  } else {
    const bool ok = coord_of(c, dir) < G.n[dir] - 1;
    Vec<T, V> xn = ok ? ldv<T, V>(x + g + G.st[dir]) : zerov<T, V>();
#pragma unroll
    for (int k = 0; k < V; ++k) {
      valid[k] = ok;
      s[k] = ok ? (nih * xc.v[k] + ih * xn.v[k]) : T(0);
    }
  }
}

// Adjoint of the forward difference along `dir`, accumulated into t[] at grid points g..g+V-1:
// t += ih*w[g-stride] (if that row exists) ; t += (-ih)*w[g] (if row g exists) -- a CSC column
// of D in ascending row order (SparseArrays mul!(tmp, A', v)).  W(e) loads the V values of w
// at padded index e; W1(e) loads one.
template <typename T, int V, typename WV, typename W1>
__device__ __forceinline__ void adj_dir_acc(const Grid& G, long long g, const Coord& c, int dir, T ih, T (&t)[V],
                                            WV wv, W1 w1) {
  const T nih = -ih;
  Vec<T, V> wc = wv(g);
  if (dir == 0) {
    T wp[V];
    wp[0] = (c.i > 0) ? w1(g - 1) : T(0);
#pragma unroll
    for (int k = 1; k < V; ++k) wp[k] = wc.v[k - 1];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      if (c.i + k > 0) t[k] = t[k] + ih * wp[k];
      if (c.i + k < G.n[0] - 1) t[k] = t[k] + nih * wc.v[k];
    }
  } else {
    const int cc = coord_of(c, dir);
    if (cc > 0) {
      Vec<T, V> wp = wv(g - G.st[dir]);
#pragma unroll
      for (int k = 0; k < V; ++k) t[k] = t[k] + ih * wp.v[k];
    }
    if (cc < G.n[dir] - 1) {
#pragma unroll
      for (int k = 0; k < V; ++k) t[k] = t[k] + nih * wc.v[k];
    }
  }
}

template <typename T>
__device__ __forceinline__ T soft_thr(T v, T th) {
  // sign(v) * max(abs(v) - th, 0)   (project_l1_Duchi!.jl:49, prox_l1!.jl:9)
  T t = fabs(v) - th;
  t = t > T(0) ? t : T(0);
  return v > T(0) ? t : (v < T(0) ? -t : v);
}

template <typename T>
struct ProxCtx {
  int prox;
  T plo, phi, rho, theta, scale, tau;
  int fill;
  long long idx_cut;
};

template <typename T>
__device__ __forceinline__ ProxCtx<T> make_prox(int prox, T plo, T phi, T rho, const ProjScalars<T>* ps) {
  ProxCtx<T> c;
  c.prox = prox;
  c.plo = plo;
  c.phi = phi;
  c.rho = rho;
  c.theta = T(0);
  c.scale = T(1);
  c.tau = T(0);
  c.fill = 0;
  c.idx_cut = -1;
  if (ps) {
    c.theta = ps->theta;
    c.scale = ps->scale;
    c.fill = ps->fill;
    c.tau = ps->tau;
    c.idx_cut = ps->quota;
  }
  if (prox == PX_PROX_L1) c.theta = T(1) / phi;   // prox_l1!(x, constraint.max): threshold 1/rho
  return c;
}

// One element of prox_i / P_i.  lb/ub: per-element bounds; m: distance-term centre; e: padded index.
template <typename T>
__device__ __forceinline__ T prox_apply(const ProxCtx<T>& c, T v, T lb, T ub, T m, long long e) {
  switch (c.prox) {
    case PX_BOUNDS: {                       // max(LB, min(x, UB))      project_bounds!.jl:9
      T t = v < c.phi ? v : c.phi;
      return c.plo > t ? c.plo : t;
    }
    case PX_BOUNDS_VEC: {                   // project_bounds!.jl:21-22
      T t = v < ub ? v : ub;
      return lb > t ? lb : t;
    }
    case PX_DIST:                           // (x*rho + m) / (rho + 1.0): Float64 division  prox_l2s!.jl:4
      return (T)((double)(v * c.rho + m) / ((double)c.rho + 1.0));
    case PX_L1:
    case PX_PROX_L1:
      return soft_thr(v, c.theta);
    case PX_L2:
    case PX_ANNULUS:                        // rmul!(x, sigma/nl2) or the constant fill  project_annulus!.jl:9-17
      return c.fill ? c.scale : v * c.scale;
    case PX_CARD: {                         // keep the k largest |v|, ties by lowest index
      const T av = fabs(v);
      return (av > c.tau || (av == c.tau && e <= c.idx_cut)) ? v : T(0);
    }
  }
  return v;
}


}  // namespace sipx
