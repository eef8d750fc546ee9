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
// Vector load that only needs element alignment (dwordx4 at any 4-byte boundary: gfx950 global loads are
// unaligned-capable): neighbours of a CDS band / stencil are one unconditional load, no index arithmetic.
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> ldv_u(const T* p) {
  Vec<T, V> r;
  if constexpr (V == 1) {
    r.v[0] = *p;
  } else {
    typedef T vu __attribute__((ext_vector_type(V), aligned(sizeof(T))));
    const vu t = *reinterpret_cast<const vu*>(p);
#pragma unroll
    for (int k = 0; k < V; ++k) r.v[k] = t[k];
  }
  return r;
}
template <typename T, int V>
__device__ __forceinline__ void stv_nt(T* p, const Vec<T, V>& x) {
  if constexpr (V == 1) {
    __builtin_nontemporal_store(x.v[0], p);
  } else if constexpr (sizeof(T) * V == 16) {
    typedef T vt __attribute__((ext_vector_type(V)));
    vt t;
#pragma unroll
    for (int k = 0; k < V; ++k) t[k] = x.v[k];
    __builtin_nontemporal_store(t, reinterpret_cast<vt*>(p));
  } else {
    typedef T vh __attribute__((ext_vector_type(V / 2)));
    vh a, b;
#pragma unroll
    for (int k = 0; k < V / 2; ++k) {
      a[k] = x.v[k];
      b[k] = x.v[V / 2 + k];
    }
    __builtin_nontemporal_store(a, reinterpret_cast<vh*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<vh*>(p) + 1);
  }
}
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> zerov() {
  Vec<T, V> z;
#pragma unroll
  for (int k = 0; k < V; ++k) z.v[k] = T(0);
  return z;
}

// One DPP hop of a float64 (two dwords); lanes without a source receive +0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_hop(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Sum over the 64 lanes in a fixed order (deterministic), returned to every lane.  DPP moves inside the vector ALU
// (quad swaps, row rotations, then the row totals are chained through lanes 15/31/47 into lane 63) instead of six
// ds_bpermute round trips with their index arithmetic: the epilogue of a 13- or 19-slot kernel is a third as long.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_hop<0xb1, 0xf>(v);      // quad_perm:[1,0,3,2]
  v += dpp_hop<0x4e, 0xf>(v);      // quad_perm:[2,3,0,1]
  v += dpp_hop<0x124, 0xf>(v);     // row_ror:4
  v += dpp_hop<0x128, 0xf>(v);     // row_ror:8   -> every lane holds the total of its row of 16
  v += dpp_hop<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
  v += dpp_hop<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
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
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) sm[k][w] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) s += sm[threadIdx.x][i];
    double* row = partials + (long long)(slot0 + threadIdx.x) * NB;
    row[blockIdx.x] = s;
    // the consumers add all NB entries of a slot: clear the ones no workgroup of this launch owns, so kernels with
    // different grids may write the same slot at different times
    for (int j = blockIdx.x + gridDim.x; j < NB; j += gridDim.x) row[j] = 0.0;
  }
}

// The same for K slots that are not neighbours: slot numbers in `slots`.
template <int K>
__device__ __forceinline__ void block_reduce_store_at(double (&acc)[K], double* __restrict__ partials, const int (&slots)[K]) {
  __shared__ double sm[K][BLOCK / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) sm[k][w] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) s += sm[threadIdx.x][i];
    int slot = slots[0];
#pragma unroll
    for (int k = 1; k < K; ++k) slot = threadIdx.x == k ? slots[k] : slot;
    double* row = partials + (long long)slot * NB;
    row[blockIdx.x] = s;
    for (int j = blockIdx.x + gridDim.x; j < NB; j += gridDim.x) row[j] = 0.0;
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

// The same sum, in the same order (hence the same bits), by a workgroup of NT >= BLOCK threads: the first BLOCK threads do
// what block_sum_partials does, the others only keep the barriers company.
template <int NT>
__device__ __forceinline__ double block_sum_partials_n(const double* __restrict__ p) {
  static_assert(NT >= BLOCK, "at least BLOCK threads");
  __shared__ double sm[BLOCK / 64];
  double v = 0;
  if (threadIdx.x < BLOCK)
    for (int i = threadIdx.x; i < NB; i += BLOCK) v += p[i];
  v = wave_sum(v);
  __syncthreads();
  if (threadIdx.x < BLOCK && (threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
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
    for (int j = blockIdx.x + gridDim.x; j < NB; j += gridDim.x) maxpart[j] = T(0);
  }
}

// Per-block min of a positive value (INFINITY = none) -> minpart[blockIdx.x].
template <typename T>
__device__ __forceinline__ void block_min_store(T vmin, T* __restrict__ minpart) {
  __shared__ T smin[BLOCK / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T o = __shfl_xor(vmin, off, 64);
    vmin = o < vmin ? o : vmin;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = vmin;
  __syncthreads();
  if (threadIdx.x == 0) {
    T m = smin[0];
    for (int i = 1; i < BLOCK / 64; ++i) m = smin[i] < m ? smin[i] : m;
    minpart[blockIdx.x] = m;
    for (int j = blockIdx.x + gridDim.x; j < NB; j += gridDim.x) minpart[j] = T(0);      // 0 = no entry (ignored by the reader)
  }
}

// First-pass reductions of the two-pass projectors, kept cheap: sums in float64, counts in 32-bit
// integers, threshold compares in the working precision (thresholds are stored TF-rounded).
// Slot layout of the block partials: 0 ||v||_1, 1 ||v||_2^2, 2 nnz, 3.. S_k = sum(|v| > t_k),
// 3+L1_K.. C_k = count(|v| > t_k).
template <typename T>
struct ProbeAcc {
  double asum = 0, sumsq = 0, S[L1_K];
  unsigned int nnz = 0, C[L1_K];
  T t[L1_K];
  __device__ __forceinline__ ProbeAcc() {
#pragma unroll
    for (int k = 0; k < L1_K; ++k) { S[k] = 0; C[k] = 0; t[k] = (T)INFINITY; }
  }
  __device__ __forceinline__ void add(T av, T x) {
    const double a = (double)av;
    asum += a;
    sumsq += (double)x * (double)x;
    nnz += av > T(0) ? 1u : 0u;
#pragma unroll
    for (int k = 0; k < L1_K; ++k) {
      const bool on = av > t[k];
      S[k] += on ? a : 0.0;
      C[k] += on ? 1u : 0u;
    }
  }
  // LEAN first pass (the speculation has been holding): only ||v||_1 and the two probes at the edges of the speculative
  // range -- a fifth of the arithmetic; the other slots are left at zero and must not be read by the decision
  __device__ __forceinline__ void add_lean(T av, int k_lo, int k_hi) {
    const double a = (double)av;
    asum += a;
#pragma unroll
    for (int k = 0; k < L1_K; ++k) {
      if (k != k_lo && k != k_hi) continue;
      const bool on = av > t[k];
      S[k] += on ? a : 0.0;
      C[k] += on ? 1u : 0u;
    }
  }
  __device__ __forceinline__ void to_slots(double (&acc)[PREP_SLOTS]) const {
    acc[0] = asum; acc[1] = sumsq; acc[2] = (double)nnz;
#pragma unroll
    for (int k = 0; k < L1_K; ++k) { acc[3 + k] = S[k]; acc[3 + L1_K + k] = (double)C[k]; }
  }
};

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
// vector range [v0, v1) of a launch over the grid (Grid::e0, e1; V divides both: planes are multiples of n1)
template <int V>
__device__ __forceinline__ void vec_range(const Grid& G, long long& v0, long long& v1) {
  v0 = G.e0 / V;
  v1 = (G.e1 < 0 ? G.N : G.e1) / V;
}
// Coordinates of linear index g (column-major, dim 1 fastest).  N < 2^31 is enforced at sipx_create.
__device__ __forceinline__ Coord coords(const Grid& G, long long g) {
  Coord c;
  const unsigned u = (unsigned)g, n1 = (unsigned)G.n[0], n2 = (unsigned)G.n[1];
  const unsigned jk = G.m1 ? (__umulhi(u, G.m1) >> G.s1) : u / n1;
  c.i = (int)(u - jk * n1);
  const unsigned k = G.m2 ? (__umulhi(jk, G.m2) >> G.s2) : jk / n2;
  c.j = (int)(jk - k * n2);
  c.k = (int)k;
  return c;
}
__device__ __forceinline__ int coord_of(const Coord& c, int dir) { return dir == 0 ? c.i : (dir == 1 ? c.j : c.k); }

// Forward difference along `dir` at the V consecutive points g..g+V-1 (same line):
// s = (-ih)*x[g] + ih*x[g+stride], the two products of a CSC row in column order
// (reference get_discrete_Grad.jl:22-23,58-60 + SparseArrays mul!).  Points on the last
// hyper-plane along `dir` are pads of the padded layout: valid=false, s=0.
// BRANCH-FREE for every direction: x carries an end halo of >= max stride elements (engine allocation),
// the neighbour vector x[g+stride ..] is ONE unconditional element-aligned load (stride 1 included) and
// validity is a select, so all loads of a work item issue back to back.
template <typename T, int V>
__device__ __forceinline__ void fwd_dir(const Grid& G, const T* __restrict__ x, const Vec<T, V>& xc, long long g,
                                        const Coord& c, int dir, T ih, T (&s)[V], bool (&valid)[V]) {
  const T nih = -ih;
  const long long st = dir == 0 ? 1 : (dir == 1 ? G.st[1] : G.st[2]);
  const int nd = (int)(dir == 0 ? G.n[0] : (dir == 1 ? G.n[1] : G.n[2]));
  const int cd = coord_of(c, dir);
  const Vec<T, V> xn = ldv_u<T, V>(x + g + st);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    valid[k] = (cd + (dir == 0 ? k : 0)) < nd - 1;
    const T d = nih * xc.v[k] + ih * xn.v[k];
    s[k] = valid[k] ? d : T(0);
  }
}

// Adjoint of the forward difference along `dir`, accumulated into t[] at grid points g..g+V-1:
// t += ih*w[g-stride] (if that row exists) ; t += (-ih)*w[g] (if row g exists) -- a CSC column
// of D in ascending row order (SparseArrays mul!(tmp, A', v)).  W(e) loads the V values of w at padded
// index e (any element alignment).  BRANCH-FREE: w carries a front halo of >= max stride elements.
template <typename T, int V, typename WV>
__device__ __forceinline__ void adj_dir_acc(const Grid& G, long long g, const Coord& c, int dir, T ih, T (&t)[V],
                                            WV wv) {
  const T nih = -ih;
  const long long st = dir == 0 ? 1 : (dir == 1 ? G.st[1] : G.st[2]);
  const int nd = (int)(dir == 0 ? G.n[0] : (dir == 1 ? G.n[1] : G.n[2]));
  const int cd = coord_of(c, dir);
  const Vec<T, V> wc = wv(g);
  const Vec<T, V> wp = wv(g - st);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int ck = cd + (dir == 0 ? k : 0);
    const T t1 = t[k] + ih * wp.v[k];
    t[k] = (ck > 0) ? t1 : t[k];
    const T t2 = t[k] + nih * wc.v[k];
    t[k] = (ck < nd - 1) ? t2 : t[k];
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
    case PX_BOUNDS_VEC: {
      if (c.plo != T(0)) {                  // per-fiber bounds: min(max(x, LB), UB)    project_bounds!.jl:47,51,65
        T t = v > lb ? v : lb;
        return t < ub ? t : ub;
      }
      T t = v < ub ? v : ub;                // project_bounds!.jl:21-22
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
