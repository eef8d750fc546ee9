// Projectors that lean on a library transform: they act on a materialised vector v (N reals).
//
//   DFT-folded l1 ball  x -> Re( F^H P_l1( F x ) ), F the unitary 3-D/2-D DFT
//       reference: get_projector.jl:29-35 with A = joDFT(...) (get_TD_operator.jl:45-47,80-82),
//       project_l1_Duchi! on Complex{TF} (project_l1_Duchi!.jl:29-32,49).  hipFFT does the
//       (unnormalised) transforms; the unitary 1/sqrt(N) factors are folded into the radius
//       (b*sqrt(N) on the raw spectrum) and into the inverse (1/N); the threshold search is the
//       engine's own l1 machinery on the magnitudes.  The joDFT normalisation is NOT pinned by
//       any reference test: unitary is assumed because the operator declares AtA_diag.
//   slice / matrix rank  x[:,:,i] <- U_r S_r V_r'      reference: projectors/project_rank!.jl:3-48
//       rocSOLVER batched SVD + rocBLAS batched GEMM: the one unit of the path that is not
//       bandwidth bound, so it is a library call, not a hand-written kernel.
#define ROCBLAS_BETA_FEATURES_API 1
#define ROCBLAS_NO_DEPRECATED_WARNINGS 1
#include <hipcub/hipcub.hpp>
#include <hipfft/hipfft.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sipx.h"
#include "ext_proj.h"
#include "sipx_device.h"

namespace sipx {

template <typename T>
struct Cplx {
  T re, im;
};

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_pack(long long N, const T* __restrict__ v, Cplx<T>* __restrict__ z) {
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK) {
    Cplx<T> c;
    c.re = v[e];
    c.im = T(0);
    z[e] = c;
  }
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_cabs(long long N, const Cplx<T>* __restrict__ z, T* __restrict__ mag) {
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK)
    mag[e] = (T)hypot((double)z[e].re, (double)z[e].im);
}
// z <- sign(z) * max(|z| - theta, 0), sign(z) = z/|z|   (project_l1_Duchi!.jl:49 on complex input)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_csoft(long long N, Cplx<T>* __restrict__ z, const T* __restrict__ mag,
                                                 const ProjScalars<T>* __restrict__ ps) {
  const T th = ps->theta;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK) {
    const T a = mag[e];
    T t = a - th;
    t = t > T(0) ? t : T(0);
    const T f = a > T(0) ? t / a : T(0);
    Cplx<T> c = z[e];
    c.re = c.re * f;
    c.im = c.im * f;
    z[e] = c;
  }
}
// z <- z .* mask: project_bounds! on a complex vector with binary bounds (project_bounds!.jl:27-36: x .= x .* UB)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_cmask(long long N, Cplx<T>* __restrict__ z, const T* __restrict__ mask) {
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK) {
    Cplx<T> c = z[e];
    c.re = c.re * mask[e];
    c.im = c.im * mask[e];
    z[e] = c;
  }
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_unpack_all(long long N, const Cplx<T>* __restrict__ z, T* __restrict__ v, T scale) {
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK)
    v[e] = z[e].re * scale;
}
// v <- Re(z)/N -- skipped when v already lies inside the ball: F'F = I, so the reference's round trip
// A'*(A*x) (get_projector.jl:31) only adds FFT rounding noise there; v is returned bit for bit instead
// (the noise would otherwise be amplified by the BB rule: l = rho*(y - s) would be pure rounding error).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_unpack(long long N, const Cplx<T>* __restrict__ z, T* __restrict__ v, T scale,
                                                  const ProjScalars<T>* __restrict__ ps) {
  if (!ps->need) return;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK)
    v[e] = z[e].re * scale;
}
// The same set through the REAL transform (round 4): the model is real, so the hipFFT R2C transform returns the nh1 = n1/2 + 1
// planes k1 = 0 .. n1/2 of the spectrum and the other n1 - nh1 are their conjugates -- half the transform, no packing.  The l1 norm
// runs over ALL N coefficients: the magnitudes of the stored ones fill mag[0, Nh), those of the planes 1 .. n1 - nh1 (whose
// conjugates are not stored) are written a second time behind them, N entries in all, and the search sees the vector it always saw.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_cabs_half(long long Nh, int nh1, int ndup, const Cplx<T>* __restrict__ z, T* __restrict__ mag) {
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < Nh; e += (long long)gridDim.x * BLOCK) {
    const T m = (T)hypot((double)z[e].re, (double)z[e].im);
    mag[e] = m;
    const long long row = e / nh1;
    const int k1 = (int)(e - row * nh1);
    if (k1 >= 1 && k1 <= ndup) mag[Nh + row * ndup + (k1 - 1)] = m;
  }
}
// v <- w * scale (w: the output of the C2R transform) unless v already lies inside the ball (see k_unpack)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_unpack_real(long long N, const T* __restrict__ w, T* __restrict__ v, T scale,
                                                       const ProjScalars<T>* __restrict__ ps) {
  if (!ps->need) return;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK) v[e] = w[e] * scale;
}
// U[:, j] *= S[j] for the first r columns of every slice
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_scale_cols(int m, int r, int ldu, long long strideU, long long strideS,
                                                      int batch, T* __restrict__ U, const T* __restrict__ S) {
  const long long tot = (long long)batch * r * m;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < tot; e += (long long)gridDim.x * BLOCK) {
    const int i = (int)(e % m);
    const long long t = e / m;
    const int j = (int)(t % r);
    const long long b = t / r;
    U[b * strideU + (long long)j * ldu + i] *= S[b * strideS + j];
  }
}
// ------------------------------------------------------------------------------------------------
// Segments of the padded array: the whole valid block, its fibers along one direction, or its slices.
// Element t of segment s lives at seg_addr(s, t); t runs in the reference's order (lower dimension fastest:
// reshape / permutedims of project_cardinality!.jl:111-120, project_subspace!.jl:91-100, view(x,i,:,:) etc.).
struct SegMap {
  long long nseg, L;
  long long SA, sSa, sSb;            // s -> (s % SA) * sSa + (s / SA) * sSb
  long long LA, LB, sTa, sTb, sTc;   // t = ta + LA * (tb + LB * tc) -> ta * sTa + tb * sTb + tc * sTc
};
__device__ __forceinline__ long long seg_addr(const SegMap& m, long long s, long long t) {
  const long long ta = t % m.LA, r = t / m.LA, tb = r % m.LB, tc = r / m.LB;
  return (s % m.SA) * m.sSa + (s / m.SA) * m.sSb + ta * m.sTa + tb * m.sTb + tc * m.sTc;
}
static SegMap make_segmap(const ExtSpec& sp) {
  SegMap m{};
  const long long* d = sp.dims;
  const long long* st = sp.G.st;
  m.SA = 1; m.LA = 1; m.LB = 1;
  if (sp.mode == SIPX_MODE_WHOLE) {
    m.nseg = 1; m.L = d[0] * d[1] * d[2];
    m.LA = d[0]; m.LB = d[1]; m.sTa = st[0]; m.sTb = st[1]; m.sTc = st[2];
  } else {
    const int dir = sp.dir;
    if (dir < 0 || dir > 2) throw std::runtime_error("application mode: direction out of range");
    const int a = dir == 0 ? 1 : 0, b = dir == 2 ? 1 : 2;
    if (sp.mode == SIPX_MODE_FIBER) {
      m.L = d[dir]; m.LA = d[dir]; m.sTa = st[dir];
      m.SA = d[a]; m.sSa = st[a]; m.sSb = st[b]; m.nseg = d[a] * d[b];
    } else if (sp.mode == SIPX_MODE_SLICE) {
      m.LA = d[a]; m.LB = d[b]; m.sTa = st[a]; m.sTb = st[b]; m.L = d[a] * d[b];
      m.SA = d[dir]; m.sSa = st[dir]; m.nseg = d[dir];
    } else {
      throw std::runtime_error("unknown application mode");
    }
  }
  return m;
}

// dense[s * L + t] <- v[seg_addr(s, t)] and back (U = double for the SVD path, T otherwise).  `flag` (optional):
// segments with flag[s] == 0 are left untouched by the scatter.
template <typename T, typename U>
__global__ __launch_bounds__(BLOCK) void k_seg_gather(SegMap m, const T* __restrict__ v, U* __restrict__ dense) {
  const long long tot = m.nseg * m.L;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < tot; e += (long long)gridDim.x * BLOCK) {
    const long long s = e / m.L, t = e - s * m.L;
    dense[e] = (U)v[seg_addr(m, s, t)];
  }
}
template <typename T, typename U>
__global__ __launch_bounds__(BLOCK) void k_seg_scatter(SegMap m, const U* __restrict__ dense, T* __restrict__ v,
                                                       const int* __restrict__ flag) {
  const long long tot = m.nseg * m.L;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < tot; e += (long long)gridDim.x * BLOCK) {
    const long long s = e / m.L, t = e - s * m.L;
    if (flag && !flag[s]) continue;
    v[seg_addr(m, s, t)] = (T)dense[e];
  }
}

// ------------------------------------------------------------------------------------------------
// Cardinality per segment (project_cardinality!.jl:23-146): keep the k entries of largest magnitude of every fiber /
// slice, zero the rest; equal magnitudes keep the earlier entry (sortperm(by=abs, rev=true) is stable).
// One workgroup per segment: radix select on the magnitude bit pattern (8 bits a pass, histogram in LDS),
// then one ordered pass that resolves the tie cut with wave ballots.
template <typename T> struct KeyOf;
template <> struct KeyOf<float> { using U = unsigned int; };
template <> struct KeyOf<double> { using U = unsigned long long; };
__device__ __forceinline__ unsigned int abs_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned long long abs_key(double v) {
  return (unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull;
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_seg_card(SegMap m, T* __restrict__ v, long long k) {
  using U = typename KeyOf<T>::U;
  constexpr int BITS = (int)sizeof(U) * 8;
  __shared__ unsigned int hist[256];
  __shared__ U s_prefix;
  __shared__ long long s_kk, s_run;
  __shared__ unsigned int s_wtot[BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (k >= m.L) return;                                   // sort_ind[k+1:end] is empty
  for (long long s = blockIdx.x; s < m.nseg; s += gridDim.x) {
    U prefix = 0, mask = 0;
    long long kk = k;
    if (k > 0) {
      for (int shift = BITS - 8; shift >= 0; shift -= 8) {
        hist[tid] = 0;                                    // BLOCK == 256 bins
        __syncthreads();
        for (long long t = tid; t < m.L; t += BLOCK) {
          const U key = abs_key(v[seg_addr(m, s, t)]);
          if ((key & mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          long long cum = 0;
          int b = 255;
          for (; b > 0; --b) {
            if (cum + hist[b] >= kk) break;
            cum += hist[b];
          }
          s_prefix = prefix | ((U)b << shift);
          s_kk = kk - cum;
        }
        __syncthreads();
        prefix = s_prefix;
        kk = s_kk;
        mask |= (U)255 << shift;
      }
    }
    // prefix = k-th largest magnitude, kk = how many entries equal to it survive (the earliest ones)
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (long long c0 = 0; c0 < m.L; c0 += BLOCK) {
      const long long t = c0 + tid;
      const bool live = t < m.L;
      const long long addr = live ? seg_addr(m, s, t) : 0;
      const U key = live ? abs_key(v[addr]) : 0;
      const bool eq = live && k > 0 && key == prefix, gt = live && k > 0 && key > prefix;
      const unsigned long long bal = __ballot(eq);
      const unsigned rank = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wtot[w] = (unsigned)__popcll(bal);
      __syncthreads();
      long long off = s_run;
      for (int i = 0; i < w; ++i) off += s_wtot[i];
      const bool keep = gt || (eq && off + rank < kk);
      if (live && !keep) v[addr] = T(0);
      __syncthreads();
      if (tid == 0) {
        long long tot = 0;
        for (int i = 0; i < BLOCK / 64; ++i) tot += s_wtot[i];
        s_run += tot;
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Singular values of every slice projected onto the l1 ball of radius sigma (project_nuclear!.jl:19-20,40-41 with
// project_l1_Duchi!.jl:23,40-46 on the descending values).  flag[b] = 1 when slice b changed.
__global__ void k_nuc_shrink(int kmin, int batch, double sigma, double* __restrict__ S, int* __restrict__ flag) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double* s = S + (long long)b * kmin;
  double sum = 0;
  for (int j = 0; j < kmin; ++j) sum += s[j];
  if (sum <= sigma) {                       // norm(v, 1) <= b && return v
    flag[b] = 0;
    return;
  }
  int rho = 0;
  double cum = 0;
  for (;;) {                                // while u[rho+1] > (sv[rho+1] - b)/(rho+1) && rho+1 < lv
    const double nxt = cum + s[rho];
    if (s[rho] > (nxt - sigma) / (double)(rho + 1) && rho + 1 < kmin) {
      cum = nxt;
      ++rho;
    } else {
      break;
    }
  }
  if (rho == 0) { rho = 1; cum = s[0]; }    // rho = max(1, rho)
  double theta = (cum - sigma) / (double)rho;
  theta = theta > 0 ? theta : 0;
  for (int j = 0; j < kmin; ++j) {
    const double t = s[j] - theta;
    s[j] = t > 0 ? t : 0;
  }
  flag[b] = 1;
}

// Gram route (Float32 models): W = ascending eigenvalues of X'X (or XX'), sigma_j = sqrt(W_j).  F_j = shrunk(sigma_j)/sigma_j
// for the nuclear-norm ball (same scan as k_nuc_shrink on the descending values), flag[b] = 0 when slice b is inside it.
__global__ void k_nuc_factors(int kmin, int batch, double sigma, const double* __restrict__ W, double* __restrict__ F,
                              int* __restrict__ flag) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const double* w = W + (long long)b * kmin;
  double* f = F + (long long)b * kmin;
  auto sv = [&](int j) {                     // j-th largest singular value
    const double l = w[kmin - 1 - j];
    return l > 0 ? sqrt(l) : 0.0;
  };
  double sum = 0;
  for (int j = 0; j < kmin; ++j) sum += sv(j);
  if (sum <= sigma) {
    flag[b] = 0;
    for (int j = 0; j < kmin; ++j) f[j] = 1.0;
    return;
  }
  int rho = 0;
  double cum = 0;
  for (;;) {
    const double nxt = cum + sv(rho);
    if (sv(rho) > (nxt - sigma) / (double)(rho + 1) && rho + 1 < kmin) {
      cum = nxt;
      ++rho;
    } else {
      break;
    }
  }
  if (rho == 0) { rho = 1; cum = sv(0); }
  double theta = (cum - sigma) / (double)rho;
  theta = theta > 0 ? theta : 0;
  for (int j = 0; j < kmin; ++j) {           // f is indexed like W (ascending)
    const double l = w[j], sg = l > 0 ? sqrt(l) : 0.0, t = sg - theta;
    f[j] = (sg > 0 && t > 0) ? t / sg : 0.0;
  }
  flag[b] = 1;
}
// Gs[:, j] = G[:, j] * F[j] for every slice (k x k eigenvector matrices)
__global__ __launch_bounds__(BLOCK) void k_scale_eigvecs(int k, int batch, const double* __restrict__ G, const double* __restrict__ F,
                                                         double* __restrict__ Gs) {
  const long long tot = (long long)batch * k * k;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < tot; e += (long long)gridDim.x * BLOCK) {
    const long long b = e / ((long long)k * k);
    const int j = (int)((e / k) % k);
    Gs[e] = G[e] * F[b * k + j];
  }
}

// ------------------------------------------------------------------------------------------------
// Relaxed histogram (project_histogram_relaxed.jl:9-27): the j-th smallest entry is clipped to [LB[j], UB[j]].
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_hist_gather(SegMap m, const T* __restrict__ v, T* __restrict__ keys,
                                                       unsigned int* __restrict__ idx) {
  for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < m.L; t += (long long)gridDim.x * BLOCK) {
    keys[t] = v[seg_addr(m, 0, t)];
    idx[t] = (unsigned int)t;
  }
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_hist_apply(SegMap m, const T* __restrict__ keys, const unsigned int* __restrict__ idx,
                                                      const T* __restrict__ lb, const T* __restrict__ ub, T* __restrict__ v) {
  for (long long j = (long long)blockIdx.x * BLOCK + threadIdx.x; j < m.L; j += (long long)gridDim.x * BLOCK) {
    T x = keys[j];
    x = x < ub[j] ? x : ub[j];              // min(x, UB) first, then max(LB, x)
    x = lb[j] > x ? lb[j] : x;
    v[seg_addr(m, 0, (long long)idx[j])] = x;
  }
}

// sum (a-b)^2 and sum b^2 into partial slots 0,1
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dist2(long long N, const T* __restrict__ a, const T* __restrict__ b,
                                                 double* __restrict__ partials) {
  double acc[2] = {0, 0};
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK) {
    const T d = a[e] - b[e];
    acc[0] += (double)d * (double)d;
    acc[1] += (double)b[e] * (double)b[e];
  }
  block_reduce_store<2>(acc, partials, 0);
}
template <typename T>
void ext_dist2(hipStream_t s, long long N, const T* projected, const T* original, double* dst) {
  hipLaunchKernelGGL((k_dist2<T>), dim3(NB), dim3(BLOCK), 0, s, N, projected, original, dst);
  SIPX_HIP(hipGetLastError());
}

// ---- block subspace iteration on the Gram matrices (rank projection, see ExtProj::project) ----
// Columns of Y (k x b per matrix) scaled to unit length; a column that vanished against the largest one (rank of the
// matrix below b) is replaced by fixed pseudo-random numbers so that the Cholesky factor of Y'Y exists.
__global__ __launch_bounds__(BLOCK) void k_sub_normalize(int k, int b, int batch, double* __restrict__ Y) {
  __shared__ double sm[BLOCK / 64];
  __shared__ double s_norm[64];      // b <= 64 is enforced by the caller
  const int l = blockIdx.x;
  double* Yl = Y + (long long)l * k * b;
  for (int j = 0; j < b; ++j) {
    double a = 0;
    for (int i = threadIdx.x; i < k; i += BLOCK) a += Yl[(long long)j * k + i] * Yl[(long long)j * k + i];
    a = wave_sum(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0;
      for (int w = 0; w < BLOCK / 64; ++w) t += sm[w];
      s_norm[j] = sqrt(t);
    }
  }
  __syncthreads();
  double mx = 0;
  for (int j = 0; j < b; ++j) mx = s_norm[j] > mx ? s_norm[j] : mx;
  for (int j = 0; j < b; ++j) {
    const double nj = s_norm[j];
    if (nj > 1e-13 * mx && nj > 0) {
      const double inv = 1.0 / nj;
      for (int i = threadIdx.x; i < k; i += BLOCK) Yl[(long long)j * k + i] *= inv;
    } else {
      for (int i = threadIdx.x; i < k; i += BLOCK) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)((j + 1) * 40503u) ^ (unsigned)(l * 69069u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        Yl[(long long)j * k + i] = ((double)(h >> 8) / 16777216.0 - 0.5) / sqrt((double)k / 12.0);
      }
    }
  }
}
// ||G_l||_F^2 of every matrix of the batch: FRO_PARTS workgroups per matrix, each over one contiguous part with four sums in
// flight per thread, then the parts added in order by k_sub_fro_sum -- the same bits for a matrix whatever the batch it sits in.
// (One workgroup per matrix with one dependent sum per thread, rounds 3-5: 0.97 ms for 512 Gram matrices of 512^2 -- 1.1 TB/s --
//  and 0.40 ms for the 64 of a rank's share, once per call.)
#define FRO_PARTS 16
__global__ __launch_bounds__(BLOCK) void k_sub_fro_part(int k, const double* __restrict__ G, double* __restrict__ part) {
  __shared__ double sm[BLOCK / 64];
  const long long kk = (long long)k * k, seg = (kk + FRO_PARTS - 1) / FRO_PARTS;
  const long long l = blockIdx.x / FRO_PARTS, p = blockIdx.x % FRO_PARTS;
  const double* Gl = G + l * kk;
  const long long s0 = p * seg, s1 = s0 + seg < kk ? s0 + seg : kk;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  long long i = s0 + threadIdx.x;
  for (; i + 3 * BLOCK < s1; i += 4 * BLOCK) {
    const double v0 = Gl[i], v1 = Gl[i + BLOCK], v2 = Gl[i + 2 * BLOCK], v3 = Gl[i + 3 * BLOCK];
    a0 += v0 * v0; a1 += v1 * v1; a2 += v2 * v2; a3 += v3 * v3;
  }
  for (; i < s1; i += BLOCK) a0 += Gl[i] * Gl[i];
  double a = wave_sum((a0 + a1) + (a2 + a3));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < BLOCK / 64; ++w) t += sm[w];
    part[blockIdx.x] = t;
  }
}
__global__ void k_sub_fro_sum(int batch, const double* __restrict__ part, double* __restrict__ fro2, unsigned long long* res) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= batch) return;
  double t = 0;
  for (int p = 0; p < FRO_PARTS; ++p) t += part[(long long)l * FRO_PARTS + p];
  fro2[l] = t;
  if (res) atomicMax(res + 7, (unsigned long long)__double_as_longlong(t));      // t >= 0: the bit patterns order like the values
}
static void sub_fro(hipStream_t s, int k, int batch, const double* G, double* part, double* fro2, unsigned long long* res = nullptr) {
  hipLaunchKernelGGL(k_sub_fro_part, dim3((unsigned)batch * FRO_PARTS), dim3(BLOCK), 0, s, k, G, part);
  hipLaunchKernelGGL(k_sub_fro_sum, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, s, batch, part, fro2, res);
}
// A start for a projector that has none (the first call of a solve; project_rank!.jl:26-45 has no state at all): fixed
// pseudo-random columns, the same hash as the re-seeded columns of k_sub_normalize.
__global__ __launch_bounds__(BLOCK) void k_sub_seed(int k, int b, int batch, double* __restrict__ X) {
  const long long per = (long long)k * b, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per, o = e - l * per;
    const int j = (int)(o / k), i = (int)(o - (long long)j * k);
    unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)((j + 1) * 40503u) ^ (unsigned)((unsigned)l * 69069u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    X[e] = ((double)(h >> 8) / 16777216.0 - 0.5) / sqrt((double)k / 12.0);
  }
}
// Largest residual of the top-r Ritz pairs, relative to the largest Ritz value: max_j ||(G Q) z_j - theta_j x_j|| / theta_max,
// with ZH = (G Q) Z and X = Q Z given (k x b per matrix, Ritz values ascending).  res[0] collects the maximum over the
// batch (bit pattern of a non-negative double), res[1] is raised when a factorisation failed or a value is not finite.
// eps_bw > 0 (Float32 models, round 5): the residual of pair j is measured against what a backward stable Float32 SVD of the slice
// X itself leaves -- project_rank!.jl:28-41 calls svd() in TF.  With theta = x'Gx the residual rho = G x - theta x is orthogonal to
// x, and (sigma, u = X x / sigma, x) is an EXACT singular triplet of X + E with E = -u rho' / sigma, ||E||_2 = ||rho|| / sigma_j: the
// pair is accepted when that is at most eps_bw ||X||_2, i.e. ||rho_j|| <= eps_bw sqrt(theta_max theta_j).  What is stored and compared
// with tol everywhere (the host's decisions, k_cheb_plan, k_sub_list) is the residual in units of its acceptance level times tol,
// never asking for more than the strict level tol theta_max; res[6] <- the largest residual / theta_max as before (what a start is
// judged by).
template <typename TS>
__global__ __launch_bounds__(BLOCK) void k_sub_residual(int k, int b, int r, int batch, const TS* __restrict__ ZH,
                                                        const TS* __restrict__ X, const double* __restrict__ W, int ldw,
                                                        const rocblas_int* __restrict__ info_chol,
                                                        const rocblas_int* __restrict__ info_eig,
                                                        const double* __restrict__ fro2, unsigned long long* res,
                                                        double* __restrict__ per_matrix = nullptr, double eps_bw = 0.0,
                                                        double tol = 1e-12, int extra = 0, const int* __restrict__ nd = nullptr,
                                                        const double* __restrict__ tmax_of = nullptr) {
  // nd / tmax_of (the Float32 iteration on the deflated matrices): the nd[l] largest pairs of matrix l were taken out of the matrix
  // before it was rounded to Float32 -- the block then owes r - nd[l] pairs -- and the residuals are still measured against the
  // largest eigenvalue of the matrix AS IT WAS, tmax_of[l]
  __shared__ double sm[BLOCK / 64];
  const int l = blockIdx.x;
  if (nd) r -= nd[l];
  const double tmax = tmax_of ? tmax_of[l] : W[(long long)l * ldw + b - 1];
  if (r <= 0) {                       // every wanted pair was deflated: nothing left to find here
    if (threadIdx.x == 0 && per_matrix) per_matrix[l] = 0.0;
    return;
  }
  if (fro2[l] == 0.0) {               // a slice of zeros (the first iteration of a solve projects v = 0): nothing to find, nothing to certify
    if (threadIdx.x == 0 && per_matrix) per_matrix[l] = 0.0;
    return;
  }
  double worst = 0, worst_raw = 0;
  // (extra: the pairs BELOW the r wanted ones that are held to the same level -- the eigenvalues of these slices cross from one
  //  PARSDMM iteration to the next, and a pair that enters the top r next time is then one the block already holds)
  for (int j = (b - r - extra > 0 ? b - r - extra : 0); j < b; ++j) {
    const double th = W[(long long)l * ldw + j];
    const TS* z = ZH + ((long long)l * b + j) * k;
    const TS* x = X + ((long long)l * b + j) * k;
    double a = 0;
    for (int i = threadIdx.x; i < k; i += BLOCK) {
      const double d = (double)z[i] - th * (double)x[i];
      a += d * d;
    }
    a = wave_sum(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < BLOCK / 64; ++w) t += sm[w];
    const double raw = sqrt(t) / tmax;
    double rel = raw;
    if (eps_bw > 0.0) {
      const double lvl = eps_bw * sqrt(tmax * (th > 0.0 ? th : 0.0));
      if (lvl > tol * tmax) rel = sqrt(t) / lvl * tol;
    }
    worst = rel > worst ? rel : worst;
    worst_raw = raw > worst_raw ? raw : worst_raw;
  }
  if (threadIdx.x == 0 && per_matrix) per_matrix[l] = worst;
  if (threadIdx.x == 0) {
    // Certificate that no eigenvalue above the r-th Ritz value hides outside the converged pairs.  With P the projector
    // on the subspace, ||G||_F^2 = ||H||_F^2 + 2 ||(I-P) G P||_F^2 + ||(I-P) G (I-P)||_F^2 and ||H||_F^2 = sum of the
    // squared Ritz values, so rest = ||G||_F^2 - sum theta^2 bounds both the coupling (<= sqrt(rest/2)) and everything
    // in the complement (<= sqrt(rest)).  Once the top-r pairs are eigenpairs, the other eigenvalues of G are those of the
    // block [guard Ritz part, coupling; coupling', complement] <= max(theta_{r+1}, sqrt(rest)) + sqrt(rest/2).
    double ritz2 = 0;
    for (int j = 0; j < b; ++j) ritz2 += W[(long long)l * ldw + j] * W[(long long)l * ldw + j];
    const double rest = fmax(fro2[l] - ritz2, 0.0) + 1e-12 * fro2[l];
    const double others = fmax(W[(long long)l * ldw + b - r - 1], sqrt(rest)) + sqrt(0.5 * rest);
    const bool hidden = !(others < W[(long long)l * ldw + b - r]);
    const unsigned long long bad = (info_chol[l] != 0 ? 1ull : 0ull) | (info_eig[l] != 0 ? 2ull : 0ull) | (!(tmax > 0) ? 4ull : 0ull) |
                                   ((!(worst == worst) || isinf(worst)) ? 8ull : 0ull) | (hidden ? 16ull : 0ull);
    if (bad & 15ull) atomicOr(res + 1, bad);     // 1: Cholesky, 2: Ritz solver, 4: no positive Ritz value, 8: not finite
    else {
      if (hidden) atomicOr(res + 1, 16ull);      // 16: not certified (yet): fine while the residual is still above the tolerance
      atomicMax(res, (unsigned long long)__double_as_longlong(worst));
      atomicMax(res + 6, (unsigned long long)__double_as_longlong(worst_raw));
    }
  }
}
// ---- Chebyshev-filtered subspace iteration (rank projection on spectra without a gap behind the block) ----
// Column j of the block carries its own damped interval [0, a_j], a_j = max(a, theta_j / CHEB_KAPPA), a = the end of the part of
// the spectrum the block does not hold: everything outside the block (eigenvalues <= a) is damped in every column, the
// column's own direction grows like T_m(2 theta_j / a_j - 1).  Directions whose Ritz value exceeds CHEB_KAPPA max(theta_j, a)
// would outgrow column j by (theta_i / a_j)^m -- the constant part of a velocity slice is 1e5 times the rest -- so they are
// projected out of the product G y_j at every step (k_cheb_mask: the filter then runs in the compression of G onto their
// complement, whose spectrum has lost them up to the square of their error).  The Rayleigh-Ritz step behind the filter works
// on the whole block again.
#define CHEB_KAPPA 2.0
// a = the Ritz value of guard column g (a few columns above the lowest; the g columns below it are not filtered), not the
// lowest: the lowest Ritz values of a block that has not converged lie below eigenvalues the block does not hold, and what
// lies above a is amplified -- an interval 5 % short loses a factor T_m(1.1) (250 at degree 14) of the contraction, one 5 %
// long about 10.  And never below 0.9 of the (r+1)-th Ritz value: a guard column that had to be re-seeded (k_chol_inv) carries
// a Rayleigh quotient from the middle of the spectrum, far below the block's true lower end.
__device__ double g_cheb_floor_factor = 0.9;      // SIPX_RANK_FLOOR (experiments): the interval never ends below this share of the (r+1)-th Ritz value
__device__ __forceinline__ double cheb_floor(const double* __restrict__ W, int b, int g, int r) {
  const double top = W[b - 1];
  double a = W[g];
  const double lo = g_cheb_floor_factor * W[b - r - 1];
  a = a > lo ? a : lo;
  return a > 1e-14 * top ? a : 1e-14 * top;      // a block deeper than the rank of the matrix: no interval of zero width
}
// C (nl x b per matrix, leading dimension b) = X_L' Z for the nl last (largest) Ritz vectors: keep entry (i, j) only where
// vector i is far above column j
template <typename TS>
__global__ void k_cheb_mask(int b, int g, int r, int nl, int batch, const double* __restrict__ W, TS* __restrict__ C,
                            const int* __restrict__ nd = nullptr) {
  const long long per = (long long)nl * b, total = per * batch;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long l = e / per;
    const int o = (int)(e - l * per), j = o / nl, i = b - nl + (o - j * nl);
    const double* Wl = W + l * b;
    const double a = cheb_floor(Wl, b, g, r - (nd ? nd[l] : 0));
    const double tj = Wl[j] > a ? Wl[j] : a;
    if (!(Wl[i] > CHEB_KAPPA * tj)) C[l * (long long)b * b + (long long)j * b + (o - j * nl)] = TS(0);
  }
}
// One step of the three-term recurrence, per column scalars: first = 1: out = (2/a_j) Z - Y0;  else out = (4/a_j) Z - 2 Y1 - Y0
// (out may alias Z or Y0: every entry is read before it is written, by the same thread)
template <typename TS>
__global__ __launch_bounds__(BLOCK) void k_cheb_step(int k, int b, int g, int r, int batch, const double* __restrict__ W, const TS* Z,
                                                     const TS* Y1, const TS* Y0, TS* out, int first, const int* __restrict__ nd = nullptr) {
  const long long per = (long long)k * b, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per;
    const int j = (int)((e - l * per) / k);
    const double* Wl = W + l * b;
    const double a = cheb_floor(Wl, b, g, r - (nd ? nd[l] : 0));
    const double aj = Wl[j] / CHEB_KAPPA > a ? Wl[j] / CHEB_KAPPA : a;
    // (TS = float: the coefficients are rounded once, the recurrence runs in Float32 like the products it combines)
    out[e] = first ? TS(2.0 / aj) * Z[e] - Y0[e] : TS(4.0 / aj) * Z[e] - TS(2) * Y1[e] - Y0[e];
  }
}
// Columns the filter must not touch (round 5): the g lowest ones as before, and EVERY column whose Ritz value lies inside the damped
// interval [0, a] of its matrix.  The interval never ends below 0.9 of the (r+1)-th Ritz value (cheb_floor), so on a flat spectrum
// about half of the guard columns lie inside it: their own direction is damped (|T_m| <= 1) while what leaked in from the columns
// above is amplified 1e4 ... 1e7 times -- the column that comes out depends on the others to 1e-4 ... 1e-7, and what the
// orthonormalisation leaves of it is amplified rounding error: a direction of garbage with a Rayleigh quotient anywhere in the
// spectrum.  A dozen of those per call sorted in AMONG the top-r Ritz values (SIPX_EXT_DEBUG=4 showed residuals of 0.1 ... 0.4
// theta_j at columns 14-16, 32 and 43 of 56 beside 1e-5 at their neighbours) and every one of them had to be filtered back into
// an eigenvector before the call could end: 25-45 products per call where 12-17 do.  Such columns now stay the Ritz vectors they
// were: no help, no harm, and their Ritz values remain honest lower bounds (Cauchy interlacing) of the eigenvalues they stand for.
template <typename TS>
__global__ __launch_bounds__(BLOCK) void k_cheb_keep(int k, int b, int g, int r, int batch, const double* __restrict__ W,
                                                     const TS* __restrict__ X, TS* __restrict__ A, const int* __restrict__ nd = nullptr) {
  const long long per = (long long)k * b, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per;
    const int j = (int)((e - l * per) / k);
    const double* Wl = W + l * b;
    if (j < g || !(Wl[j] > cheb_floor(Wl, b, g, r - (nd ? nd[l] : 0)))) A[e] = X[e];
  }
}
// The same step with the projection inside, for the common case that only the nl <= 2 largest Ritz vectors are far above
// anything (a velocity slice: its constant part): one wave per column, z_j - x_i (x_i' z_j) for the masked i, then the recurrence.
// Replaces two skinny GEMMs, the mask kernel and k_cheb_step by one launch.
template <typename TS>
__global__ __launch_bounds__(256) void k_cheb_step_proj(int k, int b, int g, int r, int nl, int batch, const double* __restrict__ W,
                                                        const TS* __restrict__ X, const TS* Z, const TS* Y1, const TS* Y0,
                                                        TS* out, int first, const int* __restrict__ nd = nullptr) {
  const long long col = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // (matrix, column) pairs, one per wave
  if (col >= (long long)b * batch) return;
  const int lane = threadIdx.x & 63;
  const long long l = col / b;
  const int j = (int)(col - l * b);
  const double* Wl = W + l * b;
  const double a = cheb_floor(Wl, b, g, r - (nd ? nd[l] : 0));
  const double tj = Wl[j] > a ? Wl[j] : a;
  const double aj = Wl[j] / CHEB_KAPPA > a ? Wl[j] / CHEB_KAPPA : a;
  const long long base = l * (long long)k * b + (long long)j * k;
  double c0 = 0, c1 = 0;
  const bool m0 = nl >= 1 && Wl[b - 1] > CHEB_KAPPA * tj, m1 = nl >= 2 && Wl[b - 2] > CHEB_KAPPA * tj;
  const TS* x0 = X + l * (long long)k * b + (long long)(b - 1) * k;
  const TS* x1 = X + l * (long long)k * b + (long long)(b - 2) * k;
  if (m0 || m1) {
    for (int i = lane; i < k; i += 64) {
      const double z = (double)Z[base + i];
      if (m0) c0 += (double)x0[i] * z;
      if (m1) c1 += (double)x1[i] * z;
    }
    c0 = wave_sum(c0);
    c1 = wave_sum(c1);
  }
  for (int i = lane; i < k; i += 64) {
    TS z = Z[base + i];
    if (m0) z -= x0[i] * (TS)c0;
    if (m1) z -= x1[i] * (TS)c1;
    out[base + i] = first ? TS(2.0 / aj) * z - Y0[base + i] : TS(4.0 / aj) * z - TS(2) * Y1[base + i] - Y0[base + i];
  }
}
// Cholesky QR without the triangular solve: M = Y'Y (b x b, b <= 64, column-major, both triangles) -> Rinv, the inverse of
// the factor R of M = R'R, so that Q = Y Rinv is one GEMM.  One workgroup per matrix, everything in LDS; the columns are
// scaled to unit length first (M' = D^-1 M D^-1), which is what keeps the factorisation of a filtered block -- columns of
// very different length -- accurate.  info[l] is WRITTEN only on failure (a pivot that is not a number), so that two passes
// can share it.
template <typename TO>
__global__ __launch_bounds__(256) void k_chol_inv(int b, int batch, const double* __restrict__ M, TO* __restrict__ Rinv,
                                                  rocblas_int* __restrict__ info) {
  __shared__ double A[64 * 65];
  __shared__ double Bv[64 * 65];
  __shared__ double dsc[64];
  const int l = blockIdx.x, t = threadIdx.x;
  const double* Ml = M + (long long)l * b * b;
  TO* Rl = Rinv + (long long)l * b * b;
  if (t < b) {
    const double dj = Ml[(long long)t * b + t];
    dsc[t] = dj > 0 ? 1.0 / sqrt(dj) : 0.0;
  }
  __syncthreads();
  for (int e = t; e < b * b; e += 256) {
    const int i = e % b, c = e / b;
    if (i <= c) A[i * 65 + c] = Ml[(long long)c * b + i] * dsc[i] * dsc[c];
    Bv[i * 65 + c] = 0.0;
  }
  bool lost = false;
  for (int j = 0; j < b; ++j) {
    __syncthreads();
    const double piv = A[j * 65 + j];
    if (!(piv == piv)) { lost = true; break; }             // the same value in every thread: a uniform exit
    // a column that depends on the ones before it to rounding (a guard vector the filter left nothing of): it is dropped --
    // unit pivot, no coupling -- and comes back as a vector of negligible length, a Ritz value near zero at the low end
    const bool dep = !(piv > 1e-13);
    const double inv = dep ? 0.0 : 1.0 / sqrt(piv);
    __syncthreads();
    if (t == 0) A[j * 65 + j] = dep ? 1.0 : sqrt(piv);
    for (int c = j + 1 + t; c < b; c += 256) A[j * 65 + c] *= inv;
    __syncthreads();
    const int nrem = b - j - 1;
    for (int e = t; e < nrem * nrem; e += 256) {
      const int ii = e / nrem, cc = e - ii * nrem;
      if (cc >= ii) A[(j + 1 + ii) * 65 + j + 1 + cc] -= A[j * 65 + j + 1 + ii] * A[j * 65 + j + 1 + cc];
    }
  }
  __syncthreads();
  if (lost) {
    if (t == 0) info[l] = 1;
    for (int e = t; e < b * b; e += 256) Rl[e] = (e % b == e / b) ? TO(1) : TO(0);
    return;
  }
  if (t < b) {                                               // column t of the inverse of the (scaled) factor, back substitution
    const int c = t;
    Bv[c * 65 + c] = 1.0 / A[c * 65 + c];
    for (int i = c - 1; i >= 0; --i) {
      double acc = 0;
      for (int q = i + 1; q <= c; ++q) acc += A[i * 65 + q] * Bv[q * 65 + c];
      Bv[i * 65 + c] = -acc / A[i * 65 + i];
    }
  }
  __syncthreads();
  for (int e = t; e < b * b; e += 256) {
    const int i = e % b, c = e / b;
    Rl[e] = i <= c ? (TO)(Bv[i * 65 + c] * dsc[i]) : TO(0);
  }
}
// The b x b Ritz problem (b <= 64): cyclic two-sided Jacobi with the round-robin ordering -- b/2 disjoint rotations per step,
// first from the right (columns of H and of the accumulated V), then from the left (rows of H) -- one workgroup per matrix,
// H and V in LDS.  Sweeps until the off-diagonal mass is below 1e-30 of ||H||_F^2 (15 at most).  Eigenvalues ascending in W,
// eigenvectors in the columns of S (column-major, leading dimension b; S may be H itself).  rocSOLVER's syevj takes 2.5 ms for
// 512 problems of 48 x 48, most of it launches; this kernel about a fifth.  (Round 4: the two passes of a step as one pass over
// 2 x 2 blocks, loads staged in front of the stores -- the same operations in the same order, the same bits: 1.5 -> 0.9 ms for 512
// problems of 56 x 56, 5-7 sweeps.)
template <typename TO>
__global__ __launch_bounds__(256) void k_ritz_jacobi(int b, int batch, const double* H_in, TO* S, double* __restrict__ W,
                                                     rocblas_int* __restrict__ info, rocblas_int* __restrict__ nsweeps = nullptr) {
  __shared__ double H[64 * 65];
  __shared__ double V[64 * 65];
  __shared__ double rc[32], rs[32];
  __shared__ int rp[32], rq[32];
  __shared__ double red[4];
  __shared__ double s_off, s_fro;
  const int l = blockIdx.x, t = threadIdx.x;
  const double* Hl = H_in + (long long)l * b * b;
  const int n = b + (b & 1), half = n / 2;
  for (int e = t; e < n * n; e += 256) {
    const int i = e % n, c = e / n;
    // the upper triangle is what the GEMM before filled reliably symmetric to rounding: mirror it (an odd b is padded by a
    // row and a column of zeros: the rotations that involve the pad are the identity, its entries stay zero)
    H[i * 65 + c] = (i < b && c < b) ? (i <= c ? Hl[(long long)c * b + i] : Hl[(long long)i * b + c]) : 0.0;
    V[i * 65 + c] = i == c ? 1.0 : 0.0;
  }
  auto block_sum = [&](double v) -> double {
    v = wave_sum(v);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  __syncthreads();
  {
    double f = 0;
    for (int e = t; e < b * b; e += 256) { const double h = H[(e % b) * 65 + e / b]; f += h * h; }
    f = block_sum(f);
    if (t == 0) s_fro = f;
  }
  int sweeps = 0;
  bool done = false;
  for (; sweeps < 15 && !done; ++sweeps) {
    for (int step = 0; step < n - 1; ++step) {
      __syncthreads();
      if (t < half) {
        int p, q;
        if (t == 0) { p = n - 1; q = step; }
        else {                                 // (step + t) and (step - t) modulo n - 1; t < n - 1
          p = step + t; if (p >= n - 1) p -= n - 1;
          q = step - t; if (q < 0) q += n - 1;
        }
        if (p > q) { const int x = p; p = q; q = x; }
        double c = 1.0, sn = 0.0;
        if (q < b) {
          const double hpq = H[p * 65 + q];
          if (hpq != 0.0) {
            const double tau = (H[q * 65 + q] - H[p * 65 + p]) / (2.0 * hpq);
            const double tt = tau >= 0 ? 1.0 / (tau + sqrt(1.0 + tau * tau)) : 1.0 / (tau - sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + tt * tt);
            sn = tt * c;
          }
        }
        rp[t] = p; rq[t] = q; rc[t] = c; rs[t] = sn;
      }
      __syncthreads();
      // H <- J' (H J), one thread per 2 x 2 block (row pair i, column pair j): the column rotation of the block's two rows, then
      // the row rotation of its two columns -- the operations of a column pass followed by a row pass, in their order, without
      // the barrier between the passes and with half the LDS traffic.  Operands of all of a thread's blocks are loaded before
      // any is stored (the blocks are disjoint, which the compiler cannot know: it would wait for every store).
      {
        double h00[4], h01[4], h10[4], h11[4], c1[4], s1[4], c2[4], s2[4];
        int a00[4], a01[4], a10[4], a11[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int it = t + 256 * k;
          if (it < half * half) {
            const int i = it / half, j = it - i * half;
            const int p1 = rp[i], q1 = rq[i], p2 = rp[j], q2 = rq[j];
            c1[k] = rc[i]; s1[k] = rs[i]; c2[k] = rc[j]; s2[k] = rs[j];
            a00[k] = p1 * 65 + p2; a01[k] = p1 * 65 + q2; a10[k] = q1 * 65 + p2; a11[k] = q1 * 65 + q2;
            h00[k] = H[a00[k]]; h01[k] = H[a01[k]]; h10[k] = H[a10[k]]; h11[k] = H[a11[k]];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int it = t + 256 * k;
          if (it < half * half) {
            const double r0 = c2[k] * h00[k] - s2[k] * h01[k], r1 = s2[k] * h00[k] + c2[k] * h01[k];       // row p1 after H J
            const double u0 = c2[k] * h10[k] - s2[k] * h11[k], u1 = s2[k] * h10[k] + c2[k] * h11[k];       // row q1 after H J
            H[a00[k]] = c1[k] * r0 - s1[k] * u0;
            H[a10[k]] = s1[k] * r0 + c1[k] * u0;
            H[a01[k]] = c1[k] * r1 - s1[k] * u1;
            H[a11[k]] = s1[k] * r1 + c1[k] * u1;
          }
        }
      }
      {                                                     // V <- V J, staged the same way
        double vp[8], vq[8], cc[8], ss[8];
        int ap[8], aq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int it = t + 256 * k;
          ap[k] = -1;
          if (it < half * b) {
            const int i = it / b, e = it - i * b;
            const int p = rp[i], q = rq[i];
            if (q < b) {
              cc[k] = rc[i]; ss[k] = rs[i];
              ap[k] = e * 65 + p; aq[k] = e * 65 + q;
              vp[k] = V[ap[k]]; vq[k] = V[aq[k]];
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (ap[k] >= 0) {
            V[ap[k]] = cc[k] * vp[k] - ss[k] * vq[k];
            V[aq[k]] = ss[k] * vp[k] + cc[k] * vq[k];
          }
      }
    }
    __syncthreads();
    double off = 0;
    for (int e = t; e < b * b; e += 256) {
      const int i = e % b, c = e / b;
      if (i != c) { const double h = H[i * 65 + c]; off += h * h; }
    }
    off = block_sum(off);
    if (t == 0) s_off = off;
    __syncthreads();
    done = !(s_off > 1e-30 * s_fro);
  }
  __syncthreads();
  if (t == 0) info[l] = (done && s_fro == s_fro) ? 0 : 1;
  if (t == 0 && nsweeps) nsweeps[l] = sweeps;
  if (t < b) {                                               // ascending order: the rank of every eigenvalue
    const double mine = H[t * 65 + t];
    int pos = 0;
    for (int j = 0; j < b; ++j) {
      const double o = H[j * 65 + j];
      pos += (o < mine || (o == mine && j < t)) ? 1 : 0;
    }
    W[(long long)l * b + pos] = mine;
    // column t of V becomes column pos of S
    TO* Sl = S + (long long)l * b * b + (long long)pos * b;
    for (int i = 0; i < b; ++i) Sl[i] = (TO)V[i * 65 + t];
  }
}
// (Round 5, measured and dropped: the same solver with ONE WAVE per matrix -- no workgroup barriers, three workgroups per CU, the 2 x 2
//  blocks of a step walked four per lane -- gives the same bits and takes TWICE as long: 512 problems of 56 x 56 in a C4 iteration
//  59.9 against 49.8 ms per iteration, a 64-slice share 25.4 against 12.3 ms.  The step's LDS round trips are latency, and 64 lanes
//  queue twelve of them behind each other where 256 threads queue four.)
// What the host needs to choose the next filter: res[2] <- min over the batch of t_r = 2 theta_r / a - 1 (bit pattern of a
// positive double, start from +inf), res[3] <- max over the batch of the number of Ritz vectors far above the lowest column
// -- both over the matrices whose residual (per_matrix, k_sub_residual) is still above tol: the others are not what the next
// filter is for (res[3]); res[5] <- the count of far-above vectors over ALL matrices (a filter that runs on the whole batch must
// project for the converged ones too: their vectors would lose what they have to a direction 1e5 times larger)
__global__ void k_cheb_plan(int b, int g, int r, int batch, const double* __restrict__ W, const double* __restrict__ per_matrix, double tol,
                            unsigned long long* res, const int* __restrict__ nd = nullptr) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= batch) return;
  const double* Wl = W + (long long)l * b;
  if (nd) r -= nd[l];
  if (r <= 0) return;
  const double a = cheb_floor(Wl, b, g, r);
  int nl = 0;
  for (int i = 0; i < b; ++i) nl += Wl[i] > CHEB_KAPPA * a ? 1 : 0;
  atomicMax(res + 5, (unsigned long long)nl);
  if (!(per_matrix[l] > tol)) return;
  double t = 2.0 * Wl[b - r] / a - 1.0;
  if (!(t > 1.0)) t = 1.0;
  atomicMin(res + 2, (unsigned long long)__double_as_longlong(t));
  atomicMax(res + 3, (unsigned long long)nl);
}
// The matrices whose residual is still above tol, in ascending order -> idx, their number -> res[4] (one workgroup).
__global__ __launch_bounds__(256) void k_sub_list(int batch, const double* __restrict__ per_matrix, double tol, int* __restrict__ idx,
                                                  unsigned long long* res) {
  __shared__ int cnt[256];
  const int t = threadIdx.x;
  const int chunk = (batch + 255) / 256, lo = t * chunk, hi = lo + chunk < batch ? lo + chunk : batch;
  int c = 0;
  for (int l = lo; l < hi; ++l) c += per_matrix[l] > tol ? 1 : 0;
  cnt[t] = c;
  __syncthreads();
  int base = 0;
  for (int j = 0; j < t; ++j) base += cnt[j];
  for (int l = lo; l < hi; ++l)
    if (per_matrix[l] > tol) idx[base++] = l;
  if (t == 255) res[4] = (unsigned long long)base;
}
// dst[j] <- src[idx[j]] (gather) or dst[idx[j]] <- src[j] (scatter), `per` values per matrix
template <typename TS>
__global__ __launch_bounds__(BLOCK) void k_sub_move(long long per, int n, const int* __restrict__ idx, const TS* __restrict__ src,
                                                    TS* __restrict__ dst, int scatter) {
  const long long total = per * n;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long j = e / per, o = e - j * per;
    if (scatter) dst[(long long)idx[j] * per + o] = src[e];
    else dst[e] = src[(long long)idx[j] * per + o];
  }
}
// Inertia certificate for a converged top-r block on a spectrum too flat for the energy bound of k_sub_residual:
// B = mu I - G + X_r Theta_r X_r' is positive definite (its Cholesky factorisation exists) exactly when G, with the r found
// pairs removed, has no eigenvalue above mu; mu = the middle of the gap between the r-th and the (r+1)-th Ritz value.
// First half: B <- mu I - G and XT <- X_r Theta_r (the rank-r term is added by a GEMM).
// (low = 1, SIPX_RANK_CERT_CHECK only: mu = half the (r+1)-th Ritz value, below an eigenvalue B must then have -- never definite)
__global__ __launch_bounds__(BLOCK) void k_cert_shift(int k, int b, int r, int batch, const double* __restrict__ G, const double* __restrict__ W,
                                                      double* __restrict__ B, int low = 0) {
  const long long per = (long long)k * k, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per, o = e - l * per;
    const int row = (int)(o % k), col = (int)(o / k);
    const double mu = low ? 0.5 * W[l * b + b - r - 1] : 0.5 * (W[l * b + b - r] + W[l * b + b - r - 1]);
    B[e] = (row == col ? mu : 0.0) - G[e];
  }
}
__global__ __launch_bounds__(BLOCK) void k_cert_scale(int k, int b, int r, int batch, const double* __restrict__ X, const double* __restrict__ W,
                                                      double* __restrict__ XT) {
  const long long per = (long long)k * r, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per, o = e - l * per;
    const int j = b - r + (int)(o / k);
    XT[l * (long long)k * b + (long long)(b - r) * k + o] = X[l * (long long)k * b + (long long)(b - r) * k + o] * W[l * b + j];
  }
}
// Blocked Cholesky for that certificate (only its success matters): the bs x bs diagonal block at (J, J) of every matrix --
// upper triangle, column-major, leading dimension k -- is factored in LDS, D = U'U, and the inverse of U goes to Winv (64 x 64 per
// matrix, leading dimension 64), so that the block row of the factor is one GEMM, U_J,rest = Winv' B_J,rest, and the trailing
// matrix takes one more, B_rest,rest -= U_J,rest' U_J,rest (rank_cert_factor below).  A pivot that is not positive raises info[l]
// and leaves the identity in Winv: the matrix is not positive definite, whatever the later steps make of it.
// (rocSOLVER's potrf_strided_batched: 7.3 ms for 512 matrices of 512 x 512, a fifth of the call.)
__global__ __launch_bounds__(256) void k_chol_diag(int k, int J, int bs, const double* __restrict__ B, long long sB, double* __restrict__ Winv,
                                                   rocblas_int* __restrict__ info) {
  __shared__ double A[64 * 65];
  __shared__ double Bv[64 * 65];
  const int l = blockIdx.x, t = threadIdx.x;
  const double* Bl = B + (long long)l * sB + (long long)J * k + J;
  double* Wl = Winv + (long long)l * 4096;
  for (int e = t; e < bs * bs; e += 256) {
    const int i = e % bs, c = e / bs;
    if (i <= c) A[i * 65 + c] = Bl[(long long)c * k + i];
    Bv[i * 65 + c] = 0.0;
  }
  bool bad = false;
  for (int j = 0; j < bs; ++j) {
    __syncthreads();
    const double piv = A[j * 65 + j];
    if (!(piv > 0.0)) { bad = true; break; }               // the same value in every thread: a uniform exit (NaN lands here too)
    const double inv = 1.0 / sqrt(piv);
    __syncthreads();
    if (t == 0) A[j * 65 + j] = sqrt(piv);
    for (int c = j + 1 + t; c < bs; c += 256) A[j * 65 + c] *= inv;
    __syncthreads();
    const int nrem = bs - j - 1;
    for (int e = t; e < nrem * nrem; e += 256) {
      const int ii = e / nrem, cc = e - ii * nrem;
      if (cc >= ii) A[(j + 1 + ii) * 65 + j + 1 + cc] -= A[j * 65 + j + 1 + ii] * A[j * 65 + j + 1 + cc];
    }
  }
  __syncthreads();
  if (bad) {
    if (t == 0) info[l] = 1;
    for (int e = t; e < 4096; e += 256) Wl[e] = (e % 64 == e / 64) ? 1.0 : 0.0;
    return;
  }
  if (t < bs) {                                              // column t of the inverse of U, back substitution
    const int c = t;
    Bv[c * 65 + c] = 1.0 / A[c * 65 + c];
    for (int i = c - 1; i >= 0; --i) {
      double acc = 0;
      for (int q = i + 1; q <= c; ++q) acc += A[i * 65 + q] * Bv[q * 65 + c];
      Bv[i * 65 + c] = -acc / A[i * 65 + i];
    }
  }
  __syncthreads();
  for (int e = t; e < 4096; e += 256) {
    const int i = e % 64, c = e / 64;
    Wl[e] = (i <= c && c < bs) ? Bv[i * 65 + c] : 0.0;
  }
}
__global__ void k_cert_or(int batch, const rocblas_int* __restrict__ info, unsigned long long* res) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < batch && info[l] != 0) atomicOr(res + 1, 32ull);
}
// After a full decomposition (eigenvalues ascending, k per matrix): the contraction factor subspace iteration on b vectors
// would see for the top-r space, theta_{b+1} / theta_r, maximum over the batch -> res[0] (bit pattern).
__global__ void k_sub_ratio(int k, int b, int r, int batch, const double* __restrict__ W, unsigned long long* res) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= batch) return;
  const double tr = W[(long long)l * k + k - r], tb = W[(long long)l * k + k - b - 1];
  double q = (tr > 0 && tb >= 0) ? tb / tr : 1.0;
  if (!(q == q) || q > 1.0) q = 1.0;
  atomicMax(res, (unsigned long long)__double_as_longlong(q));
}
// The last b eigenvector columns of the full decomposition (k x k per matrix) become the next warm start -- unless the
// factorisation of that matrix did not converge (info != 0): its previous warm start stays.
__global__ __launch_bounds__(BLOCK) void k_sub_keep(int k, int b, int batch, const double* __restrict__ E, double* __restrict__ X,
                                                    const rocblas_int* __restrict__ info) {
  const long long per = (long long)k * b, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per, o = e - l * per;
    if (info[l] != 0) continue;
    X[e] = E[l * (long long)k * k + (long long)(k - b) * k + o];
  }
}
// status words of a batched factorisation OR-ed into one flag (read by the host at the NEXT call of the projector: a
// projection built on a factorisation that did not converge must not go unnoticed)
// resid / smax given (one-sided Jacobi SVD): info = 1 there only says that the sweeps stopped short of the requested
// (machine-precision) tolerance; it counts as a failure when the off-diagonal mass it reports is not negligible against the
// largest singular value squared, or anything is not finite.
__global__ void k_info_or(int batch, const rocblas_int* __restrict__ info, int* __restrict__ fail, const double* __restrict__ resid,
                          const double* __restrict__ S, int k) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= batch || info[l] == 0) return;
  if (resid) {
    const double s0 = S[(long long)l * k], r = resid[l];
    if (r == r && s0 == s0 && r <= 1e-9 * s0 * s0) return;
  }
  atomicOr(fail, 1);
}

// ------------------------------------------------------------------------------------------------
template <typename T>
struct ExtImpl {
  ExtSpec sp;
  SegMap map{};
  hipStream_t stream = nullptr;
  std::vector<void*> owned;       // device allocations
  // DFT
  hipfftHandle plan = 0;
  bool have_plan = false;
  hipfftHandle plan_r2c = 0, plan_c2r = 0;        // l1 behind the DFT through the real transform (half the spectrum)
  bool real_fft = false;
  int nh1 = 0, ndup = 0;
  long long Nh = 0;
  Cplx<T>* z = nullptr;
  T* mag = nullptr;
  ProjScalars<T>*ps = nullptr, *psf = nullptr;
  T radius_raw = 0;
  // rank / nuclear
  rocblas_handle blas = nullptr;
  int r = 0, m = 0, n = 0, batch = 1;
  double *Ad = nullptr, *Ud = nullptr, *Sd = nullptr, *Vd = nullptr, *Ed = nullptr;
  double *Gd = nullptr, *Gs = nullptr, *Wd = nullptr;     // Gram route: eigenvectors, scaled copy, eigenvalues
  bool gram = false;
  // rank, Gram route: warm-started block subspace iteration (top-r invariant subspace of the Gram matrices)
  int sub_b = 0;                                   // block size r + 16 (0 = route not used)
  double *Xs[2] = {nullptr, nullptr};              // Ritz vectors of the previous call (y update / feasibility estimate)
  bool sub_have[2] = {false, false}, sub_try[2] = {false, false};
  double *Qs = nullptr, *Zs = nullptr, *Hs = nullptr, *Ws = nullptr, *Es = nullptr, *Fro = nullptr, *FroPart = nullptr;
  // Chebyshev-filtered variant of the same route (spectra without a gap behind the block): one more block, the matrix of the
  // inertia certificate, calls to sit out after a failed attempt
  bool cheb = false;
  double *Ys = nullptr, *Bd = nullptr, *Cs = nullptr;
  // the matrices that still need a filter when most of the batch has converged, packed (rank_cheb_route)
  // switches of the rank routes, read once when the projector is built (tests and A/B runs set them before they build a context)
  struct RankKnobs {
    int dbg = 0;                  // SIPX_EXT_DEBUG: 1 the route of every call on stderr, 2 milliseconds per phase, 3 open matrices and Jacobi sweeps per step
    int budget = 160;             // SIPX_RANK_CHEB_BUDGET: multiplications with G a call may spend
    bool own_jacobi = true;       // SIPX_RANK_JACOBI=0: rocSOLVER's syevj for the Ritz problems
    bool fused_proj = true;       // SIPX_RANK_CHEB_FUSED=0: the GEMM form of the projections whatever their number
    int m_cap = 16;               // SIPX_RANK_CHEB_MMAX: degree of one filter at most
    double tol = 1e-12;           // SIPX_RANK_CHEB_TOL (experiments: is a difference between the routes a matter of this tolerance?)
    int guard = -1;               // SIPX_RANK_CHEB_GUARD: index of the Ritz value that ends the damped interval (-1: chosen from the block)
    bool own_cert = true;         // SIPX_RANK_CERT_POTRF=1: the library's factorisation for the certificate
    bool pack = true;             // SIPX_RANK_PACK=0: every filter on the whole batch
    bool cert_check = false;      // SIPX_RANK_CERT_CHECK: both factorisations, compared matrix by matrix (tests)
    // Round 5.  eps_bw: the acceptance level of a Ritz pair as a backward error on the slice itself, ||E||_2 <= eps_bw ||X||_2
    // (k_sub_residual) -- the class of the reference's svd() in TF (project_rank!.jl:28-41).  2^-23 = eps(Float32): LAPACK's
    // sgesdd, the routine behind Julia's svd, leaves 1.4e-7 (median) to 5.9e-7 (largest) by the same measure on a 512 x 512 slice
    // of the C4 model (tests/test_svd_class.py::test_float32_svd_backward_error_class); with 2^-21 the projected slices were 2.5e-7
    // of their norm from the exact projection where sgesdd's are 2e-8 .. 1.5e-7 (tests/test_gpu_round5.py), hence the tighter level --
    // two filter degrees more per call.  0 = the strict level tol theta_max of rounds 3-4 (SIPX_RANK_STRICT=1; Float64 models never
    // come here: they keep the one-sided Jacobi SVD).
    double eps_bw = 1.1920928955078125e-07;
    bool cold = true;             // SIPX_RANK_COLD=0: a call without a usable start decomposes fully (rounds 3-4)
    int window = 0;               // SIPX_RANK_WINDOW: guard pairs below the r-th that are held to the acceptance level as well
    bool f32 = false;             // SIPX_RANK_F32=1 (measured, NOT the default): the filters of a warm call in Float32 on deflated matrices
    bool keep_damped = false;     // SIPX_RANK_KEEP=1 (measured, NOT the default): every column inside the damped interval stays out of the filter
  } knobs;
  double *Xc = nullptr, *Wc = nullptr, *Froc = nullptr;
  // the Float32 loop on the deflated matrices (round 5): its own matrices and blocks, the Ritz values of the deflated problem,
  // deflated pairs per matrix and the largest eigenvalue of every matrix as it was
  float *G32 = nullptr, *Gp32 = nullptr, *X32 = nullptr, *A32 = nullptr, *F32a = nullptr, *F32b = nullptr, *Xc32 = nullptr;
  float *Cs32 = nullptr, *Zs32 = nullptr;
  double *W32 = nullptr, *tmax32 = nullptr, *tmax32c = nullptr;
  double *Xtop = nullptr, *Ytop = nullptr, *theta32 = nullptr;      // the refined top block (k x DEFL_N per matrix), its Rayleigh quotients
  double* Wprev[2] = {nullptr, nullptr};                            // Ritz values the last accepted call of each state ended with
  bool have_wprev[2] = {false, false};
  int *nd32 = nullptr, *nd32c = nullptr;
  long long n_f32 = 0, n_f32_back = 0;
  double *cert_w = nullptr, *cert_p = nullptr;     // the certificate's blocked Cholesky: inverse diagonal factors, one block row
  int* sub_idx = nullptr;
  int sub_cap = 0;
  long long n_packed = 0;
  int cheb_skip[2] = {0, 0}, cheb_fails[2] = {0, 0};
  bool cert_warm = false;
  long long n_calls = 0, n_subspace = 0, n_full = 0, n_products = 0;       // route_counts()
  unsigned long long* sub_res = nullptr;           // device: bit pattern of the largest relative residual, failure flag
  unsigned long long* sub_res_host = nullptr;      // pinned
  // DCT: orthonormal DCT-II matrices per dimension, two work arrays, inner projector state
  T* Cm[3] = {nullptr, nullptr, nullptr};
  T *W1 = nullptr, *W2 = nullptr, *dlb = nullptr, *dub = nullptr;
  long long* cidx = nullptr;
  rocblas_int* info = nullptr;
  int* flag = nullptr;
  // status of the batched factorisations: OR of info != 0, copied to pinned memory behind the call, looked at by the next one
  int* fail = nullptr;
  int* fail_host = nullptr;
  hipEvent_t fail_ev = nullptr;
  bool fail_pending = false;
  void note_status(hipStream_t s, const double* resid = nullptr, const double* S = nullptr, int k = 0) {
    if (!fail) {
      SIPX_HIP(hipMalloc((void**)&fail, sizeof(int)));
      owned.push_back(fail);
      SIPX_HIP(hipHostMalloc((void**)&fail_host, sizeof(int), hipHostMallocDefault));
      SIPX_HIP(hipEventCreateWithFlags(&fail_ev, hipEventDisableTiming));
    }
    SIPX_HIP(hipMemsetAsync(fail, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_info_or, dim3((batch + 63) / 64), dim3(64), 0, s, batch, info, fail, resid, S, k);
    SIPX_HIP(hipMemcpyAsync(fail_host, fail, sizeof(int), hipMemcpyDeviceToHost, s));
    SIPX_HIP(hipEventRecord(fail_ev, s));
    fail_pending = true;
  }
  void check_status() {
    if (!fail_pending) return;
    fail_pending = false;
    SIPX_HIP(hipEventSynchronize(fail_ev));
    if (*fail_host != 0)
      throw std::runtime_error("rank / nuclear projector: the batched eigen / singular value decomposition did not converge on some slice "
                               "(rocSOLVER info != 0) in the previous call");
  }
  // histogram
  T *keys_in = nullptr, *keys_out = nullptr, *lb = nullptr, *ub = nullptr;
  unsigned int *idx_in = nullptr, *idx_out = nullptr;
  void* sort_tmp = nullptr;
  size_t sort_bytes = 0;
  // subspace
  T *basis = nullptr, *gram_inv = nullptr, *X = nullptr, *t1 = nullptr, *t2 = nullptr;
  int cols = 0;

  template <typename Q>
  Q* alloc(size_t count) {
    void* p = nullptr;
    SIPX_HIP(hipMalloc(&p, sizeof(Q) * (count ? count : 1)));
    if (long long* t = alloc_tally()) *t += (long long)(sizeof(Q) * (count ? count : 1));
    owned.push_back(p);
    return (Q*)p;
  }
};

static void fft_check(hipfftResult r, const char* what) {
  if (r != HIPFFT_SUCCESS) throw std::runtime_error(std::string("hipFFT: ") + what + " failed (" + std::to_string((int)r) + ")");
}
static void blas_check(rocblas_status r, const char* what) {
  if (r != rocblas_status_success) throw std::runtime_error(std::string("rocBLAS/rocSOLVER: ") + what + " failed (" + std::to_string((int)r) + ")");
}

// inverse of the r x r Gram matrix A'A in float64 (Gauss-Jordan, partial pivoting); A is L x r column-major
template <typename T>
static std::vector<double> gram_inverse(const T* A, long long L, int r) {
  std::vector<double> G((size_t)r * r, 0.0), I((size_t)r * r, 0.0);
  for (int i = 0; i < r; ++i)
    for (int j = i; j < r; ++j) {
      double acc = 0;
      const T *ai = A + (size_t)i * L, *aj = A + (size_t)j * L;
      for (long long k = 0; k < L; ++k) acc += (double)ai[k] * (double)aj[k];
      G[(size_t)i * r + j] = G[(size_t)j * r + i] = acc;
    }
  for (int i = 0; i < r; ++i) I[(size_t)i * r + i] = 1.0;
  for (int c = 0; c < r; ++c) {
    int piv = c;
    for (int i = c + 1; i < r; ++i)
      if (std::fabs(G[(size_t)i * r + c]) > std::fabs(G[(size_t)piv * r + c])) piv = i;
    if (G[(size_t)piv * r + c] == 0.0) throw std::runtime_error("subspace: the columns of A are linearly dependent");
    if (piv != c)
      for (int j = 0; j < r; ++j) {
        std::swap(G[(size_t)piv * r + j], G[(size_t)c * r + j]);
        std::swap(I[(size_t)piv * r + j], I[(size_t)c * r + j]);
      }
    const double d = 1.0 / G[(size_t)c * r + c];
    for (int j = 0; j < r; ++j) { G[(size_t)c * r + j] *= d; I[(size_t)c * r + j] *= d; }
    for (int i = 0; i < r; ++i) {
      if (i == c) continue;
      const double f = G[(size_t)i * r + c];
      if (f == 0.0) continue;
      for (int j = 0; j < r; ++j) { G[(size_t)i * r + j] -= f * G[(size_t)c * r + j]; I[(size_t)i * r + j] -= f * I[(size_t)c * r + j]; }
    }
  }
  return I;     // symmetric: row- and column-major coincide up to rounding; used as column-major below
}

template <typename T>
ExtProj<T>::ExtProj(const ExtSpec& spec, hipStream_t stream) {
  impl_ = new ExtImpl<T>();
  ExtImpl<T>& I = *impl_;
  try {
  I.sp = spec;
  I.stream = stream;
  const Grid& G = spec.G;
  const long long N = G.N;
  const int kind = spec.kind;
  if (kind == EXT_DFT_MASK) {
    if (!spec.ub) throw std::runtime_error("bounds in the DFT domain need the mask vector (constraint.max)");
    const hipfftType ty = sizeof(T) == 4 ? HIPFFT_C2C : HIPFFT_Z2Z;
    if (spec.ndim == 2) fft_check(hipfftPlan2d(&I.plan, (int)G.n[1], (int)G.n[0], ty), "plan2d");
    else fft_check(hipfftPlan3d(&I.plan, (int)G.n[2], (int)G.n[1], (int)G.n[0], ty), "plan3d");
    I.have_plan = true;
    fft_check(hipfftSetStream(I.plan, stream), "set stream");
    I.z = I.template alloc<Cplx<T>>(N);
    I.mag = I.template alloc<T>(N);                      // the mask
    SIPX_HIP(hipMemcpy(I.mag, spec.ub, sizeof(T) * N, hipMemcpyHostToDevice));
  } else if (kind == EXT_L1_DFT) {
    if (!(spec.pmax > 0)) throw std::runtime_error("Radius of L1 ball is negative");
    const char* rf_e = getenv("SIPX_DFT_REAL");            // 0: the complex transform of the packed model (A/B switch, tests)
    I.real_fft = !(rf_e && rf_e[0] == '0') && G.n[0] >= 4;
    if (I.real_fft) {
      const hipfftType tf = sizeof(T) == 4 ? HIPFFT_R2C : HIPFFT_D2Z, tb = sizeof(T) == 4 ? HIPFFT_C2R : HIPFFT_Z2D;
      if (spec.ndim == 2) {
        fft_check(hipfftPlan2d(&I.plan_r2c, (int)G.n[1], (int)G.n[0], tf), "plan2d (real)");
        fft_check(hipfftPlan2d(&I.plan_c2r, (int)G.n[1], (int)G.n[0], tb), "plan2d (real, inverse)");
      } else {
        fft_check(hipfftPlan3d(&I.plan_r2c, (int)G.n[2], (int)G.n[1], (int)G.n[0], tf), "plan3d (real)");
        fft_check(hipfftPlan3d(&I.plan_c2r, (int)G.n[2], (int)G.n[1], (int)G.n[0], tb), "plan3d (real, inverse)");
      }
      fft_check(hipfftSetStream(I.plan_r2c, stream), "set stream");
      fft_check(hipfftSetStream(I.plan_c2r, stream), "set stream");
      I.nh1 = (int)(G.n[0] / 2 + 1);
      I.ndup = (int)G.n[0] - I.nh1;
      I.Nh = (long long)I.nh1 * (N / G.n[0]);
      I.z = I.template alloc<Cplx<T>>(I.Nh);
    } else {
      const hipfftType ty = sizeof(T) == 4 ? HIPFFT_C2C : HIPFFT_Z2Z;
      if (spec.ndim == 2) fft_check(hipfftPlan2d(&I.plan, (int)G.n[1], (int)G.n[0], ty), "plan2d");   // slowest dimension first
      else fft_check(hipfftPlan3d(&I.plan, (int)G.n[2], (int)G.n[1], (int)G.n[0], ty), "plan3d");
      I.have_plan = true;
      fft_check(hipfftSetStream(I.plan, stream), "set stream");
      I.z = I.template alloc<Cplx<T>>(N);
    }
    I.mag = I.template alloc<T>(N);
    I.ps = I.template alloc<ProjScalars<T>>(1);
    I.psf = I.template alloc<ProjScalars<T>>(1);
    K<T>::ps_init(stream, I.ps, nullptr);
    K<T>::ps_init(stream, I.psf, nullptr);
    I.radius_raw = (T)(spec.pmax * sqrt((double)N));       // ||F_unitary v||_1 <= b  <=>  ||FFT v||_1 <= b sqrt(N)
  } else if (kind == EXT_RANK || kind == EXT_NUCLEAR) {
    ExtSpec sp = spec;
    if (sp.mode == SIPX_MODE_WHOLE) {                      // a matrix: one "slice" orthogonal to the unit third dimension
      if (sp.dims[2] != 1)
        throw std::runtime_error("requested rank or nuclear norm constraints on a tensor, use mode=(slice,x) e.t.c. to "
                                 "define constraints per slice");                       // setup_constraints.jl:60-62
      sp.mode = SIPX_MODE_SLICE;
      sp.dir = 2;
    } else if (sp.mode != SIPX_MODE_SLICE) {
      throw std::runtime_error("mode[1] for rank / nuclear norm projections can only be: slice");
    }
    I.map = make_segmap(sp);
    I.m = (int)I.map.LA; I.n = (int)I.map.LB; I.batch = (int)I.map.nseg;
    const int k = I.m < I.n ? I.m : I.n;
    if (kind == EXT_RANK) {
      I.r = (int)spec.pmax;
      if (I.r < 1) throw std::runtime_error("rank constraint needs r >= 1");
      if (I.r > k) I.r = k;                                // U[:,1:r] with r = min(n1,n2): the projection is the identity
    } else if (!(spec.pmax > 0)) {
      throw std::runtime_error("Radius of L1 ball is negative");                        // project_l1_Duchi!.jl:22 on F.S
    }
    blas_check(rocblas_create_handle(&I.blas), "create handle");
    blas_check(rocblas_set_stream(I.blas, stream), "set stream");
    // Float32 models take the Gram route (eigenvectors of the smaller of X'X and XX', 12-24x faster than the Jacobi SVD
    // on 256..512-sized slices); its error eps64 * cond^2 stays far below Float32 resolution.  Float64 models keep the
    // one-sided Jacobi SVD, which works on the columns of X itself.
    I.gram = sizeof(T) == 4;
    I.Ad = I.template alloc<double>((size_t)I.m * I.n * I.batch);
    I.Ud = I.template alloc<double>((size_t)I.m * k * I.batch);
    I.Vd = I.template alloc<double>((size_t)k * I.n * I.batch);
    I.Sd = I.template alloc<double>((size_t)k * I.batch);
    I.Ed = I.template alloc<double>((size_t)(I.gram ? k : 1) * I.batch);
    if (I.gram) {
      I.Gd = I.template alloc<double>((size_t)k * k * I.batch);
      I.Wd = I.template alloc<double>((size_t)k * I.batch);
      if (kind == EXT_NUCLEAR) I.Gs = I.template alloc<double>((size_t)k * k * I.batch);
      {
        auto env = [](const char* n) { return getenv(n); };
        auto& K_ = I.knobs;
        if (const char* e = env("SIPX_EXT_DEBUG")) K_.dbg = atoi(e);
        if (const char* e = env("SIPX_RANK_CHEB_BUDGET")) K_.budget = atoi(e) > 0 ? atoi(e) : K_.budget;
        if (const char* e = env("SIPX_RANK_JACOBI")) K_.own_jacobi = e[0] != '0';
        if (const char* e = env("SIPX_RANK_CHEB_FUSED")) K_.fused_proj = e[0] != '0';
        if (const char* e = env("SIPX_RANK_CHEB_MMAX")) K_.m_cap = atoi(e) >= 2 ? atoi(e) : K_.m_cap;
        if (const char* e = env("SIPX_RANK_CHEB_TOL")) K_.tol = atof(e) > 0 ? atof(e) : K_.tol;
        if (const char* e = env("SIPX_RANK_CHEB_GUARD")) K_.guard = std::max(0, atoi(e));
        if (const char* e = env("SIPX_RANK_CERT_POTRF")) K_.own_cert = e[0] != '1';
        if (const char* e = env("SIPX_RANK_PACK")) K_.pack = e[0] != '0';
        K_.cert_check = env("SIPX_RANK_CERT_CHECK") != nullptr;
        if (const char* e = env("SIPX_RANK_STRICT")) { if (e[0] != '0') K_.eps_bw = 0.0; }
        if (const char* e = env("SIPX_RANK_EPS")) K_.eps_bw = atof(e) >= 0 ? atof(e) : K_.eps_bw;
        if (const char* e = env("SIPX_RANK_COLD")) K_.cold = e[0] != '0';
        if (const char* e = env("SIPX_RANK_WINDOW")) K_.window = std::max(0, atoi(e));
        if (const char* e = env("SIPX_RANK_F32")) K_.f32 = e[0] != '0';
        if (const char* e = env("SIPX_RANK_KEEP")) K_.keep_damped = e[0] != '0';
        if (const char* e = env("SIPX_RANK_FLOOR")) {
          const double f = atof(e);
          if (f > 0 && f < 1) SIPX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_cheb_floor_factor), &f, sizeof(double)));
        }
      }
      const char* sub_e = getenv("SIPX_RANK_SUBSPACE");      // read per projector: 0 keeps the full decomposition every call
      const int sub_env = sub_e ? atoi(sub_e) : 1;
      // columns the block holds beyond the r wanted ones: 24 where the matrices are large enough for the route with them, else 16.
      // (C4, 512 slices of 512 x 512, r = 32, round 4: 12 / 16 / 20 / 24 / 28 / 32 guards -> 15.7 / 15.4 / 16.3 / 16.5 / 16.4 / 16.1 it/s:
      //  more guards move the end of the damped interval away from theta_r, and the library's GEMM tiles are 32 columns wide --
      //  48 columns cost what 64 do.)  SIPX_RANK_GUARDS: experiments.
      const char* ex_e = getenv("SIPX_RANK_GUARDS");
      const int extra = ex_e && atoi(ex_e) >= 4 ? atoi(ex_e) : ((I.r + 24) * 4 <= k && I.r + 24 <= 64 ? 24 : 16);
      if (kind == EXT_RANK && sub_env && (I.r + extra) * 4 <= k && I.r + extra <= 64) {      // worth it only for r << k
        I.sub_b = I.r + extra;
        const size_t nb = (size_t)k * I.sub_b * I.batch;
        for (int w = 0; w < 2; ++w) I.Xs[w] = I.template alloc<double>(nb);
        I.Qs = I.template alloc<double>(nb);
        I.Zs = I.template alloc<double>(nb);
        I.Hs = I.template alloc<double>((size_t)I.sub_b * I.sub_b * I.batch);
        I.Ws = I.template alloc<double>((size_t)I.sub_b * I.batch);
        I.Es = I.template alloc<double>((size_t)I.sub_b * I.batch);
        I.Fro = I.template alloc<double>((size_t)I.batch);
        I.FroPart = I.template alloc<double>((size_t)I.batch * FRO_PARTS);
        I.sub_res = I.template alloc<unsigned long long>(8);
        SIPX_HIP(hipHostMalloc((void**)&I.sub_res_host, 8 * sizeof(unsigned long long), hipHostMallocDefault));
        const char* ch_e = getenv("SIPX_RANK_CHEB");           // 0: plain subspace iteration only (spectra with a gap)
        I.cheb = !(ch_e && ch_e[0] == '0');
        if (I.cheb) {
          I.Ys = I.template alloc<double>(nb);
          I.Cs = I.template alloc<double>((size_t)I.sub_b * I.sub_b * I.batch);
          I.Bd = I.template alloc<double>((size_t)k * k * I.batch);       // the matrix of the inertia certificate
          I.cert_w = I.template alloc<double>((size_t)4096 * I.batch);
          I.cert_p = I.template alloc<double>((size_t)64 * k * I.batch);
          I.sub_cap = I.batch / 4;
          if (I.sub_cap > 0) {
            I.Xc = I.template alloc<double>((size_t)k * I.sub_b * I.sub_cap);
            I.Wc = I.template alloc<double>((size_t)I.sub_b * I.sub_cap);
            I.Froc = I.template alloc<double>((size_t)I.sub_cap);
            I.sub_idx = I.template alloc<int>((size_t)I.batch);
          }
          if (I.knobs.f32 && I.knobs.eps_bw > 0) {           // the Float32 loop's own arrays
            const size_t nb32 = (size_t)k * I.sub_b * I.batch, bb = (size_t)I.sub_b * I.sub_b * I.batch;
            I.G32 = I.template alloc<float>((size_t)k * k * I.batch);
            I.X32 = I.template alloc<float>(nb32); I.A32 = I.template alloc<float>(nb32);
            I.F32a = I.template alloc<float>(nb32); I.F32b = I.template alloc<float>(nb32);
            I.Cs32 = I.template alloc<float>(bb); I.Zs32 = I.template alloc<float>(bb);
            I.W32 = I.template alloc<double>((size_t)I.sub_b * I.batch);
            I.tmax32 = I.template alloc<double>((size_t)I.batch); I.nd32 = I.template alloc<int>((size_t)I.batch);
            I.Xtop = I.template alloc<double>((size_t)k * 8 * I.batch); I.Ytop = I.template alloc<double>((size_t)k * 8 * I.batch);
            I.theta32 = I.template alloc<double>((size_t)8 * I.batch);
            for (int w2 = 0; w2 < 2; ++w2) I.Wprev[w2] = I.template alloc<double>((size_t)I.sub_b * I.batch);
            if (I.sub_cap > 0) {
              I.Gp32 = I.template alloc<float>((size_t)k * k * I.sub_cap);
              I.Xc32 = I.template alloc<float>((size_t)k * I.sub_b * I.sub_cap);
              I.tmax32c = I.template alloc<double>((size_t)I.sub_cap); I.nd32c = I.template alloc<int>((size_t)I.sub_cap);
            }
          }
        }
      }
    }
    I.info = I.template alloc<rocblas_int>((size_t)3 * I.batch);   // info, n_sweeps / second info, sweeps of the Ritz solver
    I.flag = I.template alloc<int>((size_t)I.batch);
  } else if (kind == EXT_DCT) {
    // Orthonormal DCT-II along every dimension as dense n_d x n_d matrices (built in float64, rounded to TF once):
    // C[k, i] = s_k cos(pi (2i+1) k / (2n)), s_0 = sqrt(1/n), s_k = sqrt(2/n).  joDCT's normalisation is not pinned by any
    // reference test; orthonormal is assumed (the operator declares AtA_diag, get_TD_operator.jl:49-51,84-86).
    const double PI = 3.14159265358979323846;
    for (int a = 0; a < spec.ndim; ++a) {
      const int n = (int)G.n[a];
      std::vector<T> C((size_t)n * n);
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k)
          C[(size_t)i * n + k] = (T)((k == 0 ? std::sqrt(1.0 / n) : std::sqrt(2.0 / n)) * std::cos(PI * (2.0 * i + 1.0) * k / (2.0 * n)));
      I.Cm[a] = I.template alloc<T>((size_t)n * n);      // column-major n x n: element (k, i) at k + n i
      SIPX_HIP(hipMemcpy(I.Cm[a], C.data(), sizeof(T) * C.size(), hipMemcpyHostToDevice));
    }
    I.W1 = I.template alloc<T>(N);
    I.W2 = I.template alloc<T>(N);
    blas_check(rocblas_create_handle(&I.blas), "create handle");
    blas_check(rocblas_set_stream(I.blas, stream), "set stream");
    const int in = spec.inner;
    if (in == SIPX_PROJ_L1 || in == SIPX_PROJ_CARDINALITY) {
      I.ps = I.template alloc<ProjScalars<T>>(1);
      I.psf = I.template alloc<ProjScalars<T>>(1);
      if (in == SIPX_PROJ_CARDINALITY) I.cidx = I.template alloc<long long>(N);
      K<T>::ps_init(stream, I.ps, I.cidx);
      K<T>::ps_init(stream, I.psf, I.cidx);
    }
    if (in == SIPX_PROJ_BOUNDS_VEC) {
      if (!spec.lb || !spec.ub) throw std::runtime_error("per-element bounds need lb and ub");
      I.dlb = I.template alloc<T>(N); I.dub = I.template alloc<T>(N);
      SIPX_HIP(hipMemcpy(I.dlb, spec.lb, sizeof(T) * N, hipMemcpyHostToDevice));
      SIPX_HIP(hipMemcpy(I.dub, spec.ub, sizeof(T) * N, hipMemcpyHostToDevice));
    }
  } else if (kind == EXT_CARD_SEG) {
    if (spec.mode != SIPX_MODE_FIBER && spec.mode != SIPX_MODE_SLICE)
      throw std::runtime_error("segmented cardinality needs a fiber or slice mode");
    if (spec.ndim == 2 && spec.mode != SIPX_MODE_FIBER)
      throw std::runtime_error("for 2D models, the mode of application for project_cardinality! needs to be (fiber,x) or "
                               "(fiber,z). Or, provide the model as a vector");       // project_cardinality!.jl:57
    if (spec.pmax < 0) throw std::runtime_error("cardinality must be non-negative");
    I.map = make_segmap(spec);
  } else if (kind == EXT_HISTOGRAM) {
    if (spec.mode != SIPX_MODE_WHOLE) throw std::runtime_error("histogram constraints act on the whole vector");
    if (!spec.lb || !spec.ub) throw std::runtime_error("histogram constraints need sorted lb and ub vectors");
    I.map = make_segmap(spec);
    const long long M = I.map.L;
    I.keys_in = I.template alloc<T>(M); I.keys_out = I.template alloc<T>(M);
    I.idx_in = I.template alloc<unsigned int>(M); I.idx_out = I.template alloc<unsigned int>(M);
    I.lb = I.template alloc<T>(M); I.ub = I.template alloc<T>(M);
    SIPX_HIP(hipMemcpy(I.lb, spec.lb, sizeof(T) * M, hipMemcpyHostToDevice));
    SIPX_HIP(hipMemcpy(I.ub, spec.ub, sizeof(T) * M, hipMemcpyHostToDevice));
    SIPX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, I.sort_bytes, I.keys_in, I.keys_out, I.idx_in, I.idx_out, (int)M, 0,
                                                (int)sizeof(T) * 8, stream));
    I.sort_tmp = I.template alloc<char>(I.sort_bytes);
  } else if (kind == EXT_SUBSPACE) {
    ExtSpec sp = spec;
    if (spec.ndim == 2 && spec.mode == SIPX_MODE_SLICE) throw std::runtime_error("mode[1] for project_subspace! must be: fiber");
    if (spec.ndim == 3 && spec.mode == SIPX_MODE_FIBER)
      throw std::runtime_error("for 3D models, the mode of application for project_subspace! needs to be (slice,x) or "
                               "(slice,y) or (slice,z)");                              // project_subspace!.jl:121
    I.map = make_segmap(sp);
    if (!spec.basis || spec.basis_cols < 1) throw std::runtime_error("subspace constraints need the matrix A");
    if (spec.basis_rows != I.map.L) throw std::runtime_error("subspace: rows of A do not match the length of what it projects");
    I.cols = spec.basis_cols;
    const long long L = I.map.L;
    I.basis = I.template alloc<T>((size_t)L * I.cols);
    SIPX_HIP(hipMemcpy(I.basis, spec.basis, sizeof(T) * (size_t)L * I.cols, hipMemcpyHostToDevice));
    if (!spec.basis_orth) {
      std::vector<double> Gi = gram_inverse<T>((const T*)spec.basis, L, I.cols);
      std::vector<T> Gt(Gi.begin(), Gi.end());
      I.gram_inv = I.template alloc<T>((size_t)I.cols * I.cols);
      SIPX_HIP(hipMemcpy(I.gram_inv, Gt.data(), sizeof(T) * Gt.size(), hipMemcpyHostToDevice));
    }
    I.X = I.template alloc<T>((size_t)L * I.map.nseg);
    I.t1 = I.template alloc<T>((size_t)I.cols * I.map.nseg);
    I.t2 = I.template alloc<T>((size_t)I.cols * I.map.nseg);
    blas_check(rocblas_create_handle(&I.blas), "create handle");
    blas_check(rocblas_set_stream(I.blas, stream), "set stream");
  } else {
    throw std::runtime_error("unknown external projector");
  }
  } catch (...) {
    this->~ExtProj();
    impl_ = nullptr;
    throw;
  }
}

template <typename T>
ExtProj<T>::~ExtProj() {
  if (!impl_) return;
  ExtImpl<T>& I = *impl_;
  if (I.have_plan) (void)hipfftDestroy(I.plan);
  if (I.plan_r2c) (void)hipfftDestroy(I.plan_r2c);
  if (I.plan_c2r) (void)hipfftDestroy(I.plan_c2r);
  if (I.blas) (void)rocblas_destroy_handle(I.blas);
  if (I.sub_res_host) (void)hipHostFree(I.sub_res_host);
  if (I.fail_host) (void)hipHostFree(I.fail_host);
  if (I.fail_ev) (void)hipEventDestroy(I.fail_ev);
  for (void* p : I.owned)
    if (p) (void)hipFree(p);
  delete impl_;
  impl_ = nullptr;
}

static rocblas_status gemm_T(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const float* A,
                             int lda, const float* B, int ldb, float* C, int ldc) {
  const float one = 1.f, zero = 0.f;
  return rocblas_sgemm(h, ta, tb, m, n, k, &one, A, lda, B, ldb, &zero, C, ldc);
}
static rocblas_status gemm_T(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double* A,
                             int lda, const double* B, int ldb, double* C, int ldc) {
  const double one = 1.0, zero = 0.0;
  return rocblas_dgemm(h, ta, tb, m, n, k, &one, A, lda, B, ldb, &zero, C, ldc);
}

static rocblas_status gemm_sb(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const float* A,
                              int lda, long long sa, const float* B, int ldb, long long sb, float* C, int ldc, long long sc, int batch) {
  const float one = 1.f, zero = 0.f;
  return rocblas_sgemm_strided_batched(h, ta, tb, m, n, k, &one, A, lda, sa, B, ldb, sb, &zero, C, ldc, sc, batch);
}
static rocblas_status gemm_sb(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double* A,
                              int lda, long long sa, const double* B, int ldb, long long sb, double* C, int ldc, long long sc, int batch) {
  const double one = 1.0, zero = 0.0;
  return rocblas_dgemm_strided_batched(h, ta, tb, m, n, k, &one, A, lda, sa, B, ldb, sb, &zero, C, ldc, sc, batch);
}
// dst <- src when the last search found v outside the set (ps->need), or always when ps == nullptr
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_copy_if_needed(long long N, const T* __restrict__ src, T* __restrict__ dst,
                                                          const ProjScalars<T>* __restrict__ ps) {
  if (ps && !ps->need) return;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK) dst[e] = src[e];
}

// Is every matrix of I.Bd positive definite?  info[l] != 0 where not.  Right-looking blocked Cholesky, 64 columns a step:
// k_chol_diag, then the block row and the trailing update as two batched GEMMs (the whole trailing square: the lower half is
// wasted work the library does faster than a loop over block columns would save).
template <typename T>
static void rank_cert_factor(ExtImpl<T>& I, int k) {
  hipStream_t s = I.stream;
  const double one = 1.0, zero = 0.0, mone = -1.0;
  const long long sG = (long long)k * k, sP = (long long)64 * k;
  const auto N_ = rocblas_operation_none, T_ = rocblas_operation_transpose;
  SIPX_HIP(hipMemsetAsync(I.info, 0, sizeof(rocblas_int) * I.batch, s));
  for (int J = 0; J < k; J += 64) {
    const int bs = std::min(64, k - J), rem = k - J - bs;
    hipLaunchKernelGGL(k_chol_diag, dim3(I.batch), dim3(256), 0, s, k, J, bs, I.Bd, sG, I.cert_w, I.info);
    if (rem <= 0) break;
    double* panel = I.Bd + (long long)(J + bs) * k + J;            // rows J .. J+bs, columns from J+bs on
    blas_check(rocblas_dgemm_strided_batched(I.blas, T_, N_, bs, rem, bs, &one, I.cert_w, 64, 4096, panel, k, sG, &zero, I.cert_p, 64, sP, I.batch),
               "certificate: block row");
    double* trail = I.Bd + (long long)(J + bs) * k + (J + bs);
    blas_check(rocblas_dgemm_strided_batched(I.blas, T_, N_, rem, rem, bs, &mone, I.cert_p, 64, sP, I.cert_p, 64, sP, &one, trail, k, sG, I.batch),
               "certificate: trailing update");
  }
  SIPX_HIP(hipGetLastError());
}

// ---- Float32 iteration on deflated Gram matrices (round 5) ------------------------------------------------------------------------
// The products with the Gram matrices are what a call of the filtered route is made of (28 of them, 0.36 ms each for 512 matrices of
// 512 x 512 in Float64: 42 TFLOP/s; the same product in Float32 takes 0.20 ms).  A Gram matrix of a velocity slice cannot be rounded
// to Float32 as it is: its constant part is 1e5 ... 1e7 times the eigenvalues the block is after, and 6e-8 of THAT is several percent
// of them.  With the pairs far above the rest taken out first -- G~ = G - sum theta_d x_d x_d', formed in Float64 from Ritz pairs the
// first Rayleigh-Ritz step of the call has left accurate to 1e-15 theta_1 (they converge by a factor 1e-5 per step), then rounded --
// what is left has entries of the size of the wanted eigenvalues, its rounding (6e-8 theta_2) is 300 times below the acceptance level
// of a pair (k_sub_residual: eps(Float32) sqrt(theta_1 theta_j) = 3e-5 theta_j on such a slice), and filters, orthonormalisation and
// Rayleigh-Ritz steps run in Float32 on it: SGEMM for the big products, the small Gram matrices (Y'Y, Q'G~Q) accumulated in Float64
// by a kernel of the engine's own (Cholesky-QR of a filtered block in Float32 would need its condition below 3e3), Cholesky factor
// and Ritz problem in Float64 as before.  The deflated pairs keep their places in the Float64 block; inside G~ they are null vectors
// -- the lowest columns, which no filter touches anyway.  Acceptance is the level of k_sub_residual against the largest eigenvalue of
// the matrix as it was; the inertia certificate then runs in Float64 on G itself with the merged block, as for the Float64 loop.
// Anything the Float32 loop cannot do (a pair to deflate that is not accurate, more than 16 of them, a wanted eigenvalue so small
// that Float32's floor is above its level, a stall) goes back to the Float64 loop from the state the first step left.
// MEASURED AND NOT THE DEFAULT (SIPX_RANK_F32=1 switches it on; SIPX_RANK_STRICT=1 always keeps Float64, its level is out of
// Float32's reach).  512 slices of 512 x 512, r = 32: a filter degree costs 0.35 ms instead of 0.53 (product 0.20 instead of 0.36, the
// recurrence kernel half its bytes), the three skinny products and the conversions 1.2 ms per call, and the iterates stay where the
// Float64 loop's are (1.06e-5 against 1.06e-5 from the strict route after 16 iterations of C4) -- but the loop spends MORE products
// (C4, 18 calls: 662 against 519; {bounds, rank} alone: 223 against 245) and the first Float32 call of a context pays 1.4 s for the
// library's Float32 kernels, the first call with more than two directions to project another 0.35 s: C4 47.1 against 47.5 ms per
// iteration once those are paid, 52.6 against 50.9 over iterations 3-16.  Not worth its 600 lines as a default; kept, tested
// (tests/test_gpu_round5.py::test_float32_loop_of_the_rank_projector), for the batch sizes and ranks where the products dominate.
#define DEFL_RATIO 32.0
#define DEFL_N 8              // columns of the refined top block (a matrix with more pairs far above the rest keeps the Float64 loop)
// How many of the largest Ritz pairs of the PREVIOUS call lie far above the r-th (nd), per matrix; the top DEFL_N columns of the
// previous block, largest first, into Xtop (k x DEFL_N per matrix: column 0 the largest).
__global__ __launch_bounds__(BLOCK) void k_defl_top(int k, int b, int r, int batch, const double* __restrict__ Wprev, const double* __restrict__ X,
                                                    int* __restrict__ nd, double* __restrict__ Xtop) {
  const long long per = (long long)k * DEFL_N, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per, o = e - l * per;
    const int d = (int)(o / k), i = (int)(o - (long long)d * k);
    Xtop[e] = X[l * (long long)k * b + (long long)(b - 1 - d) * k + i];
    if (o == 0) {
      const double* Wl = Wprev + l * b;
      const double thr = Wl[b - r];
      int n = 0;
      for (int j = b - 1; j >= 0 && Wl[j] > DEFL_RATIO * thr; --j) ++n;
      nd[l] = n;
    }
  }
}
// One step of block power iteration on the top block: Xtop <- orthonormalised Y (= G Xtop), the larger columns first (modified
// Gram-Schmidt: column d loses what it shares with columns 0 .. d-1, then is scaled to unit length).  One workgroup per matrix.
__global__ __launch_bounds__(BLOCK) void k_defl_gs(int k, int batch, const double* __restrict__ Y, double* __restrict__ Xtop) {
  __shared__ double sm[BLOCK / 64];
  __shared__ double s_c;
  const int l = blockIdx.x;
  const double* Yl = Y + (long long)l * k * DEFL_N;
  double* Xl = Xtop + (long long)l * k * DEFL_N;
  auto block_sum = [&](double v) -> double {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int q = 0; q < BLOCK / 64; ++q) t += sm[q];
    return t;
  };
  for (int d = 0; d < DEFL_N; ++d) {
    for (int i = threadIdx.x; i < k; i += BLOCK) Xl[(long long)d * k + i] = Yl[(long long)d * k + i];
    __syncthreads();
    for (int e = 0; e < d; ++e) {
      double a = 0;
      for (int i = threadIdx.x; i < k; i += BLOCK) a += Xl[(long long)e * k + i] * Xl[(long long)d * k + i];
      a = block_sum(a);
      for (int i = threadIdx.x; i < k; i += BLOCK) Xl[(long long)d * k + i] -= a * Xl[(long long)e * k + i];
      __syncthreads();
    }
    double n2 = 0;
    for (int i = threadIdx.x; i < k; i += BLOCK) n2 += Xl[(long long)d * k + i] * Xl[(long long)d * k + i];
    n2 = block_sum(n2);
    if (threadIdx.x == 0) s_c = n2 > 0 ? 1.0 / sqrt(n2) : 0.0;
    __syncthreads();
    const double c = s_c;
    for (int i = threadIdx.x; i < k; i += BLOCK) Xl[(long long)d * k + i] *= c;
    __syncthreads();
  }
}
// With Y = G Xtop of the refined block: theta_d = x_d'y_d and the residual of every pair that is to be deflated; tmax; the verdict.
// res[1] |= 64 when the Float32 loop must not run for this batch: more than DEFL_N pairs far above, a pair that is not an eigenpair
// to 3 % of the level of the r-th pair (what is left of it stays in G~), a level that Float32's own floor (3e-7 of the largest
// eigenvalue of G~) does not clear by a factor of four, a matrix without a positive r-th eigenvalue.
__global__ __launch_bounds__(BLOCK) void k_defl_plan(int k, int b, int r, int batch, const double* __restrict__ Wprev, const double* __restrict__ Y,
                                                     const double* __restrict__ Xtop, double eps_bw, double tol, int* __restrict__ nd,
                                                     double* __restrict__ theta, double* __restrict__ tmax, unsigned long long* res) {
  __shared__ double sm[BLOCK / 64];
  const int l = blockIdx.x;
  const double* Wl = Wprev + (long long)l * b;
  const double thr = Wl[b - r];
  const int n = nd[l];
  bool bad = !(thr > 0) || n > DEFL_N || n >= r;
  auto block_sum = [&](double v) -> double {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int q = 0; q < BLOCK / 64; ++q) t += sm[q];
    return t;
  };
  double top = Wl[b - 1];
  double th[DEFL_N], rs[DEFL_N];
  for (int d = 0; d < DEFL_N; ++d) {
    const double* y = Y + ((long long)l * DEFL_N + d) * k;
    const double* x = Xtop + ((long long)l * DEFL_N + d) * k;
    double a = 0;
    for (int i = threadIdx.x; i < k; i += BLOCK) a += x[i] * y[i];
    th[d] = block_sum(a);
    double q = 0;
    for (int i = threadIdx.x; i < k; i += BLOCK) { const double dd = y[i] - th[d] * x[i]; q += dd * dd; }
    rs[d] = sqrt(block_sum(q));
  }
  if (n > 0) top = th[0];
  const double lvl = fmax(eps_bw * sqrt(fmax(top, 0.0) * fmax(thr, 0.0)), tol * top);
  for (int d = 0; d < n && d < DEFL_N; ++d)
    if (!(rs[d] < 0.03 * lvl) || !(th[d] > 0.5 * DEFL_RATIO * thr)) bad = true;
  // the largest eigenvalue G~ keeps: the next pair of the previous call, or the first refined one that is not deflated
  const double next = n < DEFL_N ? fmax(th[n], Wl[b - 1 - n]) : Wl[b - 1 - n];
  if (!(top > 0) || !(3e-7 * next < 0.25 * lvl)) bad = true;
  if (threadIdx.x == 0) {
    for (int d = 0; d < DEFL_N; ++d) theta[(long long)l * DEFL_N + d] = th[d];
    tmax[l] = top;
    if (bad) { nd[l] = 0; atomicOr(res + 1, 64ull); }
  }
}
// G32 <- fl32(G - sum_{d < nd[l]} theta_d x_d x_d')
__global__ __launch_bounds__(BLOCK) void k_defl_build(int k, int batch, const double* __restrict__ G, const double* __restrict__ theta,
                                                      const double* __restrict__ Xtop, const int* __restrict__ nd, float* __restrict__ G32) {
  const long long per = (long long)k * k, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per, o = e - l * per;
    const int row = (int)(o % k), col = (int)(o / k);
    double v = G[e];
    const int n = nd[l];
    for (int d = 0; d < n; ++d) {
      const double* x = Xtop + (l * DEFL_N + d) * (long long)k;
      v -= theta[l * DEFL_N + d] * x[row] * x[col];
    }
    G32[e] = (float)v;
  }
}
// The start block of the Float32 loop in the order of G~'s spectrum: the deflated pairs first (refined: null vectors of G~), then
// the previous call's other vectors, ascending as before.
__global__ __launch_bounds__(BLOCK) void k_defl_convert(int k, int b, int batch, const double* __restrict__ X, const double* __restrict__ Xtop,
                                                        const int* __restrict__ nd, float* __restrict__ X32) {
  const long long per = (long long)k * b, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per, o = e - l * per;
    const int jj = (int)(o / k), i = (int)(o - (long long)jj * k);          // destination column jj
    const int n = nd[l];
    X32[e] = jj < n ? (float)Xtop[(l * DEFL_N + jj) * (long long)k + i]      // (any order among the null vectors)
                    : (float)X[l * per + (long long)(jj - n) * k + i];
  }
}
// back: the b - nd[l] largest Ritz pairs of G~ take the places below the deflated pairs, which go to the top (largest last)
__global__ __launch_bounds__(BLOCK) void k_defl_merge(int k, int b, int batch, const double* __restrict__ W32, const float* __restrict__ X32,
                                                      const int* __restrict__ nd, const double* __restrict__ theta, const double* __restrict__ Xtop,
                                                      double* __restrict__ W, double* __restrict__ X) {
  const long long per = (long long)k * b, total = per * batch;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long l = e / per, o = e - l * per;
    const int j = (int)(o / k), i = (int)(o - (long long)j * k);            // destination column j
    const int n = nd[l];
    if (j >= b - n) {
      const int d = b - 1 - j;                                               // column b - 1: the largest pair, d = 0
      X[e] = Xtop[(l * DEFL_N + d) * (long long)k + i];
      if (i == 0) W[l * b + j] = theta[l * DEFL_N + d];
    } else {
      X[e] = (double)X32[l * per + (long long)(j + n) * k + i];
      if (i == 0) W[l * b + j] = W32[l * b + j + n];
    }
  }
}
// H (b x b, Float64, both triangles, column-major) = A'B for k x b blocks stored in TS: what Cholesky-QR and the Ritz problem are
// built on, accumulated in Float64 whatever the storage.  One workgroup per matrix, 16 x 16 threads, 4 x 4 outputs each, the
// operands staged through LDS 64 rows at a time.
template <typename TS>
__global__ __launch_bounds__(256) void k_gram_bb(int k, int b, const TS* __restrict__ A, const TS* __restrict__ B, double* __restrict__ H) {
  __shared__ TS sa[64 * 65];
  __shared__ TS sb[64 * 65];
  const int l = blockIdx.x, t = threadIdx.x, ti = t & 15, tj = t >> 4;
  const TS* Al = A + (long long)l * k * b;
  const TS* Bl = B + (long long)l * k * b;
  double acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = 0.0;
  for (int r0 = 0; r0 < k; r0 += 64) {
    const int rows = k - r0 < 64 ? k - r0 : 64;
    __syncthreads();
    for (int e = t; e < 64 * 64; e += 256) {
      const int rr = e & 63, c = e >> 6;
      const bool in = rr < rows && c < b;
      sa[c * 65 + rr] = in ? Al[(long long)c * k + r0 + rr] : TS(0);
      sb[c * 65 + rr] = in ? Bl[(long long)c * k + r0 + rr] : TS(0);
    }
    __syncthreads();
    for (int rr = 0; rr < rows; ++rr) {
      double av[4], bv[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) { av[p] = (double)sa[(ti * 4 + p) * 65 + rr]; bv[p] = (double)sb[(tj * 4 + p) * 65 + rr]; }
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] += av[p] * bv[q];
    }
  }
  double* Hl = H + (long long)l * b * b;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ti * 4 + p, j = tj * 4 + q;
      if (i < b && j < b) Hl[(long long)j * b + i] = acc[p][q];
    }
}

static rocblas_status gemm_sbx(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, float alpha, const float* A,
                               int lda, long long sa, const float* B, int ldb, long long sb, float beta, float* C, int ldc, long long sc, int batch,
                               int tune = 0) {
  (void)tune;
  return rocblas_sgemm_strided_batched(h, ta, tb, m, n, k, &alpha, A, lda, sa, B, ldb, sb, &beta, C, ldc, sc, batch);
}
// ---- which rocBLAS kernel for a Float64 batched product (round 5) ----------------------------------------------------------------
// The library's own choice for the filter product G V (512 x 56 x 512 per slice) is a 64 x 32 macro tile: the 56 columns fall into
// two tiles and every Gram matrix crosses the fabric twice (profiles/r05_c4_512_pmc.json: 2.58 GB per launch against 1.3 GB);
// rocblas_gemm_strided_batched_ex lets the caller name a solution, and another one runs the same product in 0.26 instead of 0.36 ms
// (b x b x k: 0.035 instead of 0.061 ms; tools/gemm_solutions_bench.cpp).  The first call of a shape in a process therefore tries
// every solution the library lists for it, on the call's own operands (beta = 0: the output is simply written again):
//   * solutions are grouped by the BITS they produce (a hash of the output: kernels that add in the same order give the same bits --
//     all the fast ones of a shape do) -- the group is chosen by a rule that does not depend on timing noise (the library's own
//     group unless another is more than 8 % faster; among groups within 5 % of the fastest the one with the lowest solution number),
//     so that every process, every rank and every run of a library version ends with the same arithmetic;
//   * inside the group the fastest member by the clock (any member gives the same bits).
// The choice is kept per shape for the life of the process.  SIPX_GEMM_TUNE=0: the library's choice (A/B switch).
struct GemmShape {
  int ta, tb, m, n, k, lda, ldb, ldc;
  bool operator<(const GemmShape& o) const {
    return std::memcmp(this, &o, sizeof(GemmShape)) < 0;
  }
};
__global__ __launch_bounds__(256) void k_hash_bits(const unsigned long long* __restrict__ p, long long n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += p[i] * (2ull * (unsigned long long)i + 1ull);
  atomicAdd(out, acc);       // (integer sums: the order of arrival does not matter)
}
static std::mutex& gemm_tune_mutex() { static std::mutex m; return m; }
static std::map<GemmShape, int>& gemm_tune_table() { static std::map<GemmShape, int> t; return t; }
static bool gemm_tune_on() {
  static const bool on = [] { const char* e = getenv("SIPX_GEMM_TUNE"); return !(e && e[0] == '0'); }();
  return on;
}
static rocblas_status dgemm_ex(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double* alpha, const double* A, int lda,
                               long long sa, const double* B, int ldb, long long sb, const double* beta, double* C, int ldc, long long sc, int batch, int sol) {
  return rocblas_gemm_strided_batched_ex(h, ta, tb, m, n, k, alpha, A, rocblas_datatype_f64_r, lda, sa, B, rocblas_datatype_f64_r, ldb, sb, beta, C,
                                         rocblas_datatype_f64_r, ldc, sc, C, rocblas_datatype_f64_r, ldc, sc, batch, rocblas_datatype_f64_r,
                                         sol ? rocblas_gemm_algo_solution_index : rocblas_gemm_algo_standard, sol, 0);
}
static int gemm_tune(rocblas_handle h, const GemmShape& key, const double* A, long long sa, const double* B, long long sb, double* C, long long sc, int batch) {
  const rocblas_operation ta = (rocblas_operation)key.ta, tb = (rocblas_operation)key.tb;
  const double one = 1.0, zero = 0.0;
  hipStream_t s = nullptr;
  if (rocblas_get_stream(h, &s) != rocblas_status_success) return 0;
  rocblas_int ns = 0;
  if (rocblas_gemm_strided_batched_ex_get_solutions(h, ta, tb, key.m, key.n, key.k, &one, A, rocblas_datatype_f64_r, key.lda, sa, B, rocblas_datatype_f64_r, key.ldb,
                                                    sb, &zero, C, rocblas_datatype_f64_r, key.ldc, sc, C, rocblas_datatype_f64_r, key.ldc, sc, batch,
                                                    rocblas_datatype_f64_r, rocblas_gemm_algo_solution_index, 0, nullptr, &ns) != rocblas_status_success || ns < 1)
    return 0;
  std::vector<rocblas_int> sols(ns);
  if (rocblas_gemm_strided_batched_ex_get_solutions(h, ta, tb, key.m, key.n, key.k, &one, A, rocblas_datatype_f64_r, key.lda, sa, B, rocblas_datatype_f64_r, key.ldb,
                                                    sb, &zero, C, rocblas_datatype_f64_r, key.ldc, sc, C, rocblas_datatype_f64_r, key.ldc, sc, batch,
                                                    rocblas_datatype_f64_r, rocblas_gemm_algo_solution_index, 0, sols.data(), &ns) != rocblas_status_success)
    return 0;
  sols.resize(ns);
  unsigned long long* dh = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipMalloc((void**)&dh, sizeof(unsigned long long)) != hipSuccess) return 0;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const long long span = sc * (long long)batch;
  struct Res { int sol; unsigned long long hash; float ms; };
  std::vector<Res> res;
  auto probe = [&](int sol, Res& r) -> bool {
    r.sol = sol;
    if (dgemm_ex(h, ta, tb, key.m, key.n, key.k, &one, A, key.lda, sa, B, key.ldb, sb, &zero, C, key.ldc, sc, batch, sol) != rocblas_status_success) return false;
    (void)hipMemsetAsync(dh, 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_hash_bits, dim3(1024), dim3(256), 0, s, reinterpret_cast<const unsigned long long*>(C), span, dh);
    if (hipMemcpyAsync(&r.hash, dh, sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess) return false;
    (void)hipEventRecord(e0, s);
    for (int rep = 0; rep < 2; ++rep)
      if (dgemm_ex(h, ta, tb, key.m, key.n, key.k, &one, A, key.lda, sa, B, key.ldb, sb, &zero, C, key.ldc, sc, batch, sol) != rocblas_status_success) return false;
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) return false;
    r.ms = 0;
    (void)hipEventElapsedTime(&r.ms, e0, e1);
    return r.ms > 0;
  };
  Res def{};
  const bool have_def = probe(0, def);
  for (int sol : sols) { Res r{}; if (probe(sol, r)) res.push_back(r); }
  int choice = 0;
  if (have_def && !res.empty()) {
    // groups by output bits: fastest member, lowest solution number
    std::map<unsigned long long, std::pair<float, int>> best;      // hash -> (fastest time, its solution)
    std::map<unsigned long long, int> lowest;
    for (const Res& r : res) {
      auto it = best.find(r.hash);
      if (it == best.end() || r.ms < it->second.first) best[r.hash] = {r.ms, r.sol};
      auto lt = lowest.find(r.hash);
      if (lt == lowest.end() || r.sol < lt->second) lowest[r.hash] = r.sol;
    }
    float t_best = 1e30f;
    for (auto& kv : best) t_best = std::min(t_best, kv.second.first);
    const float t_def_group = best.count(def.hash) ? std::min(def.ms, best[def.hash].first) : def.ms;
    if (t_def_group <= 1.08f * t_best) {
      choice = (best.count(def.hash) && best[def.hash].first < def.ms) ? best[def.hash].second : 0;      // the library's own arithmetic, its fastest kernel
    } else {
      unsigned long long pick = 0;
      int low = 0x7fffffff;
      for (auto& kv : best)
        if (kv.second.first <= 1.05f * t_best && lowest[kv.first] < low) { low = lowest[kv.first]; pick = kv.first; }
      choice = best[pick].second;
    }
    if (getenv("SIPX_GEMM_TUNE_DEBUG"))
      fprintf(stderr, "[sipx gemm] %c%c %d x %d x %d, batch %d: %d solutions in %zu groups by bits; library %.3f ms, fastest %.3f ms; solution %d\n",
              key.ta == rocblas_operation_none ? 'N' : 'T', key.tb == rocblas_operation_none ? 'N' : 'T', key.m, key.n, key.k, batch, ns, best.size(),
              def.ms / 2, t_best / 2, choice);
  }
  // the call's own result, by the kernel that was chosen (the probes left another candidate's output in C)
  (void)dgemm_ex(h, ta, tb, key.m, key.n, key.k, &one, A, key.lda, sa, B, key.ldb, sb, &zero, C, key.ldc, sc, batch, choice);
  (void)hipStreamSynchronize(s);
  (void)hipFree(dh);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return choice;
}
static rocblas_status gemm_sbx(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, double alpha, const double* A,
                               int lda, long long sa, const double* B, int ldb, long long sb, double beta, double* C, int ldc, long long sc, int batch,
                               int tune = 0) {
  // (the tuned path: plain products C = A B with whole outputs.  tune 1: a FIXED shape -- what the filter loop is made of; tune 2: a
  //  shape whose first dimension follows the data -- the projection on the vectors far above the rest, X_L' Z with 6 ... 31 rows,
  //  for which the library's choice takes 0.2-0.56 ms where another kernel takes 0.03: tried once, at the first such call, and the
  //  solution kept for every row count (a call it does not take falls back to the library's); 0: the library's choice)
  //  3: no trial, but the kernel a trial of the same shape has chosen, if there was one -- products that accumulate, beta = 1)
  const bool may_try = (tune == 1 || tune == 2) && alpha == 1.0 && beta == 0.0 && (const double*)C != A && (const double*)C != B;
  if (tune && gemm_tune_on() && batch > 0 && (long long)m * n * k >= (1ll << 16)) {
    const GemmShape key{(int)ta, (int)tb, tune == 2 ? -1 : m, n, k, lda, ldb, ldc};
    int sol = 0;
    bool known = false;
    {
      std::lock_guard<std::mutex> lk(gemm_tune_mutex());
      auto it = gemm_tune_table().find(key);
      if (it != gemm_tune_table().end()) { sol = it->second; known = true; }
    }
    if (!known && may_try) {
      GemmShape probe = key;
      probe.m = m;
      sol = gemm_tune(h, probe, A, sa, B, sb, C, sc, batch);      // (leaves the product in C)
      std::lock_guard<std::mutex> lk(gemm_tune_mutex());
      gemm_tune_table()[key] = sol;
      return rocblas_status_success;
    }
    if (sol != 0) {
      const rocblas_status st = dgemm_ex(h, ta, tb, m, n, k, &alpha, A, lda, sa, B, ldb, sb, &beta, C, ldc, sc, batch, sol);
      if (st == rocblas_status_success) return st;
      if (tune != 2) {
        std::lock_guard<std::mutex> lk(gemm_tune_mutex());      // (a solution that does not take this batch count: the library's choice from here on)
        gemm_tune_table()[key] = 0;
      }
    }
  }
  return rocblas_dgemm_strided_batched(h, ta, tb, m, n, k, &alpha, A, lda, sa, B, ldb, sb, &beta, C, ldc, sc, batch);
}

// the buffers of one precision of the filtered iteration
template <typename TS>
struct RouteBufs {
  TS *G, *Gp;              // the matrices; room for the packed ones
  TS* X;                   // Ritz vectors: the start, then every Rayleigh-Ritz step's result
  TS *A, *F1, *F2;         // three blocks in rotation
  TS* Xc;                  // Ritz vectors of the packed matrices
  TS* Cs;                  // b x b per matrix: inverse Cholesky factor / masked projection coefficients
  TS* Zs;                  // b x b per matrix: eigenvectors of the Ritz problem
};
// what a call carries from one loop to the next (the Float64 front step, the Float32 loop, the Float64 loop again)
struct ChebCtl {
  int w = 0, k = 0;
  bool cold = false;
  int mults = 0;
  bool fresh_start = true, tried_other = false;
  int ramp = -1, max_outer = 9;
  bool cheap_fail = false;
  bool start_rr = false;        // X, F2 = G X and the Ritz values are there already: the loop opens with that step's decisions
  bool handover = false;        // leave after the first Rayleigh-Ritz step of a warm call (the Float32 loop takes over)
  const int* nd = nullptr;      // Float32 loop: deflated pairs per matrix, the largest eigenvalue of the matrix as it was
  const double* tmax = nullptr;
  int* nd_c = nullptr;          // ... of the packed matrices
  double* tmax_c = nullptr;
  double* W = nullptr;          // Ritz values of the full batch (b per matrix) in this loop's order
  double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::chrono::steady_clock::time_point t_start, t_mark;
  double last_res = -1, last_raw = -1;
  bool hidden = true;           // at convergence: the energy bound of k_sub_residual does NOT rule out a larger eigenvalue outside the block
};
enum { CHEB_GIVE_UP = 0, CHEB_CONVERGED = 1, CHEB_HANDOVER = 2 };

// The loop of the filtered route in the precision TS of its big arrays: Rayleigh-Ritz step, residual, decisions, filter.
// CHEB_CONVERGED: every wanted pair is below its level (B.X and C.W hold the pairs of the whole batch, unpacked);
// CHEB_HANDOVER: the first step of a warm call is done and another loop may go on from it; CHEB_GIVE_UP: the caller decomposes fully.
template <typename T, typename TS>
static int cheb_loop(ExtImpl<T>& I, RouteBufs<TS> B, ChebCtl& C) {
  hipStream_t s = I.stream;
  const int k = C.k, w = C.w;
  const int b = I.sub_b, r = I.r, batch = I.batch;
  const long long sG = (long long)k * k, sX = (long long)k * b, sH = (long long)b * b;
  const auto N_ = rocblas_operation_none, T_ = rocblas_operation_transpose;
  const auto& KN = I.knobs;
  const int dbg = KN.dbg, budget = KN.budget, m_cap = KN.m_cap;
  const bool own_jacobi = KN.own_jacobi, fused_proj = KN.fused_proj;
  const double tol = KN.tol;
  constexpr bool F32 = std::is_same<TS, float>::value;
  const char* prec = F32 ? "Float32" : "Float64";
  // index of the Ritz value that ends the damped interval (C4 with 24 guards: 2 / 3 / 4 / 6 -> 16.8 / 16.5 / 16.6 / 16.0 it/s)
  const int g = KN.guard >= 0 ? std::min(KN.guard, b - r - 1) : std::max(2, (b - r) / 12);
  auto mark = [&](int which) {
    if (dbg < 2) return;
    SIPX_HIP(hipStreamSynchronize(s));
    const auto now = std::chrono::steady_clock::now();
    C.ph[which] += std::chrono::duration<double, std::milli>(now - C.t_mark).count();
    C.t_mark = now;
  };
  TS* X = B.X;
  // three blocks in rotation: A the block to orthonormalise (then the orthonormal basis), F1 and F2 free
  TS *A = B.A, *F1 = B.F1, *F2 = B.F2;
  // what the loop works on: the whole batch -- or, once three quarters of it have converged, the matrices that have not, packed
  // (G into the certificate's matrix, which is free until the end; X into a block of its own: the converged matrices' vectors
  // stay where they are; the scratch blocks are used from their front).  The second and later filters of a call were observed
  // to run for 1 to 27 of 512 slices.
  int nb = batch;
  TS* Gd = B.G;
  double *Ws = C.W, *Fro = I.Fro;
  const int* nd = C.nd;
  const double* tmax = C.tmax;
  bool packed = false;
  const bool may_pack = I.sub_cap > 0 && KN.pack;
  int m_prev = 0;
  double prev = -1;
  bool retried = false;
  int m_lim = m_cap;
  // A start that says nothing about this input -- none at all (the block is pseudo-random: `cold`), or the previous call's vectors
  // on the second iteration of a solve (first residual above 1e-3 theta_max) -- is RAMPED instead of given up (rounds 3-4 decomposed
  // fully: 105-185 ms for 512 slices of 512 x 512).  What went wrong with long filters from such a block (DESIGN_HISTORY, round 3):
  // a direction 1e5 times the rest that the block holds only roughly cannot be projected out of the products, is amplified in every
  // column and leaves a block of dependent columns; and the Ritz values of a cold block say nothing about where the unwanted part
  // of the spectrum ends.  So: three steps of degree one (a shifted power step each, no product beyond the Rayleigh-Ritz step's own:
  // whatever is far above the rest converges by its ratio per step), then degrees 2, 4, 8 -- every Rayleigh-Ritz step moves the
  // interval ends towards the spectrum's -- then the usual filters.  The stall rule waits until the ramp is over.
  static const int ramp_deg[6] = {1, 1, 1, 2, 4, 8};
  const int budget_all = budget * 2;
  bool have_rr = C.start_rr;
  C.start_rr = false;
  for (int outer = 0; outer < C.max_outer; ++outer) {
    if (!have_rr) {
      // Rayleigh-Ritz on span(A): Cholesky QR (twice behind a filter: its columns lean on each other), H = Q'GQ, X = Q S
      SIPX_HIP(hipMemsetAsync(I.info, 0, sizeof(rocblas_int) * 2 * batch, s));
      for (int pass = 0; pass < (m_prev > 0 ? 2 : 1); ++pass) {
        if constexpr (F32) hipLaunchKernelGGL((k_gram_bb<TS>), dim3(nb), dim3(256), 0, s, k, b, A, A, I.Hs);
        else blas_check(gemm_sbx(I.blas, T_, N_, b, b, k, TS(1), A, k, sX, A, k, sX, TS(0), (TS*)I.Hs, b, sH, nb, 1), "Y'Y");
        hipLaunchKernelGGL((k_chol_inv<TS>), dim3(nb), dim3(256), 0, s, b, nb, I.Hs, B.Cs, I.info);
        blas_check(gemm_sbx(I.blas, N_, N_, k, b, b, TS(1), A, k, sX, B.Cs, b, sH, TS(0), F1, k, sX, nb, 1), "Y Rinv");
        std::swap(A, F1);
      }
      mark(0);
      blas_check(gemm_sbx(I.blas, N_, N_, k, b, k, TS(1), Gd, k, sG, A, k, sX, TS(0), F1, k, sX, nb, 1), "G Q");
      ++C.mults;
      mark(1);
      if constexpr (F32) hipLaunchKernelGGL((k_gram_bb<TS>), dim3(nb), dim3(256), 0, s, k, b, A, F1, I.Hs);
      else blas_check(gemm_sbx(I.blas, T_, N_, b, b, k, TS(1), A, k, sX, F1, k, sX, TS(0), (TS*)I.Hs, b, sH, nb, 1), "Q'GQ");
      mark(2);
      if (own_jacobi || F32)
        hipLaunchKernelGGL((k_ritz_jacobi<TS>), dim3(nb), dim3(256), 0, s, b, nb, I.Hs, B.Zs, Ws, I.info + batch, I.info + 2 * batch);
      else
        blas_check(rocsolver_dsyevj_strided_batched(I.blas, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper, b, I.Hs, b, sH,
                                                    0.0, I.Es, 100, I.info + 2 * batch, Ws, b, I.info + batch, nb), "syevj (Ritz)");
      mark(3);
      blas_check(gemm_sbx(I.blas, N_, N_, k, b, b, TS(1), A, k, sX, B.Zs, b, sH, TS(0), X, k, sX, nb, 1), "Q Z");
      blas_check(gemm_sbx(I.blas, N_, N_, k, b, b, TS(1), F1, k, sX, B.Zs, b, sH, TS(0), F2, k, sX, nb, 1), "(GQ) Z");
      mark(2);
    } else {
      SIPX_HIP(hipMemsetAsync(I.info, 0, sizeof(rocblas_int) * 2 * batch, s));
    }
    have_rr = false;
    SIPX_HIP(hipMemsetAsync(I.sub_res, 0, 8 * sizeof(unsigned long long), s));
    SIPX_HIP(hipMemsetAsync(I.sub_res + 2, 0x7f, sizeof(unsigned long long), s));       // a large positive double: the minimum starts there
    hipLaunchKernelGGL((k_sub_residual<TS>), dim3(nb), dim3(BLOCK), 0, s, k, b, r, nb, F2, X, Ws, b, I.info, I.info + batch, Fro, I.sub_res, I.Es,
                       KN.eps_bw, tol, std::min(KN.window, b - r - g - 1), nd, tmax);
    hipLaunchKernelGGL(k_cheb_plan, dim3((nb + 63) / 64), dim3(64), 0, s, b, g, r, nb, Ws, I.Es, tol, I.sub_res, nd);
    if (may_pack && !packed) hipLaunchKernelGGL(k_sub_list, dim3(1), dim3(256), 0, s, nb, I.Es, tol, I.sub_idx, I.sub_res);
    SIPX_HIP(hipMemcpyAsync(I.sub_res_host, I.sub_res, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    SIPX_HIP(hipStreamSynchronize(s));
    mark(4);
    double res, tmin, res_raw;
    std::memcpy(&res, &I.sub_res_host[0], sizeof(double));
    std::memcpy(&tmin, &I.sub_res_host[2], sizeof(double));
    std::memcpy(&res_raw, &I.sub_res_host[6], sizeof(double));
    C.last_res = res; C.last_raw = res_raw;
    const bool failed = (I.sub_res_host[1] & 15ull) != 0;
    const int n_open = (int)I.sub_res_host[4];
    const bool pack_now = may_pack && !packed && res > tol && n_open >= 1 && n_open <= I.sub_cap;
    const int nl = (int)I.sub_res_host[(packed || pack_now) ? 3 : 5];
    if (dbg) fprintf(stderr, "[sipx rank] filtered subspace step %d (%s): %d products, residual %.3e (%.3e of theta_max), fail-bits %llu, t_r %.4g, "
                             "%d vectors far above, %d of %d matrices%s%s, %.2f ms\n",
                     outer, prec, C.mults, res, res_raw, I.sub_res_host[1], tmin, nl, nb, batch, packed ? " (packed)" : "", C.ramp >= 0 ? " (ramp)" : "",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - C.t_start).count());
    if (dbg >= 3 && (own_jacobi || F32)) {
      std::vector<rocblas_int> sw(nb);
      SIPX_HIP(hipMemcpy(sw.data(), I.info + 2 * batch, sizeof(rocblas_int) * nb, hipMemcpyDeviceToHost));
      long long tot = 0; int mx = 0;
      for (int l = 0; l < nb; ++l) { tot += sw[l]; mx = std::max(mx, (int)sw[l]); }
      fprintf(stderr, "[sipx rank]   Jacobi sweeps: mean %.1f, max %d\n", (double)tot / nb, mx);
    }
    if (dbg >= 3) {                                        // how many matrices of the batch still need a filter
      std::vector<double> pm(nb);
      SIPX_HIP(hipMemcpy(pm.data(), I.Es, sizeof(double) * nb, hipMemcpyDeviceToHost));
      int a12 = 0, a10 = 0, a8 = 0, worst_l = 0;
      for (int l = 0; l < nb; ++l) { a12 += pm[l] > 1e-12; a10 += pm[l] > 1e-10; a8 += pm[l] > 1e-8; if (pm[l] > pm[worst_l]) worst_l = l; }
      fprintf(stderr, "[sipx rank]   matrices above 1e-12: %d, above 1e-10: %d, above 1e-8: %d (of %d)\n", a12, a10, a8, nb);
      if (dbg >= 4) {                                      // the worst matrix: its Ritz values and the residual of every column, in units of theta_j
        std::vector<double> ww(b);
        std::vector<TS> zz((size_t)k * b), xx((size_t)k * b);
        SIPX_HIP(hipMemcpy(ww.data(), Ws + (size_t)worst_l * b, sizeof(double) * b, hipMemcpyDeviceToHost));
        SIPX_HIP(hipMemcpy(zz.data(), F2 + (size_t)worst_l * sX, sizeof(TS) * sX, hipMemcpyDeviceToHost));
        SIPX_HIP(hipMemcpy(xx.data(), X + (size_t)worst_l * sX, sizeof(TS) * sX, hipMemcpyDeviceToHost));
        fprintf(stderr, "[sipx rank]   worst matrix %d (%.3e): column: Ritz value / theta_max, residual / theta_j\n", worst_l, pm[worst_l]);
        for (int j = b - 1; j >= 0; --j) {
          double t = 0;
          for (int i = 0; i < k; ++i) { const double d = (double)zz[(size_t)j * k + i] - ww[j] * (double)xx[(size_t)j * k + i]; t += d * d; }
          fprintf(stderr, " %d:%.3e/%.2e", b - j, ww[j] / ww[b - 1], std::sqrt(t) / (ww[j] > 0 ? ww[j] : 1.0));
        }
        fprintf(stderr, "\n");
      }
    }
    if (failed) break;
    // The previous call's vectors say little about this input (the first iterations of a solve): filters started from there
    // were observed to swamp the guard columns and then stall at a residual of 1e-8 theta_max.
    if (F32 && C.fresh_start && res_raw > 1e-3) break;   // (a start that says too little: the Float64 loop has the ramp for it)
    if (C.fresh_start && C.ramp < 0 && res_raw > 1e-3) {
      // the feasibility estimate is asked for every tenth iteration only, its own vectors are ten iterations old: those of the
      // y update of this iteration (another input, but the same x behind it) may be the better start
      if (!F32 && w == 1 && I.sub_have[0] && !C.tried_other) {
        C.tried_other = true;
        SIPX_HIP(hipMemcpyAsync(A, I.Xs[0], sizeof(double) * (size_t)sX * batch, hipMemcpyDeviceToDevice, s));
        if (dbg) fprintf(stderr, "[sipx rank] poor start: once more from the vectors of the y update\n");
        continue;
      }
      if (KN.cold) {
        C.ramp = 0;                   // go on from what the step left, by the ramp
        C.max_outer = 24;
        C.cold = true;                // (its budget)
        if (dbg) fprintf(stderr, "[sipx rank] poor start (%.3e of theta_max): ramped filters\n", res_raw);
      } else {
        C.cheap_fail = true;          // one Rayleigh-Ritz step spent: nothing the next call should sit out for
        break;
      }
    }
    C.fresh_start = false;
    if (res <= tol) {
      if (packed) {                   // the vectors and Ritz values of the packed matrices go back to their places
        hipLaunchKernelGGL((k_sub_move<TS>), dim3(NB), dim3(BLOCK), 0, s, sX, nb, I.sub_idx, X, B.X, 1);
        hipLaunchKernelGGL((k_sub_move<double>), dim3(64), dim3(BLOCK), 0, s, (long long)b, nb, I.sub_idx, Ws, C.W, 1);
      }
      // (packed: the bit of the converged matrices is no longer seen; Float32: the bound was taken on the deflated matrices)
      C.hidden = F32 || packed || (I.sub_res_host[1] & 16ull) != 0;
      return CHEB_CONVERGED;
    }
    // (the Float32 loop takes over from the first step of a warm call: that step has told how far the start is off and has left the
    //  pairs far above the rest accurate enough to be taken out of the matrices)
    if (C.handover && C.ramp < 0 && !packed) return CHEB_HANDOVER;
    if (C.ramp >= 0) prev = -1;                           // (no verdict on a filter while the intervals are still being found)
    if (prev > 0 && !(res < 0.5 * prev) && !retried) {
      // The filter did not do what its degree promised -- in a long C4 solve (80 iterations) the residual ROSE behind a filter in
      // one call of nine (4.9e-5 -> 1.4e-4 for all 512 slices, 3.6e-8 -> 3.0e-6 for seven): the intervals of a call's first filter
      // come from Ritz values the previous call's vectors give on THIS call's matrices, and where a spectrum has moved a
      // column is amplified where it should be damped, swamps its neighbours and is re-seeded.  The Rayleigh-Ritz step behind
      // the filter has corrected the Ritz values; what it left is no worse than a usual start (1e-4 ... 1e-5), so the call goes
      // on from there with filters of half the degree -- once: 16-45 products against a full decomposition and the calls that
      // used to sit out behind it.
      retried = true;
      m_lim = std::max(4, m_lim / 2);
      prev = -1;
      if (dbg) fprintf(stderr, "[sipx rank] residual %.3e behind a filter (not half of the one before): once more, degree <= %d\n", res, m_lim);
    }
    if (prev > 0 && !(res < 0.5 * prev)) {               // the filter did not do what its degree promised
      if (dbg) {                                           // which matrix, and what its Ritz values look like
        std::vector<double> pm(nb), ww((size_t)b * nb);
        SIPX_HIP(hipMemcpy(pm.data(), I.Es, sizeof(double) * nb, hipMemcpyDeviceToHost));
        SIPX_HIP(hipMemcpy(ww.data(), Ws, sizeof(double) * b * nb, hipMemcpyDeviceToHost));
        int worst_l = 0, above = 0;
        for (int l = 0; l < nb; ++l) { if (pm[l] > pm[worst_l]) worst_l = l; above += pm[l] > tol ? 1 : 0; }
        fprintf(stderr, "[sipx rank] stalled: %d matrices above the tolerance, worst %d (%.3e); its Ritz values:", above, worst_l, pm[worst_l]);
        for (int j = 0; j < b; ++j) fprintf(stderr, " %.4e", ww[(size_t)worst_l * b + j]);
        fprintf(stderr, "\n");
      }
      break;
    }
    prev = res;
    if (pack_now) {
      // X, G X, the Ritz values, ||G||_F^2 and G itself of the open matrices, packed; G X lands in F1 (free: the product G Q
      // has gone into F2 = (G Q) Z), which then takes the place of F2
      hipLaunchKernelGGL((k_sub_move<TS>), dim3(NB), dim3(BLOCK), 0, s, sG, n_open, I.sub_idx, B.G, B.Gp, 0);
      hipLaunchKernelGGL((k_sub_move<TS>), dim3(NB), dim3(BLOCK), 0, s, sX, n_open, I.sub_idx, X, B.Xc, 0);
      hipLaunchKernelGGL((k_sub_move<TS>), dim3(NB), dim3(BLOCK), 0, s, sX, n_open, I.sub_idx, F2, F1, 0);
      hipLaunchKernelGGL((k_sub_move<double>), dim3(64), dim3(BLOCK), 0, s, (long long)b, n_open, I.sub_idx, C.W, I.Wc, 0);
      hipLaunchKernelGGL((k_sub_move<double>), dim3(1), dim3(BLOCK), 0, s, 1LL, n_open, I.sub_idx, I.Fro, I.Froc, 0);
      if (nd) {
        hipLaunchKernelGGL((k_sub_move<int>), dim3(1), dim3(BLOCK), 0, s, 1LL, n_open, I.sub_idx, C.nd, C.nd_c, 0);
        hipLaunchKernelGGL((k_sub_move<double>), dim3(1), dim3(BLOCK), 0, s, 1LL, n_open, I.sub_idx, C.tmax, C.tmax_c, 0);
        nd = C.nd_c; tmax = C.tmax_c;
      }
      std::swap(F1, F2);
      Gd = B.Gp; X = B.Xc; Ws = I.Wc; Fro = I.Froc;
      nb = n_open;
      packed = true;
      ++I.n_packed;
      mark(7);
    }
    // the next filter: T_m(t_r) = cosh(m acosh t_r) >= 10 res / tol, within the cap and the budget
    const double need = std::acosh(std::max(10.0 * res / tol, 2.0));
    const double per = std::acosh(std::max(tmin, 1.0 + 1e-9));
    int m = (int)std::ceil(need / per);
    if (m < 2) m = 2;
    const int m_max = m_lim;
    const bool ramp_filter = C.ramp >= 0;                 // (a block that is still being found keeps only its g lowest columns: the
                                                          //  others have to grow into guards first)
    if (C.ramp >= 0) {
      const int md = ramp_deg[C.ramp];
      m = C.ramp < 3 ? md : std::min(std::max(m, 2), md);   // (a block that is nearly there does not need the whole ramp's degrees)
      if (++C.ramp >= 6) C.ramp = -1;
      if (C.mults + m + 1 > budget_all) break;
    } else if (m > m_max) {                               // several filters: can the budget still hold them?
      const double outers = std::ceil(need / (per * m_max));
      if (C.mults + outers * (m_max + 1) > (C.cold ? budget_all : budget)) {
        if (dbg) fprintf(stderr, "[sipx rank] filtered subspace: %g more products needed, over the budget\n", outers * (m_max + 1));
        break;
      }
      m = m_max;
    } else if (C.mults + m + 1 > (C.cold ? budget_all : budget)) break;
    // Y_0 = X, Y_1 = (2/a) P G X - X with G X = F2 already there; Y_{i+1} = (4/a) P G Y_i - 2 Y_i - Y_{i-1}.  X stays (the
    // projections need it); products go to F1, the iterates alternate between F2 and A, each new one over the one two steps back
    TS *Y0 = X, *Y1 = X;
    for (int i = 1; i <= m; ++i) {
      TS* Z = F2;
      if (i > 1) {
        blas_check(gemm_sbx(I.blas, N_, N_, k, b, k, TS(1), Gd, k, sG, Y1, k, sX, TS(0), F1, k, sX, nb, 1), "G Y");
        ++C.mults;
        Z = F1;
        mark(1);
      }
      TS* out = i == 1 ? F2 : (i == 2 ? A : Y0);
      if (nl <= 2 && fused_proj) {
        mark(5);
        hipLaunchKernelGGL((k_cheb_step_proj<TS>), dim3((unsigned)(((long long)b * nb + 3) / 4)), dim3(256), 0, s, k, b, g, r, nl, nb, Ws, X, Z, Y1, Y0,
                           out, i == 1 ? 1 : 0, nd);
        mark(6);
        Y0 = Y1;
        Y1 = out;
        continue;
      }
      if (nl > 0) {
        const TS* XL = X + (long long)(b - nl) * k;
        blas_check(gemm_sbx(I.blas, T_, N_, nl, b, k, TS(1), XL, k, sX, Z, k, sX, TS(0), B.Cs, b, sH, nb, 2), "X_L' Z");
        hipLaunchKernelGGL((k_cheb_mask<TS>), dim3((unsigned)std::min<long long>(NB, ((long long)nl * b * nb + 255) / 256)), dim3(256), 0, s, b, g, r, nl, nb,
                           Ws, B.Cs, nd);
        blas_check(gemm_sbx(I.blas, N_, N_, k, b, nl, TS(-1), XL, k, sX, B.Cs, b, sH, TS(1), Z, k, sX, nb), "Z - X_L C");
      }
      mark(5);
      hipLaunchKernelGGL((k_cheb_step<TS>), dim3(NB), dim3(BLOCK), 0, s, k, b, g, r, nb, Ws, Z, Y1, Y0, out, i == 1 ? 1 : 0, nd);
      mark(6);
      Y0 = Y1;
      Y1 = out;
    }
    m_prev = m;
    if (Y1 != A) std::swap(A, F2);                       // the filtered block is the one to orthonormalise next
    // the g lowest columns sit inside the damped interval: T_m there is anything in [-1, 1], also (nearly) zero, and such a
    // column would be nothing but what leaked in from above -- dependent on the other columns.  They stay what they were.
    if (KN.keep_damped && !ramp_filter)
      hipLaunchKernelGGL((k_cheb_keep<TS>), dim3(NB), dim3(BLOCK), 0, s, k, b, g, r, nb, Ws, X, A, nd);
    else if (g > 0)
      SIPX_HIP(hipMemcpy2DAsync(A, sizeof(TS) * (size_t)sX, X, sizeof(TS) * (size_t)sX, sizeof(TS) * (size_t)k * g, nb,
                                hipMemcpyDeviceToDevice, s));
  }
  return CHEB_GIVE_UP;
}

// Rank projection, Gram route: the top-r invariant subspace of every G_l from the Ritz vectors of the previous call (I.Xs[w]),
// by Rayleigh-Ritz steps with a Chebyshev filter between them (kernels above).  One multiplication with G per filter degree
// and one per Rayleigh-Ritz step; the degree of every filter is chosen from the residual still to be removed and the
// flattest spectrum of the batch, T_m(t_r) >= 10 residual / tolerance.  Accepted when every top-r pair has a residual below
// its level (k_sub_residual) AND nothing above theta_r can hide outside the block: the energy bound of k_sub_residual where the
// spectrum decays, the inertia of G with the found pairs removed (one batched Cholesky factorisation) where it is flat.
// Returns false -- the caller then decomposes fully -- when the budget of multiplications cannot suffice, a factorisation
// fails or the certificate does not hold.
template <typename T>
static bool rank_cheb_route(ExtImpl<T>& I, int w, int k, bool& cheap_fail, bool cold = false) {
  hipStream_t s = I.stream;
  const int b = I.sub_b, r = I.r, batch = I.batch;
  const double one = 1.0;
  const long long sG = (long long)k * k, sX = (long long)k * b;
  const auto N_ = rocblas_operation_none, T_ = rocblas_operation_transpose;
  const auto& KN = I.knobs;
  const int dbg = KN.dbg;
  ChebCtl C;
  C.w = w; C.k = k; C.cold = cold;
  C.ramp = cold ? 0 : -1;
  C.max_outer = cold ? 24 : 9;
  C.W = I.Ws;
  C.t_start = C.t_mark = std::chrono::steady_clock::now();
  auto mark = [&](int which) {
    if (dbg < 2) return;
    SIPX_HIP(hipStreamSynchronize(s));
    const auto now = std::chrono::steady_clock::now();
    C.ph[which] += std::chrono::duration<double, std::milli>(now - C.t_mark).count();
    C.t_mark = now;
  };
  RouteBufs<double> B64{I.Gd, I.Bd, I.Xs[w], I.Qs, I.Ys, I.Zs, I.Xc, I.Cs, I.Hs};
  double* X = I.Xs[w];
  // Float32 loop (see above): a warm call whose previous call has left Ritz values (Wprev) to tell which pairs are far above the rest
  const bool want32 = I.G32 != nullptr && KN.f32 && KN.eps_bw > 0 && !cold && I.have_wprev[w];
  if (want32) sub_fro(s, k, batch, I.Gd, I.FroPart, I.Fro);      // (the Float64 loop takes the norms below, once)
  int rc = CHEB_GIVE_UP;
  bool done32 = false;
  if (want32) {
    const double zero = 0.0;
    const long long sT = (long long)k * DEFL_N;
    double* Wp = I.Wprev[w];
    double *Xtop = I.Xtop, *Ytop = I.Ytop;
    // the top block of the previous call, refined on THIS call's matrices: two steps of block power iteration (whatever is far above
    // the rest converges by its ratio per step: 1e-5 for the constant part of a velocity slice), then one more product for the
    // Rayleigh quotients and the residuals the verdict is taken from -- three skinny products, each reads the Gram matrices once
    hipLaunchKernelGGL(k_defl_top, dim3(NB), dim3(BLOCK), 0, s, k, b, r, batch, Wp, X, I.nd32, Xtop);
    for (int step = 0; step < 3; ++step) {
      blas_check(rocblas_dgemm_strided_batched(I.blas, N_, N_, k, DEFL_N, k, &one, I.Gd, k, sG, Xtop, k, sT, &zero, Ytop, k, sT, batch), "G X_top");
      if (step < 2) hipLaunchKernelGGL(k_defl_gs, dim3(batch), dim3(BLOCK), 0, s, k, batch, Ytop, Xtop);
    }
    SIPX_HIP(hipMemsetAsync(I.sub_res, 0, 8 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_defl_plan, dim3(batch), dim3(BLOCK), 0, s, k, b, r, batch, Wp, Ytop, Xtop, KN.eps_bw, KN.tol, I.nd32, I.theta32, I.tmax32, I.sub_res);
    SIPX_HIP(hipMemcpyAsync(I.sub_res_host, I.sub_res, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    SIPX_HIP(hipStreamSynchronize(s));
    const bool use32 = (I.sub_res_host[1] & 64ull) == 0;
    if (dbg) fprintf(stderr, "[sipx rank] Float32 loop on the deflated matrices: %s, %.2f ms\n",
                     use32 ? "yes" : "no (a pair to deflate is not accurate, too many of them, or a level below Float32's floor)",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - C.t_start).count());
    mark(7);
    if (use32) {
      hipLaunchKernelGGL(k_defl_build, dim3(NB), dim3(BLOCK), 0, s, k, batch, I.Gd, I.theta32, Xtop, I.nd32, I.G32);
      hipLaunchKernelGGL(k_defl_convert, dim3(NB), dim3(BLOCK), 0, s, k, b, batch, X, Xtop, I.nd32, I.A32);
      mark(7);
      RouteBufs<float> B32{I.G32, I.Gp32, I.X32, I.A32, I.F32a, I.F32b, I.Xc32, I.Cs32, I.Zs32};
      ChebCtl C32 = C;
      C32.nd = I.nd32; C32.tmax = I.tmax32; C32.nd_c = I.nd32c; C32.tmax_c = I.tmax32c;
      C32.W = I.W32;
      ++I.n_f32;
      rc = cheb_loop<T, float>(I, B32, C32);
      C.mults = C32.mults;
      for (int q = 0; q < 8; ++q) C.ph[q] = C32.ph[q];
      C.t_mark = C32.t_mark;
      C.hidden = true;
      if (rc == CHEB_CONVERGED) {
        hipLaunchKernelGGL(k_defl_merge, dim3(NB), dim3(BLOCK), 0, s, k, b, batch, I.W32, I.X32, I.nd32, I.theta32, Xtop, I.Ws, X);
        mark(7);
        done32 = true;
      } else {
        // (a start that said too little, a stall in Float32, a failed factorisation: the Float64 loop starts over from the previous
        //  call's vectors, which are untouched)
        if (dbg) fprintf(stderr, "[sipx rank] the Float32 loop gave up (residual %.3e): the Float64 loop takes the call\n", C32.last_res);
        ++I.n_f32_back;
      }
    }
  }
  if (!done32) {
    sub_fro(s, k, batch, I.Gd, I.FroPart, I.Fro);
    SIPX_HIP(hipMemcpyAsync(B64.A, X, sizeof(double) * (size_t)sX * batch, hipMemcpyDeviceToDevice, s));
    mark(7);
    C.W = I.Ws;
    rc = cheb_loop<T, double>(I, B64, C);
  }
  cheap_fail = C.cheap_fail;
  bool ok = false;
  if (rc == CHEB_CONVERGED && !C.hidden) {
    ok = true;                        // a spectrum that decays behind the block: the energy bound has certified the pairs
  } else if (rc == CHEB_CONVERGED) {
    double* F1 = I.Ys;
    const bool own_cert = KN.own_cert;
    // flat spectrum: the inertia certificate (X_r Theta_r goes through F1)
    hipLaunchKernelGGL(k_cert_shift, dim3(NB), dim3(BLOCK), 0, s, k, b, r, batch, I.Gd, I.Ws, I.Bd);
    hipLaunchKernelGGL(k_cert_scale, dim3(NB), dim3(BLOCK), 0, s, k, b, r, batch, X, I.Ws, F1);
    blas_check(gemm_sbx(I.blas, N_, T_, k, k, r, 1.0, F1 + (long long)(b - r) * k, k, sX, X + (long long)(b - r) * k, k, sX, 1.0, I.Bd, k, sG, batch, 3),
               "certificate: rank-r term");
    if (own_cert) rank_cert_factor<T>(I, k);
    else blas_check(rocsolver_dpotrf_strided_batched(I.blas, rocblas_fill_upper, k, I.Bd, k, sG, I.info, batch), "certificate: potrf");
    SIPX_HIP(hipMemsetAsync(I.sub_res, 0, 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_cert_or, dim3((batch + 63) / 64), dim3(64), 0, s, batch, I.info, I.sub_res);
    SIPX_HIP(hipMemcpyAsync(I.sub_res_host, I.sub_res, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    SIPX_HIP(hipStreamSynchronize(s));
    ok = (I.sub_res_host[1] & 32ull) == 0;
    mark(7);
    if (KN.cert_check) {
      // the blocked factorisation against the library's, matrix by matrix, on the certificate's own matrices and on matrices
      // that cannot be definite (tests)
      for (int low = 0; low < 2; ++low) {
        std::vector<rocblas_int> v[2];
        for (int lib = 0; lib < 2; ++lib) {
          hipLaunchKernelGGL(k_cert_shift, dim3(NB), dim3(BLOCK), 0, s, k, b, r, batch, I.Gd, I.Ws, I.Bd, low);
          blas_check(rocblas_dgemm_strided_batched(I.blas, N_, T_, k, k, r, &one, F1 + (long long)(b - r) * k, k, sX, X + (long long)(b - r) * k, k, sX,
                                                   &one, I.Bd, k, sG, batch), "certificate: rank-r term");
          if (lib) blas_check(rocsolver_dpotrf_strided_batched(I.blas, rocblas_fill_upper, k, I.Bd, k, sG, I.info, batch), "certificate: potrf");
          else rank_cert_factor<T>(I, k);
          v[lib].resize(batch);
          SIPX_HIP(hipStreamSynchronize(s));
          SIPX_HIP(hipMemcpy(v[lib].data(), I.info, sizeof(rocblas_int) * batch, hipMemcpyDeviceToHost));
        }
        int differ = 0, indef = 0;
        for (int l = 0; l < batch; ++l) { differ += (v[0][l] != 0) != (v[1][l] != 0); indef += v[1][l] != 0; }
        fprintf(stderr, "[sipx rank] certificate check (%s shift): %d of %d matrices not positive definite, the two factorisations differ on %d\n",
                low ? "low" : "the certificate's", indef, batch, differ);
        if (differ) throw std::runtime_error("internal: the blocked Cholesky of the inertia certificate and the library's disagree");
      }
    }
    if (dbg) fprintf(stderr, "[sipx rank] inertia certificate %s, %.2f ms\n", ok ? "holds" : "fails",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - C.t_start).count());
  }
  if (ok && I.Wprev[w]) {             // the next call's deflation is planned from these
    SIPX_HIP(hipMemcpyAsync(I.Wprev[w], I.Ws, sizeof(double) * (size_t)b * batch, hipMemcpyDeviceToDevice, s));
    I.have_wprev[w] = true;
  }
  if (dbg >= 2)
    fprintf(stderr, "[sipx rank] phases (ms): orthonormalise %.2f, products with G %.2f (%d), small products %.2f, Ritz solver %.2f, residual %.2f, "
                    "projections %.2f, recurrence %.2f, copy + certificate %.2f\n", C.ph[0], C.ph[1], C.mults, C.ph[2], C.ph[3], C.ph[4], C.ph[5], C.ph[6], C.ph[7]);
  I.n_products += C.mults;
  SIPX_HIP(hipGetLastError());
  return ok;
}

// v <- P(v) in place.  `feas` selects the independent warm-start state used for the feasibility estimate.
template <typename T>
void ExtProj<T>::project(T* v, bool feas, double* partials, T* maxpart, T* compact) {
  ExtImpl<T>& I = *impl_;
  const long long N = I.sp.G.N;
  hipStream_t s = I.stream;
  const int kind = I.sp.kind;
  if (kind == EXT_DFT_MASK) {       // x -> Re(F' (UB .* F x)); the unitary factors of F and F' cancel into 1/N
    hipLaunchKernelGGL((k_pack<T>), dim3(NB), dim3(BLOCK), 0, s, N, v, I.z);
    if (sizeof(T) == 4) fft_check(hipfftExecC2C(I.plan, (hipfftComplex*)I.z, (hipfftComplex*)I.z, HIPFFT_FORWARD), "forward");
    else fft_check(hipfftExecZ2Z(I.plan, (hipfftDoubleComplex*)I.z, (hipfftDoubleComplex*)I.z, HIPFFT_FORWARD), "forward");
    hipLaunchKernelGGL((k_cmask<T>), dim3(NB), dim3(BLOCK), 0, s, N, I.z, I.mag);
    if (sizeof(T) == 4) fft_check(hipfftExecC2C(I.plan, (hipfftComplex*)I.z, (hipfftComplex*)I.z, HIPFFT_BACKWARD), "inverse");
    else fft_check(hipfftExecZ2Z(I.plan, (hipfftDoubleComplex*)I.z, (hipfftDoubleComplex*)I.z, HIPFFT_BACKWARD), "inverse");
    hipLaunchKernelGGL((k_unpack_all<T>), dim3(NB), dim3(BLOCK), 0, s, N, I.z, v, (T)(1.0 / (double)N));
  } else if (kind == EXT_L1_DFT && I.real_fft) {
    ProjScalars<T>* ps = feas ? I.psf : I.ps;
    if (sizeof(T) == 4) fft_check(hipfftExecR2C(I.plan_r2c, (hipfftReal*)v, (hipfftComplex*)I.z), "forward (real)");
    else fft_check(hipfftExecD2Z(I.plan_r2c, (hipfftDoubleReal*)v, (hipfftDoubleComplex*)I.z), "forward (real)");
    hipLaunchKernelGGL((k_cabs_half<T>), dim3(NB), dim3(BLOCK), 0, s, I.Nh, I.nh1, I.ndup, I.z, I.mag);
    K<T>::proj_scalars_arr(s, N, I.mag, PX_L1, T(0), I.radius_raw, ps, partials, maxpart, compact, N);
    hipLaunchKernelGGL((k_csoft<T>), dim3(NB), dim3(BLOCK), 0, s, I.Nh, I.z, I.mag, ps);
    // (the inverse real transform may overwrite its input; its output goes through mag, free by now, so that v stays untouched
    //  when it already lies inside the ball)
    if (sizeof(T) == 4) fft_check(hipfftExecC2R(I.plan_c2r, (hipfftComplex*)I.z, (hipfftReal*)I.mag), "inverse (real)");
    else fft_check(hipfftExecZ2D(I.plan_c2r, (hipfftDoubleComplex*)I.z, (hipfftDoubleReal*)I.mag), "inverse (real)");
    hipLaunchKernelGGL((k_unpack_real<T>), dim3(NB), dim3(BLOCK), 0, s, N, I.mag, v, (T)(1.0 / (double)N), ps);
  } else if (kind == EXT_L1_DFT) {
    ProjScalars<T>* ps = feas ? I.psf : I.ps;
    hipLaunchKernelGGL((k_pack<T>), dim3(NB), dim3(BLOCK), 0, s, N, v, I.z);
    if (sizeof(T) == 4) fft_check(hipfftExecC2C(I.plan, (hipfftComplex*)I.z, (hipfftComplex*)I.z, HIPFFT_FORWARD), "forward");
    else fft_check(hipfftExecZ2Z(I.plan, (hipfftDoubleComplex*)I.z, (hipfftDoubleComplex*)I.z, HIPFFT_FORWARD), "forward");
    hipLaunchKernelGGL((k_cabs<T>), dim3(NB), dim3(BLOCK), 0, s, N, I.z, I.mag);
    K<T>::proj_scalars_arr(s, N, I.mag, PX_L1, T(0), I.radius_raw, ps, partials, maxpart, compact, N);
    hipLaunchKernelGGL((k_csoft<T>), dim3(NB), dim3(BLOCK), 0, s, N, I.z, I.mag, ps);
    if (sizeof(T) == 4) fft_check(hipfftExecC2C(I.plan, (hipfftComplex*)I.z, (hipfftComplex*)I.z, HIPFFT_BACKWARD), "inverse");
    else fft_check(hipfftExecZ2Z(I.plan, (hipfftDoubleComplex*)I.z, (hipfftDoubleComplex*)I.z, HIPFFT_BACKWARD), "inverse");
    hipLaunchKernelGGL((k_unpack<T>), dim3(NB), dim3(BLOCK), 0, s, N, I.z, v, (T)(1.0 / (double)N), ps);
  } else if (kind == EXT_RANK || kind == EXT_NUCLEAR) {
    I.check_status();
    // Batched Jacobi SVD in float64 whatever TF is: rocSOLVER's gesvdj works on A'A (condition number squared), so
    // float32 input is widened first and the truncated product is rounded back once.  Measured ~2x faster than the
    // QR-iteration gesvd on 256 slices of 256x256 and as accurate as LAPACK on the float32 data.
    const int k = I.m < I.n ? I.m : I.n;
    if (kind == EXT_RANK && I.r >= k) return;              // nothing to truncate
    const long long sU = (long long)I.m * k, sV = (long long)k * I.n, sA = (long long)I.m * I.n;
    hipLaunchKernelGGL((k_seg_gather<T, double>), dim3(NB), dim3(BLOCK), 0, s, I.map, v, I.Ad);
    if (I.gram) {
      const double one = 1.0, zero = 0.0;
      const long long sG = (long long)k * k;
      const bool right = I.n <= I.m;          // eigenvectors of X'X (right singular vectors) or of XX' (left ones)
      const auto N_ = rocblas_operation_none, T_ = rocblas_operation_transpose;
      if (right)
        blas_check(gemm_sbx(I.blas, T_, N_, k, k, I.m, 1.0, I.Ad, I.m, sA, I.Ad, I.m, sA, 0.0, I.Gd, k, sG, I.batch, 1), "gram");
      else
        blas_check(gemm_sbx(I.blas, N_, T_, k, k, I.n, 1.0, I.Ad, I.m, sA, I.Ad, I.m, sA, 0.0, I.Gd, k, sG, I.batch, 1), "gram");
      // Rank projection: only the span of the top-r eigenvectors is needed, and it moves little from one PARSDMM
      // iteration to the next.  Block subspace iteration with Rayleigh-Ritz on b = r + 16 vectors, started from the
      // previous call's Ritz vectors, is accepted when every top-r pair has a residual below 1e-12 theta_max (far inside
      // Float32 resolution of the projected slice); otherwise -- first call, slow contraction, a failed factorisation --
      // the full decomposition below runs as before and provides the next warm start.
      const int w = feas ? 1 : 0;
      bool sub_ok = false;
      const int b = I.sub_b;
      const long long sX = (long long)k * b, sH = (long long)b * b;
      if (b > 0 && I.cheb && (I.sub_have[w] || I.knobs.cold)) {
        // the filtered iteration does not need a gap behind the block; an attempt that failed costs its products on top of the
        // full decomposition, so the next attempts wait (1, 2, 4, ... calls)
        const bool cold = !I.sub_have[w];
        if (cold) {
          // No start (round 5): is there anything to project?  The first iteration of a solve from zero hands over v = 0
          // (rhs = 0, x = 0: PARSDMM.jl:101-107 with y = l = 0) -- rounds 3-4 decomposed 512 zero matrices for 119 ms and kept
          // their arbitrary eigenvectors as the next start.  P(0) = 0: v stays as it is, the state stays cold.
          SIPX_HIP(hipMemsetAsync(I.sub_res, 0, 8 * sizeof(unsigned long long), s));
          sub_fro(s, k, I.batch, I.Gd, I.FroPart, I.Fro, I.sub_res);
          SIPX_HIP(hipMemcpyAsync(I.sub_res_host, I.sub_res, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
          SIPX_HIP(hipStreamSynchronize(s));
          if (I.sub_res_host[7] == 0ull) {
            if (I.knobs.dbg) fprintf(stderr, "[sipx rank] every slice is zero: nothing to project\n");
            ++I.n_calls; ++I.n_subspace;
            return;
          }
          hipLaunchKernelGGL(k_sub_seed, dim3(NB), dim3(BLOCK), 0, s, k, b, I.batch, I.Xs[w]);
        }
        if (I.cheb_skip[w] > 0 && !cold) {
          --I.cheb_skip[w];
        } else {
          // (an attempt given up at its first Rayleigh-Ritz step -- the start said too little about this input, the second
          //  iteration of a solve -- has cost a fiftieth of a decomposition: the next call simply tries again)
          bool cheap_fail = false;
          sub_ok = rank_cheb_route<T>(I, w, k, cheap_fail, cold);
          if (sub_ok) I.sub_have[w] = true;
          if (sub_ok) I.cheb_fails[w] = 0;
          // (the decomposition that follows a failure leaves exact vectors behind, the best start there is: the next call tries
          //  again; only a second failure in a row makes calls sit out -- 1, 2, 4 ...)
          else if (!cheap_fail) { I.cheb_skip[w] = I.cheb_fails[w] >= 1 ? 1 << std::min(I.cheb_fails[w] - 1, 5) : 0; ++I.cheb_fails[w]; }
          if (I.knobs.dbg) fprintf(stderr, "[sipx rank] %s\n", sub_ok ? "subspace accepted" : "full decomposition");
        }
      } else if (b > 0 && I.sub_have[w] && I.sub_try[w]) {
        const int dbg = I.knobs.dbg;
        const int max_it = 8;
        const double tol = 1e-12;
        double prev = -1;
        double* X = I.Xs[w];
        sub_fro(s, k, I.batch, I.Gd, I.FroPart, I.Fro);
        for (int it = 0; it < max_it; ++it) {
          blas_check(gemm_sbx(I.blas, N_, N_, k, b, k, 1.0, I.Gd, k, sG, X, k, sX, 0.0, I.Qs, k, sX, I.batch, 1), "G X");
          hipLaunchKernelGGL(k_sub_normalize, dim3(I.batch), dim3(BLOCK), 0, s, k, b, I.batch, I.Qs);
          blas_check(gemm_sbx(I.blas, T_, N_, b, b, k, 1.0, I.Qs, k, sX, I.Qs, k, sX, 0.0, I.Hs, b, sH, I.batch, 1), "Y'Y");
          blas_check(rocsolver_dpotrf_strided_batched(I.blas, rocblas_fill_upper, b, I.Hs, b, sH, I.info, I.batch), "potrf");
          blas_check(rocblas_dtrsm_strided_batched(I.blas, rocblas_side_right, rocblas_fill_upper, N_, rocblas_diagonal_non_unit, k, b, &one,
                                                   I.Hs, b, sH, I.Qs, k, sX, I.batch), "trsm");          // Qs: orthonormal basis
          blas_check(gemm_sbx(I.blas, N_, N_, k, b, k, 1.0, I.Gd, k, sG, I.Qs, k, sX, 0.0, I.Zs, k, sX, I.batch, 1), "G Q");
          blas_check(gemm_sbx(I.blas, T_, N_, b, b, k, 1.0, I.Qs, k, sX, I.Zs, k, sX, 0.0, I.Hs, b, sH, I.batch, 1), "Q'GQ");
          // b x b Ritz problem: one-kernel Jacobi (the divide-and-conquer driver applies its b-1 reflectors one launch at a
          // time, 50 ms for 256 matrices of 48 x 48)
          blas_check(rocsolver_dsyevj_strided_batched(I.blas, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper, b, I.Hs,
                                                      b, sH, 0.0, I.Es, 100, I.info + 2 * I.batch, I.Ws, b, I.info + I.batch, I.batch),
                     "syevj (Ritz)");
          blas_check(gemm_sbx(I.blas, N_, N_, k, b, b, 1.0, I.Qs, k, sX, I.Hs, b, sH, 0.0, X, k, sX, I.batch, 1), "Q Z");
          blas_check(gemm_sbx(I.blas, N_, N_, k, b, b, 1.0, I.Zs, k, sX, I.Hs, b, sH, 0.0, I.Qs, k, sX, I.batch, 1), "(GQ) Z");
          SIPX_HIP(hipMemsetAsync(I.sub_res, 0, 2 * sizeof(unsigned long long), s));
          hipLaunchKernelGGL(k_sub_residual, dim3(I.batch), dim3(BLOCK), 0, s, k, b, I.r, I.batch, I.Qs, X, I.Ws, b, I.info,
                             I.info + I.batch, I.Fro, I.sub_res);
          SIPX_HIP(hipMemcpyAsync(I.sub_res_host, I.sub_res, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
          SIPX_HIP(hipStreamSynchronize(s));
          double res;
          std::memcpy(&res, &I.sub_res_host[0], sizeof(double));
          const bool failed = (I.sub_res_host[1] & 15ull) != 0;
          const bool certified = (I.sub_res_host[1] & 16ull) == 0;
          if (dbg) fprintf(stderr, "[sipx rank] subspace it %d: residual %.3e fail-bits %llu\n", it + 1, res, I.sub_res_host[1]);
          if (failed) break;
          if (res <= tol) { sub_ok = certified; break; }     // converged pairs that might not be the largest: decompose fully
          if (prev > 0) {                       // contraction observed so far: give up when the budget cannot suffice
            const double c = res / prev;
            if (!(c < 1.0)) break;
            const double need = std::log(tol / res) / std::log(c);
            if (need > (double)(max_it - 1 - it)) break;
          }
          prev = res;
        }
        if (dbg) fprintf(stderr, "[sipx rank] %s\n", sub_ok ? "subspace accepted" : "full decomposition");
      }
      ++I.n_calls;
      if (sub_ok) ++I.n_subspace; else ++I.n_full;
      const double* Esel;
      long long ldsel_stride;
      if (sub_ok) {
        Esel = I.Xs[w] + (long long)(b - I.r) * k;       // Ritz values ascend: the last r columns span the top-r space
        ldsel_stride = sX;
      } else {
        blas_check(rocsolver_dsyevd_strided_batched(I.blas, rocblas_evect_original, rocblas_fill_upper, k, I.Gd, k, sG, I.Wd, k, I.Ed, k,
                                                    I.info, I.batch),
                   "syevd");
        I.note_status(s);
        Esel = I.Gd + (long long)(k - I.r) * k;           // eigenvalues ascend: the last r columns span the top-r space
        ldsel_stride = sG;
        if (b > 0) {
          // keep the top-b eigenvectors as the next warm start, and decide from the spectrum whether to use them: the
          // iteration contracts by theta_{b+1} / theta_r per step, so a truncation inside a flat part of the spectrum
          // (ratio near 1) would never get there and the attempt is not made
          hipLaunchKernelGGL(k_sub_keep, dim3(NB), dim3(BLOCK), 0, s, k, b, I.batch, I.Gd, I.Xs[w], I.info);
          SIPX_HIP(hipMemsetAsync(I.sub_res, 0, 2 * sizeof(unsigned long long), s));
          hipLaunchKernelGGL(k_sub_ratio, dim3((I.batch + 63) / 64), dim3(64), 0, s, k, b, I.r, I.batch, I.Wd, I.sub_res);
          SIPX_HIP(hipMemcpyAsync(I.sub_res_host, I.sub_res, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
          SIPX_HIP(hipStreamSynchronize(s));
          double q;
          std::memcpy(&q, &I.sub_res_host[0], sizeof(double));
          I.sub_have[w] = true;
          I.have_wprev[w] = false;          // (the Ritz values of the last filtered call no longer describe the block)
          if (I.cheb && !I.cert_warm && !I.knobs.own_cert) {
            // the library sizes the workspace of a batched factorisation at its first call (a device allocation of its own, 100 ms
            // and more): spend it here, behind a full decomposition, not inside the first accepted call of the filtered route
            I.cert_warm = true;
            SIPX_HIP(hipMemsetAsync(I.Bd, 0, sizeof(double) * (size_t)k * k * I.batch, s));
            (void)rocsolver_dpotrf_strided_batched(I.blas, rocblas_fill_upper, k, I.Bd, k, sG, I.info, I.batch);
          }
          I.sub_try[w] = q < 0.25;
          if (I.knobs.dbg) fprintf(stderr, "[sipx rank] theta_{b+1}/theta_r = %.3e -> %s\n", q, I.sub_try[w] ? "subspace next" : "full next");
        }
      }
      const int* flag = nullptr;
      int inner = I.r;
      const double* Escl = Esel;
      if (kind == EXT_NUCLEAR) {
        hipLaunchKernelGGL(k_nuc_factors, dim3((I.batch + 63) / 64), dim3(64), 0, s, k, I.batch, I.sp.pmax, I.Wd, I.Sd, I.flag);
        hipLaunchKernelGGL(k_scale_eigvecs, dim3(NB), dim3(BLOCK), 0, s, k, I.batch, I.Gd, I.Sd, I.Gs);
        flag = I.flag;
        inner = k;
        Esel = I.Gd;
        Escl = I.Gs;
        ldsel_stride = sG;
      }
      if (right) {       // X <- (X * Escl) * Esel'
        blas_check(gemm_sbx(I.blas, N_, N_, I.m, inner, k, 1.0, I.Ad, I.m, sA, Escl, k, ldsel_stride, 0.0, I.Ud, I.m, sU, I.batch, 1), "gemm X V");
        blas_check(gemm_sbx(I.blas, N_, T_, I.m, I.n, inner, 1.0, I.Ud, I.m, sU, Esel, k, ldsel_stride, 0.0, I.Ad, I.m, sA, I.batch, 1), "gemm (XV) V'");
      } else {           // X <- Escl * (Esel' * X)
        blas_check(gemm_sbx(I.blas, T_, N_, inner, I.n, k, 1.0, Esel, k, ldsel_stride, I.Ad, I.m, sA, 0.0, I.Vd, k, sV, I.batch, 1), "gemm U' X");
        blas_check(gemm_sbx(I.blas, N_, N_, I.m, I.n, inner, 1.0, Escl, k, ldsel_stride, I.Vd, k, sV, 0.0, I.Ad, I.m, sA, I.batch, 1), "gemm U (U'X)");
      }
      hipLaunchKernelGGL((k_seg_scatter<T, double>), dim3(NB), dim3(BLOCK), 0, s, I.map, I.Ad, v, flag);
      SIPX_HIP(hipGetLastError());
      return;
    }
    blas_check(rocsolver_dgesvdj_strided_batched(I.blas, rocblas_svect_singular, rocblas_svect_singular, I.m, I.n, I.Ad,
                                                 I.m, sA, 0.0, I.Ed, 100, I.info + I.batch, I.Sd, k, I.Ud, I.m, sU, I.Vd,
                                                 k, sV, I.info, I.batch),
               "gesvdj");
    I.note_status(s, I.Ed, I.Sd, k);
    int inner = I.r;
    const int* flag = nullptr;
    if (kind == EXT_NUCLEAR) {     // slices already inside the ball keep their values bit for bit (flag = 0)
      hipLaunchKernelGGL(k_nuc_shrink, dim3((I.batch + 63) / 64), dim3(64), 0, s, k, I.batch, I.sp.pmax, I.Sd, I.flag);
      inner = k;
      flag = I.flag;
    }
    hipLaunchKernelGGL((k_scale_cols<double>), dim3(NB), dim3(BLOCK), 0, s, I.m, inner, I.m, sU, (long long)k, I.batch, I.Ud, I.Sd);
    const double one = 1.0, zero = 0.0;
    blas_check(rocblas_dgemm_strided_batched(I.blas, rocblas_operation_none, rocblas_operation_none, I.m, I.n, inner, &one,
                                             I.Ud, I.m, sU, I.Vd, k, sV, &zero, I.Ad, I.m, sA, I.batch),
               "gemm");
    hipLaunchKernelGGL((k_seg_scatter<T, double>), dim3(NB), dim3(BLOCK), 0, s, I.map, I.Ad, v, flag);
  } else if (kind == EXT_DCT) {
    const Grid& G = I.sp.G;
    const int n1 = (int)G.n[0], n2 = (int)G.n[1], n3 = (int)G.n[2];
    const auto N_ = rocblas_operation_none, T_ = rocblas_operation_transpose;
    const long long s12 = (long long)n1 * n2;
    // forward: coefficients = C1 X C2' (per z plane) ... C3' ; ping-pong between v / W1 / W2
    blas_check(gemm_T(I.blas, N_, N_, n1, n2 * n3, n1, I.Cm[0], n1, v, n1, I.W1, n1), "dct dim 1");
    T *cur = I.W1, *oth = I.W2;
    if (n2 > 1) {
      blas_check(gemm_sb(I.blas, N_, T_, n1, n2, n2, cur, n1, s12, I.Cm[1], n2, 0, oth, n1, s12, n3), "dct dim 2");
      std::swap(cur, oth);
    }
    if (n3 > 1) {
      blas_check(gemm_T(I.blas, N_, T_, (int)s12, n3, n3, cur, (int)s12, I.Cm[2], n3, oth, (int)s12), "dct dim 3");
      std::swap(cur, oth);
    }
    // the inner projector on the coefficient array
    const int in = I.sp.inner;
    const bool two = in == SIPX_PROJ_L1 || in == SIPX_PROJ_CARDINALITY;
    ProjScalars<T>* ps = two ? (feas ? I.psf : I.ps) : nullptr;
    Grid g1;
    g1.n[0] = N; g1.n[1] = 1; g1.n[2] = 1; g1.N = N; g1.st[0] = 1; g1.st[1] = N; g1.st[2] = N;
    if (two) K<T>::proj_scalars_arr(s, N, cur, in, (T)I.sp.pmin, (T)I.sp.pmax, ps, partials, maxpart, compact, N);
    proj_apply_grid<T>(s, g1, 0, nullptr, N, cur, in, in == SIPX_PROJ_BOUNDS_VEC ? T(0) : (T)I.sp.pmin, (T)I.sp.pmax, I.dlb, I.dub, ps);
    // inverse (transposed matrices, reverse order)
    if (n3 > 1) {
      blas_check(gemm_T(I.blas, N_, N_, (int)s12, n3, n3, cur, (int)s12, I.Cm[2], n3, oth, (int)s12), "idct dim 3");
      std::swap(cur, oth);
    }
    if (n2 > 1) {
      blas_check(gemm_sb(I.blas, N_, N_, n1, n2, n2, cur, n1, s12, I.Cm[1], n2, 0, oth, n1, s12, n3), "idct dim 2");
      std::swap(cur, oth);
    }
    blas_check(gemm_T(I.blas, T_, N_, n1, n2 * n3, n1, I.Cm[0], n1, cur, n1, oth, n1), "idct dim 1");
    // C'C = I: inside the l1 ball the round trip only adds rounding noise -- v is kept bit for bit there (as for the DFT)
    hipLaunchKernelGGL((k_copy_if_needed<T>), dim3(NB), dim3(BLOCK), 0, s, N, oth, v,
                       in == SIPX_PROJ_L1 ? (const ProjScalars<T>*)ps : (const ProjScalars<T>*)nullptr);
  } else if (kind == EXT_CARD_SEG) {
    const long long nb = I.map.nseg < (long long)NB * 4 ? I.map.nseg : (long long)NB * 4;
    hipLaunchKernelGGL((k_seg_card<T>), dim3((unsigned)nb), dim3(BLOCK), 0, s, I.map, v, (long long)I.sp.pmax);
  } else if (kind == EXT_HISTOGRAM) {
    const long long M = I.map.L;
    hipLaunchKernelGGL((k_hist_gather<T>), dim3(NB), dim3(BLOCK), 0, s, I.map, v, I.keys_in, I.idx_in);
    SIPX_HIP(hipcub::DeviceRadixSort::SortPairs(I.sort_tmp, I.sort_bytes, I.keys_in, I.keys_out, I.idx_in, I.idx_out, (int)M, 0,
                                                (int)sizeof(T) * 8, s));
    hipLaunchKernelGGL((k_hist_apply<T>), dim3(NB), dim3(BLOCK), 0, s, I.map, I.keys_out, I.idx_out, I.lb, I.ub, v);
  } else if (kind == EXT_SUBSPACE) {       // x .= A*(A'*x)  or  A*((A'*A)\(A'*x))   project_subspace!.jl:15-19
    const int L = (int)I.map.L, ns = (int)I.map.nseg, r = I.cols;
    hipLaunchKernelGGL((k_seg_gather<T, T>), dim3(NB), dim3(BLOCK), 0, s, I.map, v, I.X);
    blas_check(gemm_T(I.blas, rocblas_operation_transpose, rocblas_operation_none, r, ns, L, I.basis, L, I.X, L, I.t1, r), "gemm A'x");
    T* t = I.t1;
    if (I.gram_inv) {
      blas_check(gemm_T(I.blas, rocblas_operation_none, rocblas_operation_none, r, ns, r, I.gram_inv, r, I.t1, r, I.t2, r), "gemm G t");
      t = I.t2;
    }
    blas_check(gemm_T(I.blas, rocblas_operation_none, rocblas_operation_none, L, ns, r, I.basis, L, t, r, I.X, L), "gemm A t");
    hipLaunchKernelGGL((k_seg_scatter<T, T>), dim3(NB), dim3(BLOCK), 0, s, I.map, I.X, v, (const int*)nullptr);
  }
  SIPX_HIP(hipGetLastError());
}

template <typename T>
void ExtProj<T>::set_stream(hipStream_t s) {
  ExtImpl<T>& I = *impl_;
  if (I.stream == s) return;
  I.stream = s;
  if (I.blas) blas_check(rocblas_set_stream(I.blas, s), "set stream");
  if (I.have_plan) fft_check(hipfftSetStream(I.plan, s), "set stream");
  if (I.plan_r2c) fft_check(hipfftSetStream(I.plan_r2c, s), "set stream");
  if (I.plan_c2r) fft_check(hipfftSetStream(I.plan_c2r, s), "set stream");
}

// Back to the state of a freshly built projector (sipx_reset): no warm start of any kind, no pending status, counters at zero.
// A reused context then walks through exactly the calls of a new one -- same bits -- without its allocations, plans and handles.
template <typename T>
void ExtProj<T>::reset() {
  if (!impl_) return;
  ExtImpl<T>& I = *impl_;
  if (I.fail_pending) { I.fail_pending = false; (void)hipEventSynchronize(I.fail_ev); }
  if (I.ps) K<T>::ps_init(I.stream, I.ps, I.cidx);
  if (I.psf) K<T>::ps_init(I.stream, I.psf, I.cidx);
  for (int w = 0; w < 2; ++w) { I.sub_have[w] = I.sub_try[w] = false; I.cheb_skip[w] = I.cheb_fails[w] = 0; I.have_wprev[w] = false; }
  I.n_calls = I.n_subspace = I.n_full = I.n_products = 0;
  I.n_packed = 0;
  I.n_f32 = I.n_f32_back = 0;
}

template <typename T>
void ExtProj<T>::route_counts(long long out[6]) const {
  out[0] = impl_ ? impl_->n_calls : 0;
  out[1] = impl_ ? impl_->n_subspace : 0;
  out[2] = impl_ ? impl_->n_full : 0;
  out[3] = impl_ ? impl_->n_products : 0;
  out[4] = impl_ ? impl_->n_f32 : 0;
  out[5] = impl_ ? impl_->n_f32_back : 0;
}

template class ExtProj<float>;
template class ExtProj<double>;
template void ext_dist2<float>(hipStream_t, long long, const float*, const float*, double*);
template void ext_dist2<double>(hipStream_t, long long, const double*, const double*, double*);

}  // namespace sipx
