// Scalars of the non-elementwise projectors, computed entirely on the device.
//
// Replaces (reference file:line):
//   project_l1_Duchi!  src/projectors/project_l1_Duchi!.jl:21-52  -- the reference sorts all M
//       magnitudes (RadixSort/QuickSort), takes a cumsum and scans serially for the threshold.
//       Here: ||v||_1 early exit, one (count,sum) histogram pass that brackets theta in a bin,
//       compaction of that bin, and Michelot's fixed-point iteration on the compacted values
//       by a single workgroup.  theta solves sum(max(|v|-theta,0)) = b exactly (float64).
//   project_l2!        src/projectors/project_l2!.jl:3-16
//   project_annulus!   src/projectors/project_annulus!.jl:3-21
#include <stdexcept>
#include <string>

#include "sipx_device.h"

namespace sipx {

constexpr int HIST_GRID = 512;

// ||v||_1, ||v||_2^2 (slots 0,1) and per-block max|v| of a stored vector.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_ps_reduce(long long len, const T* __restrict__ v,
                                                     double* __restrict__ partials, T* __restrict__ maxpart) {
  double acc[2] = {0, 0};
  T vmax = T(0);
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < len; e += (long long)NB * BLOCK) {
    const T x = v[e], av = fabs(x);
    acc[0] += (double)av;
    acc[1] += (double)x * (double)x;
    vmax = av > vmax ? av : vmax;
  }
  block_reduce_store<2>(acc, partials, 0);
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
template <typename T>
void K<T>::ps_reduce(hipStream_t s, long long len, const T* v, double* partials, T* maxpart) {
  hipLaunchKernelGGL((k_ps_reduce<T>), dim3(NB), dim3(BLOCK), 0, s, len, v, partials, maxpart);
  SIPX_HIP(hipGetLastError());
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_ps_finish(const double* __restrict__ partials, const T* __restrict__ maxpart,
                                                     ProjScalars<T>* ps, int prox, T pmin, T pmax, long long true_len) {
  const double asum = block_sum_partials(partials);
  const double sumsq = block_sum_partials(partials + NB);
  __shared__ T smax[BLOCK / 64];
  T vmax = T(0);
  for (int i = threadIdx.x; i < NB; i += BLOCK) vmax = maxpart[i] > vmax ? maxpart[i] : vmax;
  vmax = wave_max<T>(vmax);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = vmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < BLOCK / 64; ++i) vmax = smax[i] > vmax ? smax[i] : vmax;
    ps->asum = asum;
    ps->sumsq = sumsq;
    ps->vmax = vmax;
    ps->need = 0;
    ps->theta = T(0);
    ps->scale = T(1);
    ps->fill = 0;
    ps->tau = T(0);
    ps->quota = 0x7fffffffffffffffll;
    ps->n_compact = 0;
    if (prox == PX_L1) {
      ps->need = ((T)asum <= pmax) ? 0 : 1;              // norm(v,1) <= b && return v   project_l1_Duchi!.jl:23
    } else if (prox == PX_L2) {
      const T nl2 = (T)sqrt(sumsq);                      // project_l2!.jl:8-13
      if (!(nl2 <= pmax)) {
        ps->need = 1;
        ps->scale = pmax / nl2;
      }
    } else if (prox == PX_ANNULUS) {
      const T nl2 = (T)sqrt(sumsq);                      // project_annulus!.jl:9-18
      if (pmin <= nl2 && nl2 <= pmax) {
      } else if (nl2 > pmax) {
        ps->need = 1;
        ps->scale = pmax / nl2;
      } else if (nl2 < pmin && nl2 > T(0)) {
        ps->need = 1;
        ps->scale = pmin / nl2;
      } else if (nl2 < pmin && nl2 == T(0)) {
        ps->need = 1;
        ps->fill = 1;                                    // sigma_min ./ sqrt(length(x)): Float64 sqrt of an Int
        ps->scale = (T)((double)pmin / sqrt((double)true_len));
      }
    }
  }
}
template <typename T>
void K<T>::ps_finish(hipStream_t s, const double* partials, const T* maxpart, ProjScalars<T>* ps, int prox, T pmin,
                     T pmax, long long true_len) {
  hipLaunchKernelGGL((k_ps_finish<T>), dim3(1), dim3(BLOCK), 0, s, partials, maxpart, ps, prox, pmin, pmax, true_len);
  SIPX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// l1-ball threshold
template <typename T>
__device__ __forceinline__ int l1_bin(T av, double scale) {
  int b = (int)((double)av * scale);
  return b > L1_BINS - 1 ? L1_BINS - 1 : b;
}

// (count, sum) histogram of the non-zero magnitudes over L1_BINS linear bins on [0, vmax].
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_l1_hist(long long len, const T* __restrict__ v,
                                                   const ProjScalars<T>* __restrict__ ps,
                                                   unsigned long long* __restrict__ gcnt, double* __restrict__ gsum) {
  if (!ps->need) return;
  __shared__ unsigned int cnt[L1_BINS];
  __shared__ double sm[L1_BINS];
  for (int i = threadIdx.x; i < L1_BINS; i += BLOCK) {
    cnt[i] = 0;
    sm[i] = 0;
  }
  __syncthreads();
  const double scale = (double)L1_BINS / (double)ps->vmax;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < len; e += (long long)HIST_GRID * BLOCK) {
    const T av = fabs(v[e]);
    if (av > T(0)) {
      const int b = l1_bin<T>(av, scale);
      atomicAdd(&cnt[b], 1u);
      atomicAdd(&sm[b], (double)av);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L1_BINS; i += BLOCK) {
    if (cnt[i]) {
      atomicAdd(&gcnt[i], (unsigned long long)cnt[i]);
      atomicAdd(&gsum[i], sm[i]);
    }
  }
}

// Finds the bin [e_k, e_k+1) holding theta: the largest k with f(e_k) >= 0 where
// f(t) = sum_{|v|>t}(|v| - t) - b is evaluated exactly at bin edges from the suffix sums.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_l1_bracket(ProjScalars<T>* ps, T radius,
                                                      const unsigned long long* __restrict__ gcnt,
                                                      const double* __restrict__ gsum) {
  if (!ps->need) return;
  constexpr int PER = L1_BINS / BLOCK;
  __shared__ double ssum[BLOCK];
  __shared__ unsigned long long scnt[BLOCK];
  __shared__ int sbest[BLOCK];
  const int t = threadIdx.x;
  double ls = 0;
  unsigned long long lc = 0;
  for (int j = 0; j < PER; ++j) {
    ls += gsum[t * PER + j];
    lc += gcnt[t * PER + j];
  }
  ssum[t] = ls;
  scnt[t] = lc;
  __syncthreads();
  // suffix totals of the chunks strictly above this thread's chunk (256 entries: serial per thread is fine)
  double above_s = 0;
  unsigned long long above_c = 0;
  for (int u = BLOCK - 1; u > t; --u) {
    above_s += ssum[u];
    above_c += scnt[u];
  }
  const double width = (double)ps->vmax / (double)L1_BINS;
  int best = -1;
  double run_s = above_s;
  unsigned long long run_c = above_c;
  for (int j = PER - 1; j >= 0; --j) {
    const int k = t * PER + j;
    run_s += gsum[k];
    run_c += gcnt[k];
    const double f = run_s - (double)k * width * (double)run_c - (double)radius;
    if (f >= 0 && k > best) best = k;
  }
  sbest[t] = best;
  __syncthreads();
  if (t == 0) {
    int k = 0;
    for (int u = 0; u < BLOCK; ++u) k = sbest[u] > k ? sbest[u] : k;
    double sa = 0;
    unsigned long long ca = 0;
    for (int j = L1_BINS - 1; j > k; --j) {
      sa += gsum[j];
      ca += gcnt[j];
    }
    ps->bin = k;
    ps->s_above = sa;
    ps->c_above = (long long)ca;
    ps->lo = (double)k * width;
    ps->width = width;
    ps->n_compact = 0;
  }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_l1_compact(long long len, const T* __restrict__ v, ProjScalars<T>* ps,
                                                      T* __restrict__ compact) {
  if (!ps->need) return;
  const double scale = (double)L1_BINS / (double)ps->vmax;
  const int kb = ps->bin;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < len; e += (long long)HIST_GRID * BLOCK) {
    const T av = fabs(v[e]);
    if (av > T(0) && l1_bin<T>(av, scale) == kb) {
      const unsigned long long pos = atomicAdd(&ps->n_compact, 1ull);
      compact[pos] = av;
    }
  }
}

// Michelot's iteration restricted to the bracket bin: theta <- (S_above + S_in(>theta) - b) / (C_above + C_in(>theta)),
// monotone from the bin's lower edge, exact after finitely many steps (stops when the active count repeats).
template <typename T>
__global__ __launch_bounds__(1024) void k_l1_solve(ProjScalars<T>* ps, T radius, const T* __restrict__ compact) {
  if (!ps->need) return;
  __shared__ double ssum[16];
  __shared__ long long scnt[16];
  __shared__ double sh_theta;
  __shared__ int sh_done;
  const long long n = (long long)ps->n_compact;
  const double sa = ps->s_above, b = (double)radius;
  const long long ca = ps->c_above;
  double theta = ps->lo;
  long long cprev = -1;
  for (int it = 0; it < 128; ++it) {
    double s = 0;
    long long c = 0;
    for (long long e = threadIdx.x; e < n; e += 1024) {
      const double av = (double)compact[e];
      if (av > theta) {
        s += av;
        c += 1;
      }
    }
    s = wave_sum(s);
    double cd = wave_sum((double)c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      ssum[threadIdx.x >> 6] = s;
      scnt[threadIdx.x >> 6] = (long long)cd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double S = 0;
      long long C = 0;
      for (int i = 0; i < 16; ++i) {
        S += ssum[i];
        C += scnt[i];
      }
      const long long tot = ca + C;
      double tn = theta;
      if (tot > 0) tn = (sa + S - b) / (double)tot;
      sh_done = (C == cprev || tot == 0) ? 1 : 0;
      sh_theta = tn > theta ? tn : theta;
      scnt[0] = C;
    }
    __syncthreads();
    theta = sh_theta;
    cprev = scnt[0];
    if (sh_done) break;
  }
  if (threadIdx.x == 0) {
    const T th = (T)theta;
    ps->theta = th > T(0) ? th : T(0);       // theta = max(0, .)   project_l1_Duchi!.jl:46
  }
}

template <typename T>
void K<T>::l1_theta(hipStream_t s, long long len, const T* v, ProjScalars<T>* ps, T radius,
                    unsigned long long* hist_cnt, double* hist_sum, T* compact) {
  SIPX_HIP(hipMemsetAsync(hist_cnt, 0, sizeof(unsigned long long) * L1_BINS, s));
  SIPX_HIP(hipMemsetAsync(hist_sum, 0, sizeof(double) * L1_BINS, s));
  hipLaunchKernelGGL((k_l1_hist<T>), dim3(HIST_GRID), dim3(BLOCK), 0, s, len, v, ps, hist_cnt, hist_sum);
  hipLaunchKernelGGL((k_l1_bracket<T>), dim3(1), dim3(BLOCK), 0, s, ps, radius, hist_cnt, hist_sum);
  hipLaunchKernelGGL((k_l1_compact<T>), dim3(HIST_GRID), dim3(BLOCK), 0, s, len, v, ps, compact);
  hipLaunchKernelGGL((k_l1_solve<T>), dim3(1), dim3(1024), 0, s, ps, radius, compact);
  SIPX_HIP(hipGetLastError());
}

#define SIPX_INST(T)                                                                                              \
  template void K<T>::ps_reduce(hipStream_t, long long, const T*, double*, T*);                                  \
  template void K<T>::ps_finish(hipStream_t, const double*, const T*, ProjScalars<T>*, int, T, T, long long);    \
  template void K<T>::l1_theta(hipStream_t, long long, const T*, ProjScalars<T>*, T, unsigned long long*, double*, T*);
SIPX_INST(float)
SIPX_INST(double)

}  // namespace sipx
