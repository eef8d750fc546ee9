// Scalars of the non-elementwise projectors, computed entirely on the device.
//
// Replaces (reference file:line):
//   project_l1_Duchi!  src/projectors/project_l1_Duchi!.jl:21-52  -- the reference sorts all M
//       magnitudes (RadixSort/QuickSort), takes a cumsum and scans serially for the threshold.
//       Here theta solves  f(theta) = sum(max(|v|-theta,0)) - b = 0  (f convex, piecewise linear):
//         1. the pass that materialises v also evaluates (S,C)(t) = (sum, count of |v| > t) at
//            L1_K probe thresholds centred on the previous PARSDMM iteration's theta (registers only),
//         2. a scalar kernel brackets the root between two probes and tightens the bracket with a
//            Newton step from the left (Michelot) and the secant from the right,
//         3. one sparse compaction pass gathers the few magnitudes inside the bracket and the
//            exact (S,C) above it,
//         4. a single workgroup runs Michelot's fixed-point iteration on them: exact theta in float64.
//       A gated extra probe pass handles cold starts.
//   project_l2!        src/projectors/project_l2!.jl:3-16
//   project_annulus!   src/projectors/project_annulus!.jl:3-21
#include <stdexcept>
#include <string>

#include "sipx_device.h"

namespace sipx {

constexpr long long L1_CAP = 1 << 17;    // bracket population above which one more probe pass is run
constexpr int SL_ABOVE_S = PREP_SLOTS, SL_ABOVE_C = PREP_SLOTS + 1;   // partial slots of the compaction pass

// ||v||_1, ||v||_2^2, nnz, probe sums (slots 0..PREP_SLOTS-1) and per-block max|v| of a stored vector.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_ps_reduce(long long len, const T* __restrict__ v,
                                                     const ProjScalars<T>* __restrict__ ps, int gated,
                                                     double* __restrict__ partials, T* __restrict__ maxpart) {
  if (gated && !(ps->need && ps->refine)) return;
  double acc[PREP_SLOTS];
#pragma unroll
  for (int k = 0; k < PREP_SLOTS; ++k) acc[k] = 0;
  double t[L1_K];
#pragma unroll
  for (int k = 0; k < L1_K; ++k) t[k] = ps ? ps->t[k] : INFINITY;
  T vmax = T(0);
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < len; e += (long long)NB * BLOCK) {
    const T x = v[e], av = fabs(x);
    probe_acc<T>(av, x, t, acc);
    vmax = av > vmax ? av : vmax;
  }
  block_reduce_store<PREP_SLOTS>(acc, partials, 0);
  if (!gated) block_max_store<T>(vmax, maxpart);
}
template <typename T>
void K<T>::ps_reduce(hipStream_t s, long long len, const T* v, const ProjScalars<T>* ps, double* partials, T* maxpart) {
  hipLaunchKernelGGL((k_ps_reduce<T>), dim3(NB), dim3(BLOCK), 0, s, len, v, ps, 0, partials, maxpart);
  SIPX_HIP(hipGetLastError());
}

template <typename T>
__device__ __forceinline__ T block_max_partials(const T* __restrict__ maxpart) {
  __shared__ T smax[BLOCK / 64];
  T vmax = T(0);
  for (int i = threadIdx.x; i < NB; i += BLOCK) vmax = maxpart[i] > vmax ? maxpart[i] : vmax;
  vmax = wave_max<T>(vmax);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = vmax;
  __syncthreads();
  for (int i = 0; i < BLOCK / 64; ++i) vmax = smax[i] > vmax ? smax[i] : vmax;
  return vmax;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_ps_finish(const double* __restrict__ partials, const T* __restrict__ maxpart,
                                                     ProjScalars<T>* ps, int prox, T pmin, T pmax, long long true_len) {
  const double asum = block_sum_partials(partials);
  const double sumsq = block_sum_partials(partials + NB);
  const T vmax = block_max_partials<T>(maxpart);
  if (threadIdx.x == 0) {
    ps->asum = asum;
    ps->sumsq = sumsq;
    ps->vmax = vmax;
    ps->need = 0;
    ps->theta = T(0);
    ps->scale = T(1);
    ps->fill = 0;
    ps->tau = T(0);
    ps->quota = 0x7fffffffffffffffll;
    if (prox == PX_L2) {
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

template <typename T>
__global__ void k_ps_init(ProjScalars<T>* ps) {
  ps->asum = ps->sumsq = 0;
  ps->vmax = T(0);
  ps->need = 0;
  ps->theta = T(0);
  ps->scale = T(1);
  ps->fill = 0;
  for (int k = 0; k < L1_K; ++k) ps->t[k] = INFINITY;
  ps->lo = ps->hi = 0;
  ps->refine = 0;
  ps->theta_prev = 0;
  ps->n_compact = 0;
  ps->tau = T(0);
  ps->quota = 0x7fffffffffffffffll;
}
template <typename T>
void K<T>::ps_init(hipStream_t s, ProjScalars<T>* ps) {
  hipLaunchKernelGGL((k_ps_init<T>), dim3(1), dim3(1), 0, s, ps);
  SIPX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Bracket of the l1 threshold from the probe sums.  STAGE 0: first decision after the fused
// probe (thresholds around theta_prev); STAGE 1: after the gated refinement pass.
template <typename T, int STAGE>
__global__ __launch_bounds__(BLOCK) void k_l1_decide(const double* __restrict__ partials, const T* __restrict__ maxpart,
                                                     ProjScalars<T>* ps, T radius) {
  if (STAGE == 1 && !(ps->need && ps->refine)) return;
  __shared__ double red[PREP_SLOTS];
  for (int k = 0; k < PREP_SLOTS; ++k) {
    const double s = block_sum_partials(partials + (long long)k * NB);
    if (threadIdx.x == 0) red[k] = s;
  }
  T vmax = T(0);
  if (STAGE == 0) vmax = block_max_partials<T>(maxpart);
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double b = (double)radius;
  if (STAGE == 0) {
    ps->asum = red[0];
    ps->sumsq = red[1];
    ps->vmax = vmax;
    ps->scale = T(1);
    ps->fill = 0;
    ps->n_compact = 0;
    ps->refine = 0;
    ps->need = ((T)red[0] <= radius) ? 0 : 1;          // norm(v,1) <= b && return v   project_l1_Duchi!.jl:23
    if (!ps->need) {
      ps->theta = T(0);
      return;
    }
  } else {
    vmax = ps->vmax;
  }
  // candidates: virtual t=0 (S=||v||_1, C=nnz) then the probes, ascending
  double tl = 0, Sl = ps->asum, Cl = red[2], fl = ps->asum - b;
  double th = (double)vmax, Sh = 0, Ch = 0, fh = -b;
  (void)Sh;
  if (STAGE == 1) {                                    // known outer bracket from stage 0
    tl = ps->lo;
    th = ps->hi;
    fl = INFINITY;                                     // replaced by the first probe (== lo) below
  }
  bool have_l = STAGE == 0, have_h = STAGE == 0;     // stage 0: f(vmax) = -b is exact
  for (int k = 0; k < L1_K; ++k) {
    const double t = ps->t[k];
    if (!(t < INFINITY)) continue;
    const double S = red[3 + k], C = red[3 + L1_K + k];
    const double f = S - t * C - b;
    if (f >= 0) {
      if (!have_l || t >= tl) {
        tl = t; Sl = S; Cl = C; fl = f;
        have_l = true;
      }
    } else if (t < th || (!have_h && t <= th)) {
      th = t; Sh = S; Ch = C; fh = f;
      have_h = true;
    }
  }
  if (!have_l) {                                       // rounding pushed f(lo) below zero: fall back to t = 0
    tl = 0; Sl = ps->asum; Cl = red[2]; fl = ps->asum - b;
    ps->t[0] = INFINITY;
  }
  // Newton from the left (Michelot step) and secant from the right: theta* in [thN, thS]
  double thN = Cl > 0 ? (Sl - b) / Cl : tl;
  if (!(thN >= tl)) thN = tl;
  double thS = th;
  if (have_h && fl < INFINITY && fl - fh > 0) thS = tl + fl * (th - tl) / (fl - fh);
  if (!(thS <= th)) thS = th;
  if (!(thS >= thN)) thS = th;
  const double lo = thN * (1.0 - 1e-9);
  const double hi = thS * (1.0 + 1e-9) + 1e-300;
  ps->lo = lo > tl ? lo : tl;
  ps->hi = hi < th ? hi : th;
  const double n_est = Cl - Ch;
  if (STAGE == 0 && n_est > (double)L1_CAP) {          // cold start: subdivide [lo, hi] once more
    ps->refine = 1;
    for (int k = 0; k < L1_K; ++k) ps->t[k] = ps->lo + (ps->hi - ps->lo) * (double)k / (double)(L1_K - 1);
  } else {
    ps->refine = 0;
  }
}

// Gathers the magnitudes in (lo, hi] and reduces (S,C) of those above hi.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_l1_compact(long long len, const T* __restrict__ v, ProjScalars<T>* ps,
                                                      T* __restrict__ compact, double* __restrict__ partials) {
  if (!ps->need) return;
  const double lo = ps->lo, hi = ps->hi;
  double acc[2] = {0, 0};
  for (long long e0 = (long long)blockIdx.x * BLOCK; e0 < len; e0 += (long long)NB * BLOCK) {
    const long long e = e0 + threadIdx.x;
    const T av = e < len ? fabs(v[e]) : T(0);
    const double a = (double)av;
    if (a > hi) {
      acc[0] += a;
      acc[1] += 1.0;
    }
    const bool in = a > lo && a <= hi;
    const unsigned long long mask = __ballot(in);
    if (mask) {
      const int lane = threadIdx.x & 63;
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(&ps->n_compact, (unsigned long long)__popcll(mask));
      base = __shfl(base, 0, 64);
      if (in) compact[base + __popcll(mask & ((1ull << lane) - 1ull))] = av;
    }
  }
  block_reduce_store<2>(acc, partials, SL_ABOVE_S);
}

// Michelot's iteration on the compacted magnitudes: theta <- (S_above + S_in(>theta) - b) / (C_above + C_in(>theta)),
// monotone from the bracket's lower end, exact after finitely many steps (stops when the active count repeats).
template <typename T>
__global__ __launch_bounds__(1024) void k_l1_solve(ProjScalars<T>* ps, T radius, const T* __restrict__ compact,
                                                   const double* __restrict__ partials) {
  if (!ps->need) return;
  __shared__ double ssum[16];
  __shared__ double scnt[16];
  __shared__ double sh_theta, sh_sa, sh_ca;
  __shared__ int sh_done;
  {   // (S,C) above the bracket: sum the compaction pass's block partials
    double s = 0, c = 0;
    for (int i = threadIdx.x; i < NB; i += 1024) {
      s += partials[(long long)SL_ABOVE_S * NB + i];
      c += partials[(long long)SL_ABOVE_C * NB + i];
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
      ssum[threadIdx.x >> 6] = s;
      scnt[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double S = 0, Cc = 0;
      for (int i = 0; i < 16; ++i) {
        S += ssum[i];
        Cc += scnt[i];
      }
      sh_sa = S;
      sh_ca = Cc;
    }
    __syncthreads();
  }
  const long long n = (long long)ps->n_compact;
  const double sa = sh_sa, ca = sh_ca, b = (double)radius;
  double theta = ps->lo;
  double cprev = -1;
  for (int it = 0; it < 200; ++it) {
    double s = 0, c = 0;
    for (long long e = threadIdx.x; e < n; e += 1024) {
      const double av = (double)compact[e];
      if (av > theta) {
        s += av;
        c += 1.0;
      }
    }
    s = wave_sum(s);
    c = wave_sum(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      ssum[threadIdx.x >> 6] = s;
      scnt[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double S = 0, Cc = 0;
      for (int i = 0; i < 16; ++i) {
        S += ssum[i];
        Cc += scnt[i];
      }
      const double tot = ca + Cc;
      double tn = theta;
      if (tot > 0) tn = (sa + S - b) / tot;
      sh_done = (Cc == cprev || !(tot > 0)) ? 1 : 0;
      sh_theta = tn > theta ? tn : theta;
      scnt[0] = Cc;
    }
    __syncthreads();
    theta = sh_theta;
    cprev = scnt[0];
    if (sh_done) break;
  }
  if (threadIdx.x == 0) {
    const T th = (T)theta;
    ps->theta = th > T(0) ? th : T(0);       // theta = max(0, .)   project_l1_Duchi!.jl:46
    if (theta > 0) {
      ps->theta_prev = theta;                // warm start of the next probe: thresholds around theta
      const double mult[L1_K] = {0.5, 0.9, 0.99, 0.999, 1.001, 1.01, 1.1, 2.0};
      for (int k = 0; k < L1_K; ++k) ps->t[k] = theta * mult[k];
    }
  }
}

template <typename T>
void K<T>::l1_theta(hipStream_t s, long long len, const T* v, ProjScalars<T>* ps, T radius, double* partials,
                    const T* maxpart, T* compact) {
  hipLaunchKernelGGL((k_l1_decide<T, 0>), dim3(1), dim3(BLOCK), 0, s, partials, maxpart, ps, radius);
  hipLaunchKernelGGL((k_ps_reduce<T>), dim3(NB), dim3(BLOCK), 0, s, len, v, ps, 1, partials, (T*)nullptr);
  hipLaunchKernelGGL((k_l1_decide<T, 1>), dim3(1), dim3(BLOCK), 0, s, partials, maxpart, ps, radius);
  hipLaunchKernelGGL((k_l1_compact<T>), dim3(NB), dim3(BLOCK), 0, s, len, v, ps, compact, partials);
  hipLaunchKernelGGL((k_l1_solve<T>), dim3(1), dim3(1024), 0, s, ps, radius, compact, partials);
  SIPX_HIP(hipGetLastError());
}

#define SIPX_INST(T)                                                                                              \
  template void K<T>::ps_reduce(hipStream_t, long long, const T*, const ProjScalars<T>*, double*, T*);           \
  template void K<T>::ps_finish(hipStream_t, const double*, const T*, ProjScalars<T>*, int, T, T, long long);    \
  template void K<T>::ps_init(hipStream_t, ProjScalars<T>*);                                                     \
  template void K<T>::l1_theta(hipStream_t, long long, const T*, ProjScalars<T>*, T, double*, const T*, T*);
SIPX_INST(float)
SIPX_INST(double)

}  // namespace sipx
