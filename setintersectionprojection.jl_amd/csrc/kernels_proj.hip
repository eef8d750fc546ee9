// Scalars of the non-elementwise projectors, computed entirely on the device.
//
// Replaces (reference file:line):
//   project_l1_Duchi!  src/projectors/project_l1_Duchi!.jl:21-52  -- the reference sorts all M
//       magnitudes (RadixSort/QuickSort), takes a cumsum and scans serially for the threshold.
//       Here theta solves  f(theta) = sum(max(|v|-theta,0)) - b = 0  (f convex, piecewise linear):
//         1. ONE pass produces v on the fly (stencil of x, y, l -- nothing is stored), evaluates
//            (S,C)(t) = (sum, count of |v| > t) at L1_K probe thresholds centred on the previous
//            PARSDMM iteration's theta (registers only) and speculatively gathers the magnitudes in a
//            narrow range around that theta through a per-workgroup LDS buffer (one global atomic
//            per workgroup),
//         2. a scalar kernel brackets the root between two probes and tightens the bracket with a
//            Newton step from the left (Michelot) and the secant from the right; if the bracket lies
//            inside the speculative range the gathered values are all that is needed,
//         3. otherwise (cold start / theta moved a lot) gated fallback passes refine the bracket and
//            gather it explicitly,
//         4. a single workgroup runs Michelot's fixed-point iteration on the gathered values:
//            exact theta in float64.
//   project_l2!        src/projectors/project_l2!.jl:3-16
//   project_annulus!   src/projectors/project_annulus!.jl:3-21
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "sipx_device.h"

namespace sipx {

// MEMORY-MODEL ASSUMPTION of the last-workgroup hand-offs in this file (k_slot_sums<FUSE>, k_sample, the coop_* publishing of
// k_l1_solve): a workgroup publishes its values with RELAXED agent-scope atomic stores, waits for them with s_waitcnt(0), then
// takes a ticket with a RELAXED agent-scope fetch-add; the workgroup that draws the last ticket reads the values back with
// relaxed agent-scope atomic loads.  The HIP / LLVM memory model does not order relaxed operations on different addresses; the
// hardware this file is written for does: on gfx950 (CDNA4) an agent-scope atomic store is performed at the coherence point
// (sc1: write-through, it never sits dirty in the XCD's L2) and is counted in vmcnt, so s_waitcnt(0) before the ticket means
// "my stores have been performed"; an agent-scope atomic load bypasses the non-coherent levels.  A release / acquire pair
// instead would write back / invalidate the XCD's whole L2 -- which holds whatever the other set stream left dirty -- and cost
// 15 of k_sample's 40 us (measured).  A target with a separate store counter or a different cache protocol must not
// compile this as is:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "kernels_proj.hip: the relaxed-atomics hand-offs assume gfx950 (see the comment above); port them to release/acquire first"
#endif

constexpr long long SOLVE_COOP_MIN_DEFAULT = 1ll << 17;
constexpr double L1_CAP = 131072.0;     // bracket population above which one more probe pass is run (floor; scales with the length)
constexpr int L1_REFINES = 1;           // gated refinement passes enqueued per search (one rank: the final gather never drops)
constexpr int L1_REFINES_SLAB = 6;      // slab-decomposed: rounds the caller may enqueue until the bracket fits the exchange segments
#ifndef SIPX_SPEC_CAP
#define SIPX_SPEC_CAP 1024
#endif
constexpr int SPEC_CAP = SIPX_SPEC_CAP;  // per-workgroup LDS buffer of the speculative compaction
// largest relative half-width of the speculative range (SIPX_L1_HWMAX overrides; A/B switch)
static int l1_lean_on() {       // SIPX_L1_LEAN=0: every first pass evaluates all eight probes (A/B switch)
  static const int v = [] { const char* e = getenv("SIPX_L1_LEAN"); return e ? atoi(e) : 1; }();
  return v;
}
static long long solve_coop_min() {      // SIPX_SOLVE_COOP_MIN: gathered values from which the sweeps of the solve are shared (A/B switch)
  static const long long v = [] { const char* e = getenv("SIPX_SOLVE_COOP_MIN"); return e ? atoll(e) : SOLVE_COOP_MIN_DEFAULT; }();
  return v;
}
static double l1_hw_max() {
  static const double v = [] { const char* e = getenv("SIPX_L1_HWMAX"); return e ? atof(e) : 1e-2; }();
  return v;
}
// Probe thresholds of the next call, as multiples of the half-width hw around the predicted theta: the two inner probes
// are the edges of the speculative gather range, the outer ones catch a theta that moved further (geometric spacing, so one
// pass brackets it between neighbouring probes whatever the size of the move up to 64 hw).
constexpr int L1_WIN_LO = 3, L1_WIN_HI = 4;
__device__ __forceinline__ double l1_probe_mult(int k) {
  const double m[L1_K] = {-64.0, -16.0, -4.0, -1.0, 1.0, 4.0, 16.0, 64.0};
  return m[k];
}
// Workgroups of every k_pass launch (the only producers of the search's partial slots): five per compute unit, the
// occupancy of the first pass, so the grid is resident in one round (+1 % at 256^3 over seven per CU).
#define SIPX_PASS_GRID launch_blocks(5)
constexpr int SL_ABOVE_S = PREP_SLOTS, SL_ABOVE_C = PREP_SLOTS + 1;   // partial slots of the fallback compaction
enum { M_FIRST = 0, M_PROBE = 1, M_COMPACT = 2, M_DIST = 3, M_STORE = 4 /* materialise v into `compact` */,
       M_LEAN = 5 /* first pass with the two edge probes of the speculative range only (a kernel of its own: 52 instead of 113 VGPRs) */ };

// One pass over the vector.  SRC 0: stored array (V == 1); SRC 1: produced on the fly by a set.
template <typename T, int V, int MODE, int SRC>
__global__ __launch_bounds__(BLOCK) void k_pass(Grid G, SetArgs<T> a, int v_is_s, const T* __restrict__ varr,
                                                long long len, ProjScalars<T>* ps, T* __restrict__ compact,
                                                double* __restrict__ partials, T* __restrict__ maxpart) {
  if (MODE == M_PROBE && !(ps->need && !ps->spec_ok && ps->refine)) return;
  long long* cidx = nullptr;
  if (MODE == M_COMPACT && a.prox == PX_CARD) cidx = ps->cidx;
  if (MODE == M_COMPACT && !(ps->need && !ps->spec_ok)) return;
  constexpr bool GATHERS = MODE == M_FIRST || MODE == M_COMPACT || MODE == M_LEAN;
  __shared__ T sbuf[GATHERS ? SPEC_CAP : 1];
  __shared__ unsigned int scnt, sused;     // reserved / actually filled prefix of sbuf
  __shared__ int sovf;
  __shared__ unsigned long long sbase;
  double acc[PREP_SLOTS];
#pragma unroll
  for (int k = 0; k < PREP_SLOTS; ++k) acc[k] = 0;
  ProbeAcc<T> pa;
  if ((a.prox == PX_L1 || a.prox == PX_CARD) && (MODE <= M_PROBE || MODE == M_LEAN)) {
#pragma unroll
    for (int k = 0; k < L1_K; ++k) pa.t[k] = (T)ps->t[k];      // stored TF-rounded: exact
  }
  double r_lo = 0, r_hi = -1;      // gather range (lo, hi]
  if ((MODE == M_FIRST || MODE == M_LEAN) && a.prox == PX_L1 && !(a.flags & F_NOSPEC)) { r_lo = ps->spec_lo; r_hi = ps->spec_hi; }
  if (MODE == M_COMPACT) { r_lo = ps->lo; r_hi = ps->hi; }
  const bool gather = r_hi > r_lo;
  // The first pass of an l1 search is one of two kernels, launched back to back: the LEAN one when the previous search (or the
  // sampled estimate) announced that the speculative range will hold theta, the full one otherwise; the other returns here.
  const bool lean = (MODE == M_FIRST || MODE == M_LEAN) && a.prox == PX_L1 && gather && ps->lean != 0;
  if (MODE == M_FIRST && lean) return;
  if (MODE == M_LEAN && !lean) return;
  T vmax = T(0), vminp = (T)INFINITY;     // largest magnitude, smallest non-zero magnitude
  if (GATHERS) {
    if (threadIdx.x == 0) { scnt = 0; sused = 0; sovf = 0; }
    __syncthreads();
  }
  ProxCtx<T> pc;
  if (MODE == M_DIST) pc = make_prox<T>(a.prox, a.plo, a.phi, T(0), ps);
  const int lane = threadIdx.x & 63;

  auto body = [&](T x, long long e, bool live) {
    const T av = fabs(x);
    const double ad = (double)av;
    if (MODE == M_LEAN) {
      pa.add_lean(av, L1_WIN_LO, L1_WIN_HI);
      vminp = (av > T(0) && av < vminp) ? av : vminp;
    } else if (MODE == M_FIRST || MODE == M_PROBE) {
      pa.add(av, x);
      vmax = av > vmax ? av : vmax;
      if (MODE == M_FIRST) vminp = (av > T(0) && av < vminp) ? av : vminp;
    }
    if (MODE == M_COMPACT && ad > r_hi) {
      acc[0] += ad;
      acc[1] += 1.0;
    }
    if (MODE == M_STORE && live) compact[e] = x;
    if (MODE == M_DIST && live) {          // (lanes beyond the range carry x = 0, and P(0) - 0 is not zero for every set: bounds away from zero)
      const T pv = prox_apply<T>(pc, x, T(0), T(0), T(0), e);
      const T dlt = pv - x;
      acc[0] += (double)dlt * (double)dlt;
      acc[1] += (double)x * (double)x;
    }
    if (GATHERS && gather) {
      const bool in = ad > r_lo && ad <= r_hi;
      const unsigned long long mask = __ballot(in);
      if (mask) {
        const int leader = __ffsll((long long)mask) - 1;
        const int cnt = __popcll(mask);
        const int my = __popcll(mask & ((1ull << lane) - 1ull));
        unsigned int base = SPEC_CAP;
        if (!cidx) {
          if (lane == leader) base = atomicAdd(&scnt, (unsigned int)cnt);
          base = __shfl(base, leader, 64);
        }
        if (cidx) {                                         // cardinality: (magnitude, index) pairs, few of them
          unsigned long long gb = 0;
          if (lane == leader) gb = atomicAdd(&ps->n_compact, (unsigned long long)cnt);
          gb = __shfl(gb, leader, 64);
          if (in) { compact[gb + my] = av; cidx[gb + my] = e; }
        } else if (base + cnt <= SPEC_CAP) {                // room in the workgroup's LDS buffer
          if (in) sbuf[base + my] = av;
          if (lane == leader) atomicMax(&sused, base + (unsigned int)cnt);
        } else if (MODE == M_FIRST || MODE == M_LEAN) {
          sovf = 1;                                         // speculation gathered too much: give it up
        } else {                                            // fallback compaction never drops: go to global memory
          unsigned long long gb = 0;
          if (lane == leader) gb = atomicAdd(&ps->n_compact, (unsigned long long)cnt);
          gb = __shfl(gb, leader, 64);
          if (in) compact[gb + my] = av;
        }
      }
    }
  };

  // Fallback compaction: the workgroup's LDS buffer is emptied into the global one whenever it is more than half full (one global
  // atomic per 512 values).  Round 3 let a full buffer send every further wave to the global counter by itself: a bracket that holds
  // a percent of a 512^3 vector (the first search of a set, the magnitudes behind the DFT) made that two million atomics on one
  // address -- 2.8 ms for a pass that streams in 0.2 (C4: once per iteration for the l1-DFT set; 4.8 ms per set at sipx_finalize).
  auto drain = [&]() {
    if (MODE != M_COMPACT || !gather || cidx) return;      // (uniform over the workgroup)
    __syncthreads();
    if (scnt <= SPEC_CAP / 2) return;
    const unsigned int cnt = sused;
    if (threadIdx.x == 0) sbase = cnt ? atomicAdd(&ps->n_compact, (unsigned long long)cnt) : 0ull;
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < cnt; i += BLOCK) compact[sbase + i] = sbuf[i];
    __syncthreads();
    if (threadIdx.x == 0) { scnt = 0; sused = 0; }
    __syncthreads();
  };
  if (SRC == 0) {
    int round = 0;
    for (long long e0 = (long long)blockIdx.x * BLOCK; e0 < len; e0 += (long long)gridDim.x * BLOCK) {
      const long long e = e0 + threadIdx.x;
      body(e < len ? varr[e] : T(0), e, e < len);   // uniform trip count: every lane takes part in the ballots
      if ((++round & 3) == 0) drain();              // (a round is 256 values: looked at every 1024, as on the stencil path)
    }
  } else {
    const bool ident = a.nblk == 0;
    const int nb = ident ? 1 : a.nblk;
    const bool relax = !(a.gamma == T(1));
    const T gam = a.gamma, omg = T(1) - a.gamma;
    long long v0, nvec;                      // this launch's share of the grid (all of it unless slab-decomposed)
    vec_range<V>(G, v0, nvec);
    const long long nit = (nvec - v0 + (long long)gridDim.x * BLOCK - 1) / ((long long)gridDim.x * BLOCK);
    for (long long it = 0; it < nit; ++it) {
      const long long vi = v0 + it * (long long)gridDim.x * BLOCK + (long long)blockIdx.x * BLOCK + threadIdx.x;
      const bool live = vi < nvec;
      const long long g = live ? vi * V : v0 * V;      // (lanes beyond the range shadow its first point: every array is backed there)
      const Coord c = coords(G, g);
      const Vec<T, V> xc = ldv<T, V>(a.x + g);
      for (int q = 0; q < nb; ++q) {
        const long long e = (long long)q * G.N + g;
        T s[V];
        bool valid[V];
        if (ident) {
#pragma unroll
          for (int k = 0; k < V; ++k) { s[k] = xc.v[k]; valid[k] = true; }
        } else {
          fwd_dir<T, V>(G, a.x, xc, g, c, a.dir[q], a.ih[q], s, valid);
        }
        T out[V];
        if (v_is_s) {
#pragma unroll
          for (int k = 0; k < V; ++k) out[k] = s[k];
        } else {
          const Vec<T, V> yv = ldv<T, V>(a.y + e), lv = ldv<T, V>(a.l + e);
#pragma unroll
          for (int k = 0; k < V; ++k) {
            const T xh = relax ? (gam * s[k] + omg * yv.v[k]) : s[k];       // update_y_l.jl:72
            out[k] = valid[k] ? (xh - lv.v[k] * a.rho1) : T(0);            // :67 / :74
          }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) body(live ? out[k] : T(0), e + k, live);
      }
      drain();
    }
  }

  if (MODE == M_FIRST || MODE == M_PROBE) {
    pa.to_slots(acc);
    block_reduce_store<PREP_SLOTS>(acc, partials, 0);
    if (MODE == M_FIRST) {
      block_max_store<T>(vmax, maxpart);
      block_min_store<T>(vminp, maxpart + NB);      // second half of the array
    }
  } else if (MODE == M_LEAN) {
    // only ||v||_1 and (S, C) at the two edges of the speculative range: the decision of a lean pass reads nothing else
    double a5[5] = {pa.asum, pa.S[L1_WIN_LO], pa.S[L1_WIN_HI], (double)pa.C[L1_WIN_LO], (double)pa.C[L1_WIN_HI]};
    const int slots[5] = {0, 3 + L1_WIN_LO, 3 + L1_WIN_HI, 3 + L1_K + L1_WIN_LO, 3 + L1_K + L1_WIN_HI};
    block_reduce_store_at<5>(a5, partials, slots);
    block_min_store<T>(vminp, maxpart + NB);
  } else if (MODE == M_COMPACT) {
    double a2[2] = {acc[0], acc[1]};
    block_reduce_store<2>(a2, partials, SL_ABOVE_S);
  } else if (MODE == M_STORE) {
  } else {
    double a2[2] = {acc[0], acc[1]};
    block_reduce_store<2>(a2, partials, 0);
  }
  if (GATHERS && gather) {   // flush the workgroup's buffer: one global atomic
    __syncthreads();
    // reservations grow monotonically, so the stored entries form the prefix [0, sused)
    const unsigned int cnt = sused;
    if (threadIdx.x == 0) {
      sbase = cnt ? atomicAdd(&ps->n_compact, (unsigned long long)cnt) : 0ull;
      if ((MODE == M_FIRST || MODE == M_LEAN) && sovf) atomicOr(&ps->spec_overflow, 1);
    }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < cnt; i += BLOCK) compact[sbase + i] = sbuf[i];
  }
}

// ---------------------------------------------------------------------------------------------
// The LEAN first passes of up to three l1 searches in ONE sweep (round 3): the sets share x, so it is read once -- 1 + 2 per
// block of every set instead of 3 per block -- and one launch replaces three.  For every set whose device-side state asks for
// a lean pass (ProjScalars::lean; a set that does not is left to its own full first pass, launched as before) the kernel does
// exactly what k_pass<M_LEAN> does: the same thread-to-element mapping, the same arithmetic, the same partial slots in the
// set's own buffers, the same values gathered into the set's own buffer (in another order, which the solve does not depend on).
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_lean_multi(Grid G, LeanMulti<T> m) {
  __shared__ T sbuf[LEAN_MAX][SPEC_CAP];
  __shared__ unsigned int scnt[LEAN_MAX], sused[LEAN_MAX];
  __shared__ int sovf[LEAN_MAX];
  __shared__ unsigned long long sbase[LEAN_MAX];
  bool on[LEAN_MAX];
  T tlo[LEAN_MAX], thi[LEAN_MAX], vminp[LEAN_MAX];
  double r_lo[LEAN_MAX], r_hi[LEAN_MAX], asum[LEAN_MAX], Slo[LEAN_MAX], Shi[LEAN_MAX];
  unsigned int Clo[LEAN_MAX], Chi[LEAN_MAX];
  bool any = false;
#pragma unroll
  for (int q = 0; q < LEAN_MAX; ++q) {
    on[q] = false;
    tlo[q] = thi[q] = (T)INFINITY; vminp[q] = (T)INFINITY;
    r_lo[q] = 0; r_hi[q] = -1; asum[q] = Slo[q] = Shi[q] = 0; Clo[q] = Chi[q] = 0;
    if (q < m.ns) {
      const ProjScalars<T>* ps = m.s[q].ps;
      r_lo[q] = ps->spec_lo; r_hi[q] = ps->spec_hi;
      on[q] = ps->lean != 0 && r_hi[q] > r_lo[q];         // (the caller offers l1 sets without F_NOSPEC only)
      tlo[q] = (T)ps->t[L1_WIN_LO]; thi[q] = (T)ps->t[L1_WIN_HI];
      any |= on[q];
    }
  }
  if (!any) return;
  if (threadIdx.x < LEAN_MAX) { scnt[threadIdx.x] = 0; sused[threadIdx.x] = 0; sovf[threadIdx.x] = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  long long v0, nvec;
  vec_range<V>(G, v0, nvec);
  const long long nit = (nvec - v0 + (long long)gridDim.x * BLOCK - 1) / ((long long)gridDim.x * BLOCK);
  const T* x = m.s[0].a.x;
  for (long long it = 0; it < nit; ++it) {
    const long long vi = v0 + it * (long long)gridDim.x * BLOCK + (long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = vi < nvec;
    const long long g = live ? vi * V : v0 * V;      // (lanes beyond the range shadow its first point: every array is backed there)
    const Coord c = coords(G, g);
    const Vec<T, V> xc = ldv<T, V>(x + g);
#pragma unroll
    for (int q = 0; q < LEAN_MAX; ++q) {
      if (q >= m.ns || !on[q]) continue;
      const SetArgs<T>& a = m.s[q].a;
      const bool ident = a.nblk == 0;
      const int nb = ident ? 1 : a.nblk;
      const bool relax = !(a.gamma == T(1));
      const T gam = a.gamma, omg = T(1) - a.gamma;
      for (int b = 0; b < nb; ++b) {
        const long long e = (long long)b * G.N + g;
        T s[V];
        bool valid[V];
        if (ident) {
#pragma unroll
          for (int k = 0; k < V; ++k) { s[k] = xc.v[k]; valid[k] = true; }
        } else {
          fwd_dir<T, V>(G, a.x, xc, g, c, a.dir[b], a.ih[b], s, valid);
        }
        Vec<T, V> yv = zerov<T, V>(), lv = zerov<T, V>();
        if (!m.v_is_s) { yv = ldv<T, V>(a.y + e); lv = ldv<T, V>(a.l + e); }
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const T xh = relax ? (gam * s[k] + omg * yv.v[k]) : s[k];       // update_y_l.jl:72
          const T vv = m.v_is_s ? (live ? s[k] : T(0)) : (live ? (valid[k] ? (xh - lv.v[k] * a.rho1) : T(0)) : T(0));      // :67 / :74
          const T av = fabs(vv);
          const double ad = (double)av;
          asum[q] += ad;
          const bool o_lo = av > tlo[q], o_hi = av > thi[q];
          Slo[q] += o_lo ? ad : 0.0; Clo[q] += o_lo ? 1u : 0u;
          Shi[q] += o_hi ? ad : 0.0; Chi[q] += o_hi ? 1u : 0u;
          vminp[q] = (av > T(0) && av < vminp[q]) ? av : vminp[q];
          const bool in = ad > r_lo[q] && ad <= r_hi[q];
          const unsigned long long mask = __ballot(in);
          if (mask) {
            const int leader = __ffsll((long long)mask) - 1;
            const int cnt = __popcll(mask);
            const int my = __popcll(mask & ((1ull << lane) - 1ull));
            unsigned int base = SPEC_CAP;
            if (lane == leader) base = atomicAdd(&scnt[q], (unsigned int)cnt);
            base = __shfl(base, leader, 64);
            if (base + cnt <= SPEC_CAP) {                   // room in the workgroup's LDS buffer of this set
              if (in) sbuf[q][base + my] = av;
              if (lane == leader) atomicMax(&sused[q], base + (unsigned int)cnt);
            } else {
              sovf[q] = 1;                                  // speculation gathered too much: give it up
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < LEAN_MAX; ++q) {
    if (q >= m.ns || !on[q]) continue;
    __syncthreads();                                        // (the reduction helpers share their LDS scratch between calls)
    double a5[5] = {asum[q], Slo[q], Shi[q], (double)Clo[q], (double)Chi[q]};
    const int slots[5] = {0, 3 + L1_WIN_LO, 3 + L1_WIN_HI, 3 + L1_K + L1_WIN_LO, 3 + L1_K + L1_WIN_HI};
    block_reduce_store_at<5>(a5, m.s[q].partials, slots);
    block_min_store<T>(vminp[q], m.s[q].maxpart + NB);
    __syncthreads();
    const unsigned int cnt = sused[q];                      // reservations grow monotonically: the stored entries form the prefix [0, sused)
    if (threadIdx.x == 0) {
      sbase[q] = cnt ? atomicAdd(&m.s[q].ps->n_compact, (unsigned long long)cnt) : 0ull;
      if (sovf[q]) atomicOr(&m.s[q].ps->spec_overflow, 1);
    }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < cnt; i += BLOCK) m.s[q].compact[sbase[q] + i] = sbuf[q][i];
  }
}

template <typename T>
void K<T>::lean_multi(hipStream_t s, const Grid& g, const LeanMulti<T>& m) {
  if (m.ns < 1 || m.ns > LEAN_MAX || g.n[0] % 4 != 0) throw std::runtime_error("lean_multi: unsupported call");
  // (one set: the same kernel -- the caller keeps ONE chain of launches for every set list)
  double bytes = (double)range_len(g);                 // x once, y and l of every block of every set
  for (int q = 0; q < m.ns; ++q) bytes += (m.v_is_s ? 0.0 : 2.0) * (double)m.s[q].a.nblk_or1() * (double)range_len(g);
  ObsScope obs_(KID_PASS_LEAN, s, bytes * sizeof(T));
  hipLaunchKernelGGL((k_lean_multi<T, 4>), dim3(fit_grid(range_len(g) / 4, SIPX_PASS_GRID)), dim3(BLOCK), 0, s, g, m);
  SIPX_HIP(hipGetLastError());
}


// ---------------------------------------------------------------------------------------------
// The FULL first passes, the gated refinement passes or the gated compaction passes of up to three searches in ONE sweep
// (round 4; the lean first passes have k_lean_multi): x is read once, one launch replaces three, and sets that have nothing to
// do in this mode (device-side state, exactly the tests k_pass makes) cost nothing.  For every set that takes part the kernel
// does what k_pass<MODE> does -- the same thread-to-element mapping on the same grid, the same arithmetic, the same partial
// slots in the set's own buffers, the same values gathered into the set's own buffer (in another order, which the solve does
// not depend on) -- so sums, decisions and theta are those of the per-set passes, bit for bit.  v_is_s: the searches of the
// feasibility estimates (the vector is s = A x itself).  Cardinality searches keep their own chain.
template <typename T, int V, int MODE>
__global__ __launch_bounds__(BLOCK) void k_pass_multi(Grid G, LeanMulti<T> m, int v_is_s) {
  static_assert(MODE == M_FIRST || MODE == M_PROBE || MODE == M_COMPACT, "modes of k_pass_multi");
  constexpr bool GATHERS = MODE == M_FIRST || MODE == M_COMPACT;
  __shared__ T sbuf[GATHERS ? LEAN_MAX : 1][GATHERS ? SPEC_CAP : 1];
  __shared__ unsigned int scnt[LEAN_MAX], sused[LEAN_MAX];
  __shared__ int sovf[LEAN_MAX];
  __shared__ unsigned long long sbase[LEAN_MAX];
  bool on[LEAN_MAX], gather[LEAN_MAX];
  double r_lo[LEAN_MAX], r_hi[LEAN_MAX], above_s[LEAN_MAX], above_c[LEAN_MAX];
  T vmax[LEAN_MAX], vminp[LEAN_MAX];
  ProbeAcc<T> pa[LEAN_MAX];
  bool any = false;
#pragma unroll
  for (int q = 0; q < LEAN_MAX; ++q) {
    on[q] = gather[q] = false;
    r_lo[q] = 0; r_hi[q] = -1; above_s[q] = above_c[q] = 0;
    vmax[q] = T(0); vminp[q] = (T)INFINITY;
    if (q < m.ns) {
      const ProjScalars<T>* ps = m.s[q].ps;
      const SetArgs<T>& a = m.s[q].a;
      const bool l1 = a.prox == PX_L1;
      if (MODE == M_FIRST) {
        if (l1 && !(a.flags & F_NOSPEC)) { r_lo[q] = ps->spec_lo; r_hi[q] = ps->spec_hi; }
        gather[q] = r_hi[q] > r_lo[q];
        on[q] = !(l1 && gather[q] && ps->lean != 0);       // (a lean search has had its pass: k_lean_multi / k_pass<M_LEAN>)
      } else if (MODE == M_PROBE) {
        on[q] = l1 && ps->need && !ps->spec_ok && ps->refine;
      } else {
        on[q] = l1 && ps->need && !ps->spec_ok;
        r_lo[q] = ps->lo; r_hi[q] = ps->hi;
        gather[q] = r_hi[q] > r_lo[q];
      }
      if (on[q] && l1 && MODE != M_COMPACT) {
#pragma unroll
        for (int k = 0; k < L1_K; ++k) pa[q].t[k] = (T)ps->t[k];      // stored TF-rounded: exact
      }
      any |= on[q];
    }
  }
  if (!any) return;
  if (GATHERS) {
    if (threadIdx.x < LEAN_MAX) { scnt[threadIdx.x] = 0; sused[threadIdx.x] = 0; sovf[threadIdx.x] = 0; }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  long long v0, nvec;
  vec_range<V>(G, v0, nvec);
  const long long nit = (nvec - v0 + (long long)gridDim.x * BLOCK - 1) / ((long long)gridDim.x * BLOCK);
  const T* x = m.s[0].a.x;
  for (long long it = 0; it < nit; ++it) {
    const long long vi = v0 + it * (long long)gridDim.x * BLOCK + (long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = vi < nvec;
    const long long g = live ? vi * V : v0 * V;      // (lanes beyond the range shadow its first point: every array is backed there)
    const Coord c = coords(G, g);
    const Vec<T, V> xc = ldv<T, V>(x + g);
#pragma unroll
    for (int q = 0; q < LEAN_MAX; ++q) {
      if (q >= m.ns || !on[q]) continue;
      const SetArgs<T>& a = m.s[q].a;
      const bool ident = a.nblk == 0;
      const int nb = ident ? 1 : a.nblk;
      const bool relax = !(a.gamma == T(1));
      const T gam = a.gamma, omg = T(1) - a.gamma;
      for (int b = 0; b < nb; ++b) {
        const long long e = (long long)b * G.N + g;
        T s[V];
        bool valid[V];
        if (ident) {
#pragma unroll
          for (int k = 0; k < V; ++k) { s[k] = xc.v[k]; valid[k] = true; }
        } else {
          fwd_dir<T, V>(G, a.x, xc, g, c, a.dir[b], a.ih[b], s, valid);
        }
        T out[V];
        if (v_is_s) {
#pragma unroll
          for (int k = 0; k < V; ++k) out[k] = s[k];
        } else {
          const Vec<T, V> yv = ldv<T, V>(a.y + e), lv = ldv<T, V>(a.l + e);
#pragma unroll
          for (int k = 0; k < V; ++k) {
            const T xh = relax ? (gam * s[k] + omg * yv.v[k]) : s[k];       // update_y_l.jl:72
            out[k] = valid[k] ? (xh - lv.v[k] * a.rho1) : T(0);            // :67 / :74
          }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const T vv = live ? out[k] : T(0);
          const T av = fabs(vv);
          const double ad = (double)av;
          if (MODE == M_FIRST || MODE == M_PROBE) {
            pa[q].add(av, vv);
            vmax[q] = av > vmax[q] ? av : vmax[q];
            if (MODE == M_FIRST) vminp[q] = (av > T(0) && av < vminp[q]) ? av : vminp[q];
          }
          if (MODE == M_COMPACT && ad > r_hi[q]) { above_s[q] += ad; above_c[q] += 1.0; }
          if (GATHERS && gather[q]) {
            const bool in = ad > r_lo[q] && ad <= r_hi[q];
            const unsigned long long mask = __ballot(in);
            if (mask) {
              const int leader = __ffsll((long long)mask) - 1;
              const int cnt = __popcll(mask);
              const int my = __popcll(mask & ((1ull << lane) - 1ull));
              unsigned int base = SPEC_CAP;
              if (lane == leader) base = atomicAdd(&scnt[q], (unsigned int)cnt);
              base = __shfl(base, leader, 64);
              if (base + cnt <= SPEC_CAP) {                   // room in the workgroup's LDS buffer of this set
                if (in) sbuf[GATHERS ? q : 0][base + my] = av;
                if (lane == leader) atomicMax(&sused[q], base + (unsigned int)cnt);
              } else if (MODE == M_FIRST) {
                sovf[q] = 1;                                  // speculation gathered too much: give it up
              } else {                                        // the compaction never drops: global memory
                unsigned long long gb = 0;
                if (lane == leader) gb = atomicAdd(&m.s[q].ps->n_compact, (unsigned long long)cnt);
                gb = __shfl(gb, leader, 64);
                if (in) m.s[q].compact[gb + my] = av;
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < LEAN_MAX; ++q) {
    if (q >= m.ns || !on[q]) continue;
    __syncthreads();                                          // (the reduction helpers share their LDS scratch between calls)
    if (MODE == M_FIRST || MODE == M_PROBE) {
      double acc[PREP_SLOTS];
      pa[q].to_slots(acc);
      block_reduce_store<PREP_SLOTS>(acc, m.s[q].partials, 0);
      if (MODE == M_FIRST) {
        __syncthreads();
        block_max_store<T>(vmax[q], m.s[q].maxpart);
        __syncthreads();
        block_min_store<T>(vminp[q], m.s[q].maxpart + NB);
      }
    } else {
      double a2[2] = {above_s[q], above_c[q]};
      block_reduce_store<2>(a2, m.s[q].partials, SL_ABOVE_S);
    }
    if (GATHERS && gather[q]) {
      __syncthreads();
      const unsigned int cnt = sused[q];                      // reservations grow monotonically: the stored entries form the prefix [0, sused)
      if (threadIdx.x == 0) {
        sbase[q] = cnt ? atomicAdd(&m.s[q].ps->n_compact, (unsigned long long)cnt) : 0ull;
        if (MODE == M_FIRST && sovf[q]) atomicOr(&m.s[q].ps->spec_overflow, 1);
      }
      __syncthreads();
      for (unsigned int i = threadIdx.x; i < cnt; i += BLOCK) m.s[q].compact[sbase[q] + i] = sbuf[GATHERS ? q : 0][i];
    }
  }
}

template <typename T>
void K<T>::pass_multi(int mode, hipStream_t s, const Grid& g, const LeanMulti<T>& m, int v_is_s) {
  if (m.ns < 1 || m.ns > LEAN_MAX || g.n[0] % 4 != 0) throw std::runtime_error("pass_multi: unsupported call");
  double bytes = (double)range_len(g);                 // x once, y and l of every block of every set (gated sets are not booked: see noop_launches)
  for (int q = 0; q < m.ns; ++q) bytes += (v_is_s ? 0.0 : 2.0) * (double)m.s[q].a.nblk_or1() * (double)range_len(g);
  const dim3 grid(fit_grid(range_len(g) / 4, SIPX_PASS_GRID));
  if (mode == M_FIRST) {
    ObsScope obs_(KID_PASS_FIRST, s, bytes * sizeof(T));
    hipLaunchKernelGGL((k_pass_multi<T, 4, M_FIRST>), grid, dim3(BLOCK), 0, s, g, m, v_is_s);
  } else if (mode == M_PROBE) {
    ObsScope obs_(KID_PASS_PROBE, s, bytes * sizeof(T));
    hipLaunchKernelGGL((k_pass_multi<T, 4, M_PROBE>), grid, dim3(BLOCK), 0, s, g, m, v_is_s);
  } else if (mode == M_COMPACT) {
    ObsScope obs_(KID_PASS_COMPACT, s, bytes * sizeof(T));
    hipLaunchKernelGGL((k_pass_multi<T, 4, M_COMPACT>), grid, dim3(BLOCK), 0, s, g, m, v_is_s);
  } else {
    throw std::runtime_error("pass_multi: unknown mode");
  }
  SIPX_HIP(hipGetLastError());
}

template <typename T>
__global__ void k_ps_init(ProjScalars<T>* ps, long long* cidx) {
  ps->cidx = cidx;
  ps->c_lo = ps->c_hi = 0;
  ps->tau_prev = T(0);
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
  ps->spec_lo = 0;
  ps->spec_hi = -1;
  ps->s_above = ps->c_above = 0;
  ps->hw = 1e-2;
  ps->lean = 0;
  ps->spec_ok = ps->spec_overflow = 0;
  ps->tau = T(0);
  ps->quota = 0x7fffffffffffffffll;
  ps->coop_arrive = ps->coop_finish = ps->coop_abort = 0;
  for (int j = 0; j < SAMPLE_BINS; ++j) ps->hist[j] = 0;
  ps->sampled = ps->dbg_sampled = ps->want_sample = 0;
  ps->samp_ticket = 0;
  ps->pass_ticket = 0;
  ps->gather_overflow = 0;
  ps->rounds_used = 0;
  ps->br_tl = ps->br_Sl = ps->br_Cl = ps->br_th = ps->br_Sh = ps->br_Ch = 0;
  ps->ovf = 0;
  for (int r = 0; r < 2 * SIPX_MAX_WORLD; ++r) ps->mm[r] = 0;
  ps->rescaled = 0;
  ps->resc_bad = 1;
  ps->samp_theta = 0;
  ps->samp_bias = 0;
  ps->samp_bias_ok = 0;
}
template <typename T>
void K<T>::ps_init(hipStream_t s, ProjScalars<T>* ps, long long* cidx) {
  hipLaunchKernelGGL((k_ps_init<T>), dim3(1), dim3(1), 0, s, ps, cidx);
  SIPX_HIP(hipGetLastError());
}

// Sums of all PREP_SLOTS partial slots by one workgroup of NT threads.  The loads of CH slots are issued together
// (four at NT = 256), then the slots are reduced wave -> LDS.  The decision kernels run beside the streaming passes of the
// other set stream: a 256-thread workgroup finds a free slot on a busy CU much sooner than a 1024-thread one (+1.8 % at
// 256^3), which outweighs its slower reduction.
#ifndef SIPX_DECIDE_NT
#define SIPX_DECIDE_NT 256
#endif
template <int NT>
__device__ __forceinline__ void reduce_slots(const double* __restrict__ partials, double* red /*LDS, PREP_SLOTS*/) {
  static_assert(NB % NT == 0, "NB must be a multiple of the reducing workgroup");
  constexpr int PER = NB / NT, NW = NT / 64, CH = NT >= 1024 ? PREP_SLOTS : (NT >= 512 ? 8 : 4);
  __shared__ double sm[PREP_SLOTS][NW];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
  for (int k0 = 0; k0 < PREP_SLOTS; k0 += CH) {
    double v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      double a = 0;
      if (k0 + c < PREP_SLOTS) {
#pragma unroll
        for (int j = 0; j < PER; ++j) a += partials[(long long)(k0 + c) * NB + j * NT + threadIdx.x];
      }
      v[c] = a;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const double r = wave_sum(v[c]);
      if (lane == 0 && k0 + c < PREP_SLOTS) sm[k0 + c][w] = r;
    }
  }
  __syncthreads();
  if (threadIdx.x < PREP_SLOTS) {
    double r = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += sm[threadIdx.x][i];
    red[threadIdx.x] = r;
  }
}

// The PREP_SLOTS sums of a probe pass, one workgroup per slot (fixed order), plus one workgroup for the largest and the
// smallest non-zero magnitude.  A single workgroup summing all 19 x 2048 partials took 33 us at 256^3 and 135 us at 512^3
// (its dependent batches of loads queue behind the streaming passes of the other set stream); spread over 20 workgroups
// every load of a slot is in flight at once.
template <typename T>
__device__ __forceinline__ T next_up(T v);
template <>
__device__ __forceinline__ float next_up<float>(float v) { return nextafterf(v, INFINITY); }
template <>
__device__ __forceinline__ double next_up<double>(double v) { return nextafter(v, (double)INFINITY); }
// Probes of a refinement round: L1_K values spread over [lo, hi], each a TF number strictly above the one before and strictly
// inside (tl, th) -- the bracket [lo, hi] left by the Newton / secant steps is often narrower than the spacing of TF numbers,
// and probes that coincide (or fall on tl / th themselves) would not shrink anything.  Unused probes are +Inf (skipped).
template <typename T>
__device__ __forceinline__ void place_probes(ProjScalars<T>* ps, double lo, double hi, double tl, double th) {
  T prev = (T)tl;
  const T top = (T)th;
  for (int k = 0; k < L1_K; ++k) {
    T tc = (T)(lo + (hi - lo) * (double)k / (double)(L1_K - 1));
    if (!(tc > prev)) tc = next_up<T>(prev);
    if (tc < top) { ps->t[k] = (double)tc; prev = tc; }
    else ps->t[k] = INFINITY;
  }
}
template <typename T, int STAGE>
__device__ void decide_body(ProjScalars<T>* ps, int prox, T pmin, T pmax, long long true_len, int nospec, double capdiv, int world,
                            double cap_max, const double* reg);

// FUSE (one rank: no all-reduce between the sums and the decision): every workgroup hands its value over with a device-scope
// store, waits for it and takes a ticket; the workgroup that draws the last one reads them all back and its thread 0 takes
// the scalar decision -- the launch of k_decide, about 6 us on the stream, is saved twice per search.
template <typename T, int STAGE, bool FUSE>
__global__ __launch_bounds__(BLOCK) void k_slot_sums(const double* __restrict__ partials, const T* __restrict__ maxpart,
                                                     ProjScalars<T>* ps, int rank, int world, double* __restrict__ reg, DecideArgs da) {
  // reg: PREP_SLOTS sums | ovf | (max, min) per rank -- ps->red / ovf / mm themselves, or this set's region of the staging
  // buffer that one all-reduce makes global for all sets of a slab-decomposed iteration
  if (STAGE == 1 && !(ps->need && !ps->spec_ok && ps->refine)) {
    // (slab-decomposed: the all-reduce that follows runs regardless; it then sums stale values nobody reads)
    return;
  }
  __shared__ double sreg[PREP_SLOTS + 1 + 2 * SIPX_MAX_WORLD];
  __shared__ unsigned int sh_ticket;
  auto finish = [&]() {      // FUSE: ticket; the last workgroup decides
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) sh_ticket = __hip_atomic_fetch_add(&ps->pass_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sh_ticket != gridDim.x - 1) return;
    const int nreg = PREP_SLOTS + 1 + 2 * world;
    for (int i = threadIdx.x; i < nreg; i += BLOCK) sreg[i] = __hip_atomic_load(&reg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x != 0) return;
    __hip_atomic_store(&ps->pass_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    decide_body<T, STAGE>(ps, da.prox, (T)da.pmin, (T)da.pmax, da.true_len, da.nospec, da.capdiv, world, da.cap_max, sreg);
  };
  if (blockIdx.x < PREP_SLOTS) {
    const double v = block_sum_partials(partials + (long long)blockIdx.x * NB);
    if (threadIdx.x == 0) {
      if (FUSE) __hip_atomic_store(&reg[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else reg[blockIdx.x] = v;
    }
    if (FUSE) finish();
    return;
  }
  __shared__ T smax[BLOCK / 64], smin[BLOCK / 64];
  T vmax = T(0), vmin = (T)INFINITY;
  for (int i = threadIdx.x; i < NB; i += BLOCK) {
    vmax = maxpart[i] > vmax ? maxpart[i] : vmax;
    const T mn = maxpart[NB + i];                 // 0 = entry beyond the pass's grid, or a workgroup that saw no non-zero magnitude
    vmin = (mn > T(0) && mn < vmin) ? mn : vmin;
  }
  vmax = wave_max<T>(vmax);
  vmin = -wave_max<T>(-vmin);
  if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = vmax; smin[threadIdx.x >> 6] = vmin; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < BLOCK / 64; ++i) { vmax = smax[i] > vmax ? smax[i] : vmax; vmin = smin[i] < vmin ? smin[i] : vmin; }
    double* mm = reg + PREP_SLOTS + 1;
    for (int r = 0; r < 2 * world; ++r)
      if (r != 2 * rank && r != 2 * rank + 1) __hip_atomic_store(&mm[r], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mm[2 * rank], (double)vmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mm[2 * rank + 1], (vmin < (T)INFINITY) ? (double)vmin : 0.0, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);                       // 0 = this rank saw no non-zero magnitude
    __hip_atomic_store(&reg[PREP_SLOTS], ps->spec_overflow ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (FUSE) finish();
}

// Scalar decisions after a probe pass (one thread; the sums come from k_slot_sums).  STAGE 0: after the first pass;
// STAGE 1: after the gated refinement.
template <typename T, int STAGE>
__device__ void decide_body(ProjScalars<T>* ps, int prox, T pmin, T pmax, long long true_len, int nospec, double capdiv, int world,
                            double cap_max, const double* reg) {
  const double* red = reg;
  T vmax = T(0), vmin = (T)INFINITY;
  if (STAGE == 0) {
    const double* mm = reg + PREP_SLOTS + 1;
    for (int r = 0; r < world; ++r) {              // (one rank unless slab-decomposed: then red, ovf, mm are all-reduced sums)
      const T mx = (T)mm[2 * r], mn = (T)mm[2 * r + 1];
      vmax = mx > vmax ? mx : vmax;
      vmin = (mn > T(0) && mn < vmin) ? mn : vmin;
    }
    ps->spec_overflow = reg[PREP_SLOTS] > 0 ? 1 : 0;
    ps->vmin = vmin;
    ps->asum = red[0];
    ps->sumsq = red[1];
    ps->vmax = vmax;
    ps->need = 0;
    ps->theta = T(0);
    ps->scale = T(1);
    ps->fill = 0;
    ps->refine = 0;
    ps->rounds_used = 0;
    ps->spec_ok = 0;
    if (prox == PX_L2) {
      const T nl2 = (T)sqrt(red[1]);                     // project_l2!.jl:8-13
      if (!(nl2 <= pmax)) { ps->need = 1; ps->scale = pmax / nl2; }
      return;
    }
    if (prox == PX_ANNULUS) {
      const T nl2 = (T)sqrt(red[1]);                     // project_annulus!.jl:9-18
      if (pmin <= nl2 && nl2 <= pmax) {
      } else if (nl2 > pmax) { ps->need = 1; ps->scale = pmax / nl2; }
      else if (nl2 < pmin && nl2 > T(0)) { ps->need = 1; ps->scale = pmin / nl2; }
      else if (nl2 < pmin && nl2 == T(0)) {              // sigma_min ./ sqrt(length(x)): Float64 sqrt of an Int
        ps->need = 1; ps->fill = 1;
        ps->scale = (T)((double)pmin / sqrt((double)true_len));
      }
      return;
    }
    ps->need = ((T)red[0] <= pmax) ? 0 : 1;              // norm(v,1) <= b && return v   project_l1_Duchi!.jl:23
    if (!ps->need) {
      ps->n_compact = 0;
      ps->spec_overflow = 0;
      ps->lean = 0;
      return;
    }
    if (ps->lean) {
      // LEAN first pass: only (S, C) at the two edges of the speculative range are known.  theta* lies inside iff
      // f(spec_lo) >= 0 > f(spec_hi); then the bracket is tightened by the Newton / secant steps as usual and the gathered
      // magnitudes are all that is needed.  Otherwise the root left the range: one-sided geometric probes for the refinement
      // pass (the same number of sweeps as a failed speculation of the full pass), and the next first pass is a full one.
      const double b = (double)pmax;
      const double t3 = ps->t[L1_WIN_LO], t4 = ps->t[L1_WIN_HI];
      const double S3 = red[3 + L1_WIN_LO], C3 = red[3 + L1_K + L1_WIN_LO], S4 = red[3 + L1_WIN_HI], C4 = red[3 + L1_K + L1_WIN_HI];
      const double f3 = S3 - t3 * C3 - b, f4 = S4 - t4 * C4 - b;
      ps->lean = 0;
      // (slab-decomposed: a range that holds more than the exchange takes counts as an overflow -- the next range is half as wide)
      if (cap_max > 0 && C3 - C4 > cap_max) ps->spec_overflow = 1;
      if (f3 >= 0 && f4 < 0 && !ps->spec_overflow && !(nospec & 1)) {
        double thN = C3 > 0 ? (S3 - b) / C3 : t3;
        if (!(thN >= t3)) thN = t3;
        double thS = t3 + f3 * (t4 - t3) / (f3 - f4);
        if (!(thS <= t4)) thS = t4;
        if (!(thS >= thN)) thS = t4;
        double lo = thN * (1.0 - 1e-9), hi = thS * (1.0 + 1e-9) + 1e-300;
        ps->lo = lo > t3 ? lo : t3;
        ps->hi = hi < t4 ? hi : t4;
        ps->spec_ok = 1;
        ps->s_above = S4;
        ps->c_above = C4;
        ps->vmax = (T)INFINITY;                              // not measured by a lean pass (unused on this route)
        return;
      }
      ps->n_compact = 0;                                     // discard what the speculation gathered
      ps->vmax = (T)ps->asum;                                // a valid upper bound of every magnitude
      if (f3 >= 0 && f4 < 0) {
        // theta* IS inside the range, but what the range gathered cannot be used (an LDS buffer or an exchange segment overflowed,
        // or it holds more than the caller can take): the bracket is the range tightened by the Newton / secant steps; it is
        // gathered at once by the compaction pass if it is small enough, and subdivided by refinement rounds first if not
        double thN = C3 > 0 ? (S3 - b) / C3 : t3;
        if (!(thN >= t3)) thN = t3;
        double thS = t3 + f3 * (t4 - t3) / (f3 - f4);
        if (!(thS <= t4)) thS = t4;
        if (!(thS >= thN)) thS = t4;
        double lo = thN * (1.0 - 1e-9), hi = thS * (1.0 + 1e-9) + 1e-300;
        lo = lo > t3 ? lo : t3;
        hi = hi < t4 ? hi : t4;
        ps->lo = lo;
        ps->hi = hi;
        double cap = fmax(L1_CAP, (double)true_len / capdiv);
        if (cap_max > 0 && cap > cap_max) cap = cap_max;
        double pop = C3 - C4;
        if (t4 > t3 && !(cap_max > 0)) pop *= fmin(1.0, 2.0 * (hi - lo) / (t4 - t3));
        ps->br_tl = t3; ps->br_Sl = S3; ps->br_Cl = C3; ps->br_th = t4; ps->br_Sh = S4; ps->br_Ch = C4;
        if (next_up<T>((T)t3) >= (T)t4) {                    // no magnitude can lie strictly between: see the end of this function
          ps->lo = ps->hi = t3;
        } else if (pop > cap && hi > lo) {
          ps->refine = 1;
          ps->rounds_used = 1;
          place_probes<T>(ps, lo, hi, t3, t4);
        }
        return;
      }
      ps->refine = 1;
      ps->rounds_used = 1;
      if (f4 >= 0) {                                         // theta* >= spec_hi: geometric probes above it
        ps->lo = t4;
        ps->hi = ps->asum;
        double step = (t4 - t3) > 0 ? (t4 - t3) : t4 * 1e-3;
        for (int k = 0; k < L1_K; ++k) { step *= 4.0; ps->t[k] = (double)(T)(t4 + step); }
        ps->br_tl = t4; ps->br_Sl = S4; ps->br_Cl = C4; ps->br_th = ps->asum; ps->br_Sh = 0; ps->br_Ch = 0;
      } else {                                               // theta* < spec_lo: geometric probes below it
        ps->lo = 0;
        ps->hi = t3;
        double step = (t4 - t3) > 0 ? (t4 - t3) : t3 * 1e-3;
        for (int k = L1_K - 1; k >= 0; --k) { step *= 4.0; const double t = t3 - step; ps->t[k] = t > 0 ? (double)(T)t : 0.0; }
        // (a lean pass does not count the non-zero entries: C(0) unknown until a probe below theta has been evaluated)
        ps->br_tl = 0; ps->br_Sl = ps->asum; ps->br_Cl = -1.0; ps->br_th = t3; ps->br_Sh = S3; ps->br_Ch = C3;
      }
      return;
    }
  } else {
    vmax = ps->vmax;
  }
  const double b = (double)pmax;
  // The two probes that bracket theta*, with their exact (S, C): tl (f >= 0) and th (f < 0).  STAGE 0 starts from the virtual
  // probes t = 0 (S = ||v||_1, C = nnz) and t = vmax (S = C = 0); a refinement round starts from the pair the previous decision
  // left in ps->br_* -- so the bracket only ever shrinks and C(tl) - C(th) is the exact count of what lies in (tl, th].
  double tl = 0, Sl = ps->asum, Cl = red[2];
  double th = (double)vmax, Sh = 0, Ch = 0;
  if (STAGE == 1) { tl = ps->br_tl; Sl = ps->br_Sl; Cl = ps->br_Cl; th = ps->br_th; Sh = ps->br_Sh; Ch = ps->br_Ch; }
  for (int k = 0; k < L1_K; ++k) {
    const double t = ps->t[k];
    if (!(t < INFINITY)) continue;
    const double S = red[3 + k], C = red[3 + L1_K + k];
    const double f = S - t * C - b;
    if (f >= 0) {
      if (t >= tl) { tl = t; Sl = S; Cl = C; }
    } else if (t <= th) {
      th = t; Sh = S; Ch = C;
    }
  }
  const bool know_l = Cl >= 0;                 // (C(0) is not known after a lean pass)
  const double fl = know_l ? Sl - tl * Cl - b : INFINITY, fh = Sh - th * Ch - b;
  // Newton from the left (Michelot step) and secant from the right: theta* in [thN, thS]
  double thN = (know_l && Cl > 0) ? (Sl - b) / Cl : tl;
  if (!(thN >= tl)) thN = tl;
  double thS = th;
  if (fl < INFINITY && fl - fh > 0) thS = tl + fl * (th - tl) / (fl - fh);
  if (!(thS <= th)) thS = th;
  if (!(thS >= thN)) thS = th;
  double lo = thN * (1.0 - 1e-9), hi = thS * (1.0 + 1e-9) + 1e-300;
  lo = lo > tl ? lo : tl;
  hi = hi < th ? hi : th;
  ps->lo = lo;
  ps->hi = hi;
  ps->br_tl = tl; ps->br_Sl = Sl; ps->br_Cl = Cl; ps->br_th = th; ps->br_Sh = Sh; ps->br_Ch = Ch;
  if (STAGE == 0) {
    // speculative gather usable?  range edges are probes L1_WIN_LO and L1_WIN_HI, so (S,C) above it are known
    // (slab-decomposed: what the range gathered over all ranks, C(spec_lo) - C(spec_hi), has to fit a rank's exchange segment)
    if (cap_max > 0 && red[3 + L1_K + L1_WIN_LO] - red[3 + L1_K + L1_WIN_HI] > cap_max) ps->spec_overflow = 1;
    const bool spec = !(nospec & 1) && ps->spec_hi > ps->spec_lo && !ps->spec_overflow && lo >= ps->spec_lo && hi <= ps->spec_hi;
    if (spec) {
      ps->spec_ok = 1;
      ps->s_above = red[3 + L1_WIN_HI];
      ps->c_above = red[3 + L1_K + L1_WIN_HI];
      return;
    }
    ps->n_compact = 0;                                   // discard what the speculation gathered
  }
  // cold start / theta moved far: while the bracket still holds too many magnitudes, subdivide it again
  // (each gated refinement pass narrows it by >= L1_K-1 and by the Newton/secant step on top)
  // one more probe pass costs a full sweep of the vector, gathering a larger bracket costs the one-workgroup solve a
  // longer scan: the break-even population grows with the length (measured at 256^3 and 512^3)
  double cap = fmax(L1_CAP, (double)true_len / capdiv);
  if (cap_max > 0 && cap > cap_max) cap = cap_max;
  // population of the tightened bracket (lo, hi]: the count between the two probes, scaled by the share of the interval
  // that is left (factor 2 for a density that is not flat).  A wrong guess only costs time: the gather never drops.
  double pop = (know_l ? Cl : (double)true_len) - Ch;
  // Slab-decomposed (cap_max > 0): what the final bracket gathers over ALL ranks has to fit the exchange segments, and a
  // rank cannot keep what does not fit -- so the count between the two bracketing probes is taken as it is (an upper bound of
  // what (lo, hi] holds), and refinement goes on, round after round, until it fits.
  if (th > tl && !(cap_max > 0)) pop *= fmin(1.0, 2.0 * (hi - lo) / (th - tl));
  const int max_refines = cap_max > 0 ? L1_REFINES_SLAB : L1_REFINES;
  if (know_l && next_up<T>((T)tl) >= (T)th) {
    // tl and th are neighbours in TF: no magnitude lies strictly between them, so {|v| > t} is the same set for every t in
    // [tl, th) and f is linear there -- its root, the Newton step from tl, IS theta*.  Nothing has to be gathered, however many
    // magnitudes equal th (ties: no bracket could separate them): the compaction pass sums what lies above tl and gathers nothing.
    ps->lo = ps->hi = tl;
    ps->refine = 0;
  } else if (pop > cap && hi > lo && (STAGE == 0 || ps->refine < max_refines)) {
    ps->refine = (STAGE == 0) ? 1 : ps->refine + 1;
    ps->rounds_used = ps->refine;
    place_probes<T>(ps, lo, hi, tl, th);
  } else {
    ps->refine = 0;
  }
}

// the decision as a kernel of its own (slab-decomposed grid: an all-reduce sits between the sums and the decision)
template <typename T, int STAGE>
__global__ __launch_bounds__(64) void k_decide(ProjScalars<T>* ps, int prox, T pmin, T pmax, long long true_len,
                                               int nospec, double capdiv, int world, double cap_max, const double* __restrict__ reg) {
  if (STAGE == 1 && !(ps->need && !ps->spec_ok && ps->refine)) return;
  if (threadIdx.x != 0) return;
  decide_body<T, STAGE>(ps, prox, pmin, pmax, true_len, nospec, capdiv, world, cap_max, reg);
}


// ... and, in the fallback of a speculative exchange, with the verdict the host reads: does this set need another refinement round?
template <typename T>
__global__ __launch_bounds__(64) void k_decide_round(ProjScalars<T>* ps, int prox, T pmin, T pmax, long long true_len, double capdiv,
                                                     int world, double cap_max, const double* __restrict__ reg, unsigned seq,
                                                     unsigned* verdict) {
  if (threadIdx.x != 0) return;
  const bool live = ps->need && !ps->spec_ok && ps->refine;
  if (live) decide_body<T, 1>(ps, prox, pmin, pmax, true_len, 0, capdiv, world, cap_max, reg);
  const unsigned word = (seq << 2) | 1u | ((live && ps->refine) ? 2u : 0u);
  __hip_atomic_store(verdict, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// SAMPLED PREDICTION of theta.  While rho and gamma are still being adapted, theta moves by up to a factor of three from one
// PARSDMM iteration to the next: the speculative gather around the previous theta fails and the search pays two more sweeps
// of the vector (refinement + compaction, 6 N w bytes per set).  A sweep over a SAMPLE of the vector (every stride-th run of
// 64 consecutive entries, the run inside its group chosen by a hash so that no lattice direction is favoured -- neighbouring
// entries are correlated, so many short runs beat few long ones; about a million entries) costs a hundredth of that and
// predicts theta to a few tenths of a percent: the sampled magnitudes go into a histogram (bin key = leading bits of the
// floating-point pattern, 128 bins per octave, 8 octaves around the old prediction; count and FIXED-POINT sum packed in one
// word and added through integer atomics, so the totals do not depend on the order of arrival), the workgroup that finishes
// last takes suffix sums and finds the bin in which f_sample(t) = sum(max(|v|-t,0)) - b n_sample/n changes sign.  (S, C) at
// the bin's edges are exact for the sample, so its root lies between the Newton step from the lower edge and the secant
// (f is convex) whatever the density inside the bin does -- near convergence the magnitudes pile up right at theta.  That
// interval, widened by four standard deviations of the sampling error sqrt(sum_sample max(|v|-theta,0)^2) / C, becomes the
// speculative range of the first pass (shrunk again if the bins it covers promise more than the gather can hold), which is
// then a lean one.  Nothing of the result depends on the sample: a miss is caught by the pass's own bracket test and costs
// the fallback sweeps it would have cost anyway.
template <typename T>
struct KeyBits;
template <>
struct KeyBits<float> {
  typedef unsigned int U;
  static constexpr int SH = 23 - SAMPLE_MBITS;
  static constexpr long long KEY_INF = 0xffll << SAMPLE_MBITS;
  static __device__ __forceinline__ long long key(float v) { return (long long)(__float_as_uint(v) >> SH); }
  static __device__ __forceinline__ double edge(long long k) { return (double)__uint_as_float((unsigned int)k << SH); }
};
template <>
struct KeyBits<double> {
  typedef unsigned long long U;
  static constexpr int SH = 52 - SAMPLE_MBITS;
  static constexpr long long KEY_INF = 0x7ffll << SAMPLE_MBITS;
  static __device__ __forceinline__ long long key(double v) { return (long long)((unsigned long long)__double_as_longlong(v) >> SH); }
  static __device__ __forceinline__ double edge(long long k) { return __longlong_as_double((long long)((unsigned long long)k << SH)); }
};
template <typename T>
__device__ __forceinline__ long long sample_key_lo(const ProjScalars<T>* ps) {
  long long k = KeyBits<T>::key((T)ps->theta_prev) - SAMPLE_BINS / 2;
  k = k < 1 ? 1 : k;
  const long long kmax = KeyBits<T>::KEY_INF - SAMPLE_BINS - 1;
  return k > kmax ? kmax : k;
}
constexpr double SAMPLE_FIX = 1048576.0;          // 2^SAMPLE_VAL_BITS

constexpr int SAMPLE_NT = 512;                   // threads of a k_sample workgroup
constexpr int SAMPLE_WG = 256;                   // workgroups of k_sample (each merges its LDS histogram into the global one)
constexpr int SAMPLE_RUN = 16;                   // lanes (x V grid points) of one sampled run: 64 consecutive entries.  Neighbouring
                                                 // entries are correlated, so many short runs beat few long ones
// A histogram bin is ONE 64-bit word: the count in the upper SAMPLE_CNT_BITS bits, the fixed-point sum of the magnitudes below
// (top edge of the range = 2^SAMPLE_VAL_BITS: at most 2^SAMPLE_CNT_BITS values fit without a carry into the count) -- one LDS
// atomic per sampled entry, one global atomic per non-empty bin and workgroup.
constexpr int SAMPLE_CNT_BITS = 22, SAMPLE_VAL_BITS = 20;
static_assert(SAMPLE_CNT_BITS + SAMPLE_CNT_BITS + SAMPLE_VAL_BITS <= 64, "sum field must hold 2^CNT values of 2^VAL");
constexpr unsigned long long SAMPLE_SUM_MASK = (1ull << (64 - SAMPLE_CNT_BITS)) - 1ull;

// Second half of the sampled prediction, run by the workgroup of k_sample that finishes last: suffix sums over the bins,
// the bin of the sample's root, its Newton / secant bounds, the sampling error, the probes of the coming first pass.
// Everything the other workgroups contributed was written with device-scope atomics and is read with device-scope loads.
template <typename T>
__device__ __forceinline__ void sample_decide(ProjScalars<T>* ps, double* __restrict__ partials, int nwg, T radius,
                                              long long true_len, double hw_max, int lean_on, double gather_cap, double* sS,
                                              double* sC, const double* __restrict__ dsrc = nullptr) {
  // dsrc (slab-decomposed grid): counts [0, BINS), fixed-point sums [BINS, 2 BINS) and the three partial sums, all-reduced
  // over the ranks as float64 (integers below 2^53: exact, whatever the order) -- instead of ps->hist and the partial slots
  constexpr int NT = SAMPLE_NT, PER = SAMPLE_BINS / NT, NW = NT / 64;
  static_assert(SAMPLE_BINS % NT == 0 && NW <= 16, "bins per thread");
  __shared__ int sh_bin, sh_ok;
  __shared__ double sh_thN, sh_thS, sh_cact, sh_lo, sh_hi;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = threadIdx.x * PER;
  // all loads first: this thread's bins, its share of the three partial sums of the nwg workgroups
  unsigned long long word[PER];
  double dc[PER], dsum[PER];
  double v[3] = {0, 0, 0};
  if (dsrc) {
#pragma unroll
    for (int i = 0; i < PER; ++i) { dc[i] = dsrc[j0 + i]; dsum[i] = dsrc[SAMPLE_BINS + j0 + i]; word[i] = 0; }
    if (threadIdx.x < 3) v[threadIdx.x] = dsrc[2 * SAMPLE_BINS + threadIdx.x];
    // (thread k holds sum k: spread so that the wave sums below deliver all three)
    const double a0 = threadIdx.x == 0 ? v[0] : 0.0, a1 = threadIdx.x == 1 ? v[1] : 0.0, a2 = threadIdx.x == 2 ? v[2] : 0.0;
    v[0] = a0; v[1] = a1; v[2] = a2;
  } else {
#pragma unroll
    for (int i = 0; i < PER; ++i) word[i] = __hip_atomic_load(&ps->hist[j0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = threadIdx.x; i < nwg; i += NT)
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] += __hip_atomic_load(&partials[(long long)k * NB + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < PER; ++i) __hip_atomic_store(&ps->hist[j0 + i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next search
  }
  const long long key_lo = sample_key_lo<T>(ps);
  const double fscale = SAMPLE_FIX / KeyBits<T>::edge(key_lo + SAMPLE_BINS);
  double cnt[PER], sum[PER];
  double tS = 0, tC = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    cnt[i] = dsrc ? dc[i] : (double)(word[i] >> (64 - SAMPLE_CNT_BITS));
    sum[i] = (dsrc ? dsum[i] : (double)(word[i] & SAMPLE_SUM_MASK)) / fscale;
    tS += sum[i];
    tC += cnt[i];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = wave_sum(v[k]);
  double sufS = tS, sufC = tC;                  // inclusive suffix sums inside the wave (fixed order)
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double aS = __shfl_down(sufS, d, 64), aC = __shfl_down(sufC, d, 64);
    if (lane + d < 64) { sufS += aS; sufC += aC; }
  }
  const double exS = __shfl_down(sufS, 1, 64), exC = __shfl_down(sufC, 1, 64);     // of the lanes above this one
  if (lane == 0) { sS[wv] = sufS; sC[wv] = sufC; sS[16 + wv] = v[0]; sS[32 + wv] = v[1]; sS[48 + wv] = v[2]; }
  if (threadIdx.x == 0) { sh_bin = SAMPLE_BINS; sh_ok = 0; }
  __syncthreads();
  double S_top = 0, C_top = 0, n_val = 0;       // above the histogram's range; valid entries sampled
  for (int i = 0; i < NW; ++i) { S_top += sS[16 + i]; C_top += sS[32 + i]; n_val += sS[48 + i]; }
  // (sum, count) of the sampled magnitudes above the upper edge of this thread's last bin
  double Sab = S_top + (lane < 63 ? exS : 0.0), Cab = C_top + (lane < 63 ? exC : 0.0);
  for (int i = NW - 1; i > wv; --i) { Sab += sS[i]; Cab += sC[i]; }
  const double bs = (double)radius * (n_val / (double)true_len);          // the ball's radius, scaled to the sample
  double e_up = KeyBits<T>::edge(key_lo + j0 + PER);
  double f_up = Sab - e_up * Cab - bs;
  int my_bin = SAMPLE_BINS;
  double r_thN = 0, r_thS = 0, r_cact = 0;
#pragma unroll
  for (int i = PER - 1; i >= 0; --i) {
    Sab += sum[i];
    Cab += cnt[i];
    const double e_lo = KeyBits<T>::edge(key_lo + j0 + i);
    const double f_lo = Sab - e_lo * Cab - bs;
    if (f_lo >= 0 && f_up < 0) {
      my_bin = j0 + i;
      double thN = Cab > 0 ? (Sab - bs) / Cab : e_lo;                     // Newton from the lower edge: <= root
      thN = thN < e_lo ? e_lo : (thN > e_up ? e_up : thN);
      double thS = e_lo + f_lo * (e_up - e_lo) / (f_lo - f_up);           // secant: >= root
      thS = thS < thN ? thN : (thS > e_up ? e_up : thS);
      r_thN = thN; r_thS = thS; r_cact = Cab - 0.5 * cnt[i];
    }
    e_up = e_lo;
    f_up = f_lo;
  }
  if (my_bin < SAMPLE_BINS) atomicMin(&sh_bin, my_bin);
  __syncthreads();
  if (sh_bin == SAMPLE_BINS) return;            // the sample's root is outside the histogram: keep the old prediction
  if (my_bin == sh_bin) { sh_thN = r_thN; sh_thS = r_thS; sh_cact = r_cact; }
  __syncthreads();
  // sampling error of the root: sqrt(sum over the active sample of (|v| - theta)^2) / C, from the bins above the root's
  const double th = 0.5 * (sh_thN + sh_thS);
  double q = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (j0 + i > sh_bin) {
      const double mid = 0.5 * (KeyBits<T>::edge(key_lo + j0 + i) + KeyBits<T>::edge(key_lo + j0 + i + 1)) - th;
      q += cnt[i] * mid * mid;
    }
  }
  q = wave_sum(q);
  if (lane == 0) sC[wv] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double c_act = sh_cact;
    if (th > 0 && c_act >= 16.0) {
      double Q = 0;
      for (int i = 0; i < NW; ++i) Q += sC[i];
      if (C_top > 0) { const double mt = S_top / C_top - th; Q += C_top * mt * mt; }
      double m = 4.0 * sqrt(Q) / (c_act * th);        // four standard deviations, relative
      m = m < 5e-4 ? 5e-4 : (m > 0.5 ? 0.5 : m);
      // the estimate corrected by what the last sampled estimate of this set was off by (the same entries are sampled every time:
      // its error persists from iteration to iteration); a correction beyond a tenth of theta is not believed
      double bias = ps->samp_bias_ok ? ps->samp_bias : 0.0;
      if (!(fabs(bias) <= 0.1 * th)) bias = 0.0;
      sh_thN -= bias;
      sh_thS -= bias;
      sh_lo = sh_thN * (1.0 - m);
      sh_hi = sh_thS * (1.0 + m);
      sh_ok = 1;
    }
  }
  __syncthreads();
  if (!sh_ok) return;
  // How many magnitudes of the WHOLE vector will the range (lo, hi] gather?  The bins it touches, scaled by the sampling
  // ratio.  The relative width does not matter (when theta is small against the spread of the values its sampling error is
  // several percent, yet few values lie that close to it): the capacity of the speculative gather does.
  double gcnt = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const double e0 = KeyBits<T>::edge(key_lo + j0 + i), e1 = KeyBits<T>::edge(key_lo + j0 + i + 1);
    if (e1 > sh_lo && e0 <= sh_hi) gcnt += cnt[i];
  }
  gcnt = wave_sum(gcnt);
  if (lane == 0) sS[wv] = gcnt;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double G = 0;
  for (int i = 0; i < NW; ++i) G += sS[i];
  G *= (double)true_len / n_val;
  double lo = sh_lo, hi = sh_hi;
  if (G > gather_cap) {            // shrink towards the Newton / secant interval (a miss only costs the fallback sweeps)
    const double f = gather_cap / G;
    lo = sh_thN - (sh_thN - lo) * f;
    hi = sh_thS + (hi - sh_thS) * f;
  }
  const double c_act = sh_cact;
  const double ctr = 0.5 * (lo + hi);
  double hw = (hi - lo) / (2.0 * ctr);
  hw = hw < 1e-3 ? 1e-3 : hw;
  ps->hw = hw < hw_max ? hw : hw_max;      // (the next search's own rule starts from a range of the usual width)
  for (int k = 0; k < L1_K; ++k) {
    const double t = ctr * (1.0 + hw * l1_probe_mult(k));
    ps->t[k] = t > 0 ? (double)(T)t : 0.0;
  }
  ps->spec_lo = ps->t[L1_WIN_LO];
  ps->spec_hi = ps->t[L1_WIN_HI];
  ps->lean = lean_on ? 1 : 0;
  ps->sampled = 1;
  ps->samp_theta = th;
  ps->samp_lo = sh_thN; ps->samp_hi = sh_thS; ps->samp_c = c_act;
}

// (the body as a device function: k_sample runs it for one set, k_sample_multi for every sampling set of a slab-decomposed
//  iteration, grid.y = set; it uses blockIdx.x / gridDim.x only)
template <typename T, int V>
__device__ void sample_body(const Grid& G, const SetArgs<T>& a, ProjScalars<T>* ps, double* __restrict__ partials,
                            long long nchunks, long long nsamp, unsigned int stride, long long true_len,
                            double hw_max, int lean_on, double gather_cap, double* __restrict__ defer_to, int v_is_s = 0) {
  if (!ps->want_sample || !(ps->theta_prev > 0)) return;
  constexpr int NT = SAMPLE_NT;
  __shared__ unsigned long long hs[SAMPLE_BINS];
  __shared__ double sS[NT], sC[NT];
  __shared__ unsigned int sh_ticket;
  for (int j = threadIdx.x; j < SAMPLE_BINS; j += NT) hs[j] = 0;
  __syncthreads();
  const long long key_lo = sample_key_lo<T>(ps);
  const double fscale = SAMPLE_FIX / KeyBits<T>::edge(key_lo + SAMPLE_BINS);
  double acc[3] = {0, 0, 0};      // sum and count of the sampled magnitudes above the histogram's range, valid entries sampled
  const bool ident = a.nblk == 0;
  const int nb = ident ? 1 : a.nblk;
  const bool relax = !(a.gamma == T(1));
  const T gam = a.gamma, omg = T(1) - a.gamma;
  long long v0, nvec;
  vec_range<V>(G, v0, nvec);
  // sampled run u of SAMPLE_RUN lanes -> run u * stride + hash(u) mod stride of the vector (one out of every `stride` runs,
  // chosen by a hash so that no lattice direction of the grid is favoured)
  const long long total = nsamp * SAMPLE_RUN;      // lanes to process
  for (long long t0 = (long long)blockIdx.x * NT; t0 < total; t0 += (long long)gridDim.x * NT) {
    const long long t = t0 + threadIdx.x;
    const long long u = t / SAMPLE_RUN;
    long long ru = u * stride + (long long)((unsigned int)(((unsigned long long)u * 2654435761ull) >> 13) % stride);
    if (ru >= nchunks) ru = u * stride;
    const long long vi = t < total ? v0 + ru * SAMPLE_RUN + (t % SAMPLE_RUN) : nvec;
    const bool live = vi < nvec;
    const long long g = live ? vi * V : v0 * V;      // (lanes beyond the range shadow its first point: every array is backed there)
    const Coord cd = coords(G, g);
    const Vec<T, V> xc = ldv<T, V>(a.x + g);
    for (int q = 0; q < nb; ++q) {
      const long long e = (long long)q * G.N + g;
      T s[V];
      bool valid[V];
      if (ident) {
#pragma unroll
        for (int k = 0; k < V; ++k) { s[k] = xc.v[k]; valid[k] = true; }
      } else {
        fwd_dir<T, V>(G, a.x, xc, g, cd, a.dir[q], a.ih[q], s, valid);
      }
      Vec<T, V> yv, lv;
      if (!v_is_s) { yv = ldv<T, V>(a.y + e); lv = ldv<T, V>(a.l + e); }
#pragma unroll
      for (int k = 0; k < V; ++k) {
        if (!(live && valid[k])) continue;
        T av;
        if (v_is_s) {
          av = fabs(s[k]);                                              // the feasibility estimate projects s = A x itself
        } else {
          const T xh = relax ? (gam * s[k] + omg * yv.v[k]) : s[k];     // update_y_l.jl:72
          av = fabs(xh - lv.v[k] * a.rho1);                             // :67 / :74
        }
        acc[2] += 1.0;
        const long long j = KeyBits<T>::key(av) - key_lo;
        if (j >= SAMPLE_BINS) {
          acc[0] += (double)av;
          acc[1] += 1.0;
        } else if (j >= 0) {
          atomicAdd(&hs[j], (1ull << (64 - SAMPLE_CNT_BITS)) + (unsigned long long)((double)av * fscale + 0.5));
        }
      }
    }
  }
  // this workgroup's three sums -> partials[k * NB + blockIdx.x] (fixed order)
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = wave_sum(acc[k]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sS[threadIdx.x >> 6] = acc[0]; sC[threadIdx.x >> 6] = acc[1]; sS[64 + (threadIdx.x >> 6)] = acc[2]; }
  __syncthreads();
  // What leaves the workgroup goes through DEVICE-SCOPE ATOMICS (performed at the coherence point, not in this XCD's L2), so no
  // cache write-back / invalidation is needed to hand it to the workgroup that decides: a __threadfence() pair here flushes
  // the L2 of work the other set stream left dirty and cost 15 of the kernel's 40 us.
  if (threadIdx.x == 0) {
    double t0 = 0, t1 = 0, t2 = 0;
    for (int i = 0; i < NT / 64; ++i) { t0 += sS[i]; t1 += sC[i]; t2 += sS[64 + i]; }
    __hip_atomic_store(&partials[blockIdx.x], t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partials[NB + blockIdx.x], t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partials[2 * NB + blockIdx.x], t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int j = threadIdx.x; j < SAMPLE_BINS; j += NT)
    if (hs[j]) __hip_atomic_fetch_add(&ps->hist[j], hs[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // every thread waits until its own atomics have been performed, then the workgroup takes a ticket: the one that arrives last
  // finds every other one's contribution in place and decides
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) sh_ticket = __hip_atomic_fetch_add(&ps->samp_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (sh_ticket != gridDim.x - 1) return;
  if (threadIdx.x == 0) __hip_atomic_store(&ps->samp_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (defer_to) {
    // slab-decomposed grid: this is one rank's share of the sample.  Counts, fixed-point sums and the three partial sums go out
    // as float64 (exact integers), an all-reduce adds the ranks' shares, k_sample_decide2 takes it from there.
    for (int j = threadIdx.x; j < SAMPLE_BINS; j += NT) {
      const unsigned long long w = __hip_atomic_load(&ps->hist[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ps->hist[j], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      defer_to[j] = (double)(w >> (64 - SAMPLE_CNT_BITS));
      defer_to[SAMPLE_BINS + j] = (double)(w & SAMPLE_SUM_MASK);
    }
    // the three partial sums of the workgroups: every thread one load (a single thread walking them took longer than the sampling)
    double t3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double t = 0;
      for (int i = threadIdx.x; i < (int)gridDim.x; i += NT)
        t += __hip_atomic_load(&partials[(long long)k * NB + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t3[k] = wave_sum(t);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sS[threadIdx.x >> 6] = t3[0]; sC[threadIdx.x >> 6] = t3[1]; sS[64 + (threadIdx.x >> 6)] = t3[2]; }
    __syncthreads();
    if (threadIdx.x < 3) {
      const double* src = threadIdx.x == 0 ? sS : (threadIdx.x == 1 ? sC : sS + 64);
      double t = 0;
      for (int i = 0; i < NT / 64; ++i) t += src[i];
      defer_to[2 * SAMPLE_BINS + threadIdx.x] = t;
    }
    return;
  }
  // (a feasibility estimate's last sample is ten iterations old: what it was off by then says nothing now -- iteration 20 of the
  //  headline run missed its range WITH the correction of iteration 10)
  if (v_is_s && threadIdx.x == 0) ps->samp_bias_ok = 0;
  sample_decide<T>(ps, partials, (int)gridDim.x, a.phi, true_len, hw_max, lean_on, gather_cap, sS, sC);
}

template <typename T, int V>
__global__ __launch_bounds__(SAMPLE_NT) void k_sample(Grid G, SetArgs<T> a, ProjScalars<T>* ps, double* __restrict__ partials,
                                                      long long nchunks, long long nsamp, unsigned int stride, long long true_len,
                                                      double hw_max, int lean_on, double gather_cap, double* __restrict__ defer_to, int v_is_s) {
  sample_body<T, V>(G, a, ps, partials, nchunks, nsamp, stride, true_len, hw_max, lean_on, gather_cap, defer_to, v_is_s);
}
template <typename T, int V>
__global__ __launch_bounds__(SAMPLE_NT) void k_sample_multi(Grid G, SampleMulti<T> A, long long nchunks, long long nsamp, unsigned int stride,
                                                            double hw_max, int lean_on, double gather_cap) {
  const SampleSet<T>& S = A.s[blockIdx.y];
  sample_body<T, V>(G, S.a, S.ps, S.partials, nchunks, nsamp, stride, S.true_len, hw_max, lean_on, gather_cap, S.reg, A.v_is_s);
}
template <typename T>
__global__ __launch_bounds__(SAMPLE_NT) void k_sample_decide2_multi(SampleMulti<T> A, double hw_max, int lean_on, double gather_cap) {
  const SampleSet<T>& S = A.s[blockIdx.x];
  if (!S.ps->want_sample || !(S.ps->theta_prev > 0)) return;
  __shared__ double sS[SAMPLE_NT], sC[SAMPLE_NT];
  sample_decide<T>(S.ps, nullptr, 0, S.a.phi, S.true_len, hw_max, lean_on, gather_cap, sS, sC, S.reg);
}

template <typename T>
__global__ __launch_bounds__(SAMPLE_NT) void k_sample_decide2(ProjScalars<T>* ps, const double* __restrict__ dsrc, T radius,
                                                              long long true_len, double hw_max, int lean_on, double gather_cap) {
  if (!ps->want_sample || !(ps->theta_prev > 0)) return;
  __shared__ double sS[SAMPLE_NT], sC[SAMPLE_NT];
  sample_decide<T>(ps, nullptr, 0, radius, true_len, hw_max, lean_on, gather_cap, sS, sC, dsrc);
}

// Double-double accumulation (Knuth's TwoSum): the gathered magnitudes arrive in an order that changes from run to run (they
// are compacted through atomics), and a plain float64 sum of them would change in its last bits with that order -- and the
// threshold, and every y, l and x after it, with it.  Carried as (hi, lo) pairs the sum is good to ~1e-32 relative whatever
// the order, so its rounding to float64 is the same in every run (short of the exact sum sitting on a rounding boundary).
struct DD {
  double hi, lo;
};
__device__ __forceinline__ DD dd_add(DD a, double b) {
  const double s = a.hi + b, bb = s - a.hi;
  const double e = (a.hi - (s - bb)) + (b - bb);
  return DD{s, a.lo + e};
}
__device__ __forceinline__ DD dd_add(DD a, DD b) {
  const double s = a.hi + b.hi, bb = s - a.hi;
  const double e = (a.hi - (s - bb)) + (b.hi - bb);
  const double lo = (a.lo + b.lo) + e;
  const double hi = s + lo;                   // renormalise
  return DD{hi, lo - (hi - s)};
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  const long long b = __double_as_longlong(v);
  const int lo = __shfl_xor((int)(b & 0xffffffffll), m, 64), hi = __shfl_xor((int)(b >> 32), m, 64);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ DD wave_sum_dd(DD v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = dd_add(v, DD{shfl_xor_f64(v.hi, m), shfl_xor_f64(v.lo, m)});
  return v;
}

// Michelot's iteration on the gathered magnitudes: theta <- (S_above + S_in(>theta) - b) / (C_above + C_in(>theta)),
// monotone from the bracket's lower end, exact after finitely many steps (stops when the active count repeats).
// Then prepares the next call: probes and speculative range centred on the new theta.
#ifndef SIPX_SOLVE_NT
#define SIPX_SOLVE_NT 1024
#endif
// COOPERATIVE SWEEPS.  The kernel is launched with SOLVE_G workgroups.  A small gather (the rule while the speculation holds) is
// handled by workgroup 0 alone, the others return at once.  From SOLVE_COOP_MIN gathered values on, every sweep of the
// iteration is split over the workgroups: each sums its slice, publishes (hi, lo, count) in its slot of a double-buffered
// scratch inside ProjScalars, arrives at a counter and waits for the others (agent-scope release / acquire; the spin is
// bounded and leaves through an abort flag), then ALL of them add the SOLVE_G partials in slot order and take the same step.
// One workgroup streaming 2 M magnitudes per sweep took up to 1 ms at 512^3; split 32 ways the sweep is bandwidth-trivial,
// which is what lets the bracket be gathered earlier instead of being refined by one more full pass over v (L1 cap).
// The workgroup that finishes last writes theta and prepares the next call (nobody may reset the state while another
// workgroup has yet to read it).
constexpr int SOLVE_G = 32;
constexpr long long SOLVE_COOP_MIN = 1ll << 17;      // (documentation of the default; see solve_coop_min())
static_assert(SOLVE_G <= SIPX_SOLVE_SLOTS, "ProjScalars holds SIPX_SOLVE_SLOTS cooperative slots");

// (the body as a device function: k_l1_solve runs it on its own grid, k_spec_finish -- one workgroup per set -- behind the
//  decision and the unpacking of a slab-decomposed search; G workgroups take part, this one is number wg)
template <typename T>
__device__ void l1_solve_body(ProjScalars<T>* ps, T radius, const T* __restrict__ compact, const double* __restrict__ partials,
                              long long true_len, double hw_max, int lean_on, int* host_want, long long coop_min, int only_if_settled,
                              const int G, const int wg) {
  constexpr int NT = SIPX_SOLVE_NT;
  // (slab-decomposed, speculative exchange: queued before the host knows whether the search needs its fallback sweeps --
  //  then this launch is not the one that solves it)
  if (only_if_settled && ps->need && !ps->spec_ok) return;
  __shared__ double ssum[NT / 64];
  __shared__ double ssum_lo[NT / 64];
  __shared__ double scnt[NT / 64];
  __shared__ double sh_theta, sh_sa, sh_ca;
  __shared__ int sh_done;
  const int need = ps->need;
  const long long n_all = need ? (long long)ps->n_compact : 0;
  const bool coop = G > 1 && n_all >= coop_min;
  if (!coop && wg != 0) return;
  double theta = 0;
  int iters_done = 0;
  if (need) {
    if (ps->spec_ok) {
      if (threadIdx.x == 0) { sh_sa = ps->s_above; sh_ca = ps->c_above; }
    } else {   // (S,C) above the bracket: block partials of the fallback compaction pass
      double s = 0, c = 0;
      for (int i = threadIdx.x; i < NB; i += NT) {
        s += partials[(long long)SL_ABOVE_S * NB + i];
        c += partials[(long long)SL_ABOVE_C * NB + i];
      }
      s = wave_sum(s);
      c = wave_sum(c);
      if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; scnt[threadIdx.x >> 6] = c; }
      __syncthreads();
      if (threadIdx.x == 0) {
        double S = 0, Cc = 0;
        for (int i = 0; i < NT / 64; ++i) { S += ssum[i]; Cc += scnt[i]; }
        sh_sa = S; sh_ca = Cc;
      }
    }
    __syncthreads();
    // this workgroup's slice of the gathered values (all of them unless the sweeps are shared), in vectors of four
    const long long nv_all = n_all / 4;
    const long long per = coop ? (nv_all + G - 1) / G : nv_all;
    const long long v0 = coop ? (long long)wg * per : 0, v1 = coop ? (v0 + per < nv_all ? v0 + per : nv_all) : nv_all;
    const bool tail_owner = !coop || wg == G - 1;
    const double sa = sh_sa, ca = sh_ca, b = (double)radius;
    theta = ps->lo;
    double cprev = -1;
    for (int it = 0; it < 200; ++it) {
      DD s = {0.0, 0.0};
      double c = 0;
      {   // 16-byte loads, four in flight per thread
        const long long nv = v1;
        for (long long i0 = v0 + threadIdx.x; i0 < nv; i0 += 4 * NT) {
          Vec<T, 4> q[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const long long i = i0 + (long long)u * NT;
            if (i < nv) q[u] = ldv_u<T, 4>(compact + 4 * i);
            else { q[u].v[0] = q[u].v[1] = q[u].v[2] = q[u].v[3] = T(0); }     // magnitudes are >= 0 = not above theta >= 0
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const double av = (double)q[u].v[k];
              if (av > theta) { s = dd_add(s, av); c += 1.0; }
            }
        }
        if (tail_owner)
          for (long long e = 4 * nv_all + threadIdx.x; e < n_all; e += NT) {
            const double av = (double)compact[e];
            if (av > theta) { s = dd_add(s, av); c += 1.0; }
          }
      }
      s = wave_sum_dd(s);
      c = wave_sum(c);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s.hi; ssum_lo[threadIdx.x >> 6] = s.lo; scnt[threadIdx.x >> 6] = c; }
      __syncthreads();
      if (threadIdx.x == 0) {
        DD Sd = {0.0, 0.0};
        double Cc = 0;
        for (int i = 0; i < NT / 64; ++i) { Sd = dd_add(Sd, DD{ssum[i], ssum_lo[i]}); Cc += scnt[i]; }
        if (coop) {        // publish this workgroup's share, wait for the others, add all shares in slot order
          const int buf = it & 1;
          __hip_atomic_store(&ps->coop_hi[buf][wg], Sd.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&ps->coop_lo[buf][wg], Sd.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&ps->coop_c[buf][wg], Cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(&ps->coop_arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned want = (unsigned)G * (unsigned)(it + 1);
          unsigned spins = 0;
          while (__hip_atomic_load(&ps->coop_arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (__hip_atomic_load(&ps->coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (++spins > 8000000u) {      // several seconds: something is wrong -- leave, flagged, rather than hang
              __hip_atomic_store(&ps->coop_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
            __builtin_amdgcn_s_sleep(4);
          }
          Sd = DD{0.0, 0.0};
          Cc = 0;
          for (int k = 0; k < G; ++k) {
            Sd = dd_add(Sd, DD{__hip_atomic_load(&ps->coop_hi[buf][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __hip_atomic_load(&ps->coop_lo[buf][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)});
            Cc += __hip_atomic_load(&ps->coop_c[buf][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        const double S = Sd.hi + Sd.lo;
        const double tot = ca + Cc;
        double tn = theta;
        if (tot > 0) tn = (sa + S - b) / tot;
        sh_done = (Cc == cprev || !(tot > 0)) ? 1 : 0;
        if (coop && __hip_atomic_load(&ps->coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) sh_done = 1;
        sh_theta = tn > theta ? tn : theta;
        sh_sa = sa; sh_ca = tot;                     // sh_ca: size of the active set at the last evaluated theta
        scnt[0] = Cc;
      }
      __syncthreads();
      theta = sh_theta;
      cprev = scnt[0];
      iters_done = it + 1;
      if (sh_done) break;
    }
  }
  if (threadIdx.x == 0) {
    if (coop) {     // only the workgroup that finishes last may write the result and reset the state
      const unsigned t = __hip_atomic_fetch_add(&ps->coop_finish, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t != (unsigned)G - 1u) return;
      ps->dbg[3] = ps->coop_abort ? -1.0 : 0.0;
      ps->coop_arrive = 0;
      ps->coop_finish = 0;
      ps->coop_abort = 0;
    }
    if (need) {
      // The reference's scan `while u[rho+1] > (sv[rho+1]-b)/(rho+1) && rho+1 < lv` (project_l1_Duchi!.jl:42) never lets
      // the active set reach the whole vector: when every entry would stay active it stops at lv-1 and thresholds with
      // theta = (||v||_1 - min|v| - b) / (lv - 1).  Replicated (all entries active => none is zero => the smallest
      // non-zero magnitude of the first pass is min|v|).
      if (sh_ca >= (double)true_len && true_len > 1) theta = (ps->asum - (double)ps->vmin - (double)radius) / (double)(true_len - 1);
      const T th = (T)theta;
      ps->theta = th > T(0) ? th : T(0);       // theta = max(0, .)   project_l1_Duchi!.jl:46
      if (ps->gather_overflow) ps->theta = (T)NAN;     // slab-decomposed: a rank's share did not fit its exchange segment
    }
    ps->dbg[0] = (double)ps->n_compact; ps->dbg[1] = ps->spec_overflow; ps->dbg[2] = ps->spec_ok;
    ps->dbg[3] = (coop && ps->dbg[3] < 0) ? -1.0 : (double)iters_done;       // -1: a cooperative sweep was abandoned
    // ---- state for the next call ----
    // a search that the speculative gather settled is followed by a LEAN first pass (two probes instead of eight)
    ps->lean = (lean_on && need && theta > 0 && ps->spec_ok && !ps->spec_overflow && sh_ca < (double)true_len) ? 1 : 0;
    ps->want_sample = 0;
    if (need && theta > 0) {
      double hw = ps->hw;
      if (!(ps->theta_prev > 0)) ps->want_sample = 1;      // the first theta of this set: nothing is known about how it moves
      if (ps->theta_prev > 0) {
        const double d = fabs(theta / ps->theta_prev - 1.0);
        // theta moved by more than a third of the widest speculative range: the coming prediction is not to be trusted -- a
        // sampled estimate first (k_sample), when the caller provides for it.  A search that needed its fallback sweeps although
        // theta hardly moved asks for NO sample: the error of a sampled estimate is absolute (rms excess / sqrt(active sample),
        // and the same from one iteration to the next because the sample is the same subset), so against a small theta it is
        // percent while theta itself moves by hundredths of a percent -- there the previous theta is the better prediction
        // (512^3, default window: searches of the D_x set missed their range on every iteration from 18 on, sampled each time).
        ps->want_sample = (3.0 * d > hw_max) ? 1 : 0;
        // this search followed a change of rho: was theta_prev * rho_old / rho_new (k_ps_rescale) good to the range it gets?
        // Early on it is not (theta is set by x_hat, not by l / rho) and the sampled estimate is; late it is, and then more
        // accurate than a sample, whose error grows as theta shrinks against the spread of the values
        if (ps->rescaled) ps->resc_bad = d > hw_max ? 1 : 0;
        hw = 3.0 * d;                                  // theta moves slowly while rho, gamma stay put
        // The floor of the half-width follows what the range GATHERS, not a fixed relative width: theta wanders by +-0.1 ... 0.2 %
        // from one iteration to the next long after rho and gamma have settled (256^3, iterations 22 and 26 of the headline run:
        // -0.19 % and +0.11 % against a range of +-0.1 %: two searches each fell back to their refinement + compaction sweeps,
        // ~100 us apiece on the critical path), while a range of +-0.1 % holds a few thousand magnitudes -- a fraction of what the
        // solve takes in its stride.  So: as wide as gathers about max(2^15, len / 1024) magnitudes (density from this search's own
        // count when the speculative range was what it gathered), between 0.1 % and 0.4 %.
        double hw_floor = 2e-3;
        if (ps->spec_ok && n_all > 0 && ps->hw > 0) {
          const double tgt = fmax(32768.0, (double)true_len / 1024.0);
          hw_floor = ps->hw * tgt / (double)n_all;
          hw_floor = hw_floor < 1e-3 ? 1e-3 : (hw_floor > 4e-3 ? 4e-3 : hw_floor);
        }
        hw = hw < hw_floor ? hw_floor : (hw > hw_max ? hw_max : hw);
        if (ps->spec_overflow) hw = ps->hw * 0.5;      // the last range gathered too much
      }
      ps->hw = hw;
      ps->theta_prev = theta;
      for (int k = 0; k < L1_K; ++k) ps->t[k] = (double)(T)(theta * (1.0 + hw * l1_probe_mult(k)));
      ps->spec_lo = ps->t[L1_WIN_LO];
      ps->spec_hi = ps->t[L1_WIN_HI];
    } else if (ps->theta_prev > 0) {            // inside the ball now: keep probing around the last theta
      for (int k = 0; k < L1_K; ++k) ps->t[k] = (double)(T)(ps->theta_prev * (1.0 + ps->hw * l1_probe_mult(k)));
      ps->spec_lo = ps->t[L1_WIN_LO];
      ps->spec_hi = ps->t[L1_WIN_HI];
    }
    ps->n_compact = 0;
    ps->spec_overflow = 0;
    if (ps->sampled && need && theta > 0) { ps->samp_bias = ps->samp_theta - theta; ps->samp_bias_ok = 1; }
    ps->dbg_sampled = ps->sampled;
    ps->sampled = 0;
    ps->rescaled = 0;
    // (bit 16: the coming first pass will be a lean one -- nothing between here and that pass can take the flag back, so a caller
    //  that reads the word may leave the full first pass unlaunched)
    if (host_want)
      __hip_atomic_store(host_want, ps->want_sample | (ps->rounds_used << 8) | ((ps->lean && ps->spec_hi > ps->spec_lo) ? (1 << 16) : 0), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <typename T>
__global__ __launch_bounds__(SIPX_SOLVE_NT) void k_l1_solve(ProjScalars<T>* ps, T radius, const T* __restrict__ compact,
                                                   const double* __restrict__ partials, long long true_len, double hw_max,
                                                   int lean_on, int* host_want, long long coop_min, int only_if_settled) {
  l1_solve_body<T>(ps, radius, compact, partials, true_len, hw_max, lean_on, host_want, coop_min, only_if_settled, (int)gridDim.x,
                   (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Cardinality: tau = k-th largest magnitude.  Bracket (lo, hi] with C(lo) >= k > C(hi) from the probe counts,
// refined by gated probe passes until it holds at most CARD_CAP magnitudes (or the passes run out).
constexpr double CARD_CAP = 8192.0;
constexpr int CARD_REFINES = 5;

// SLAB 0: one rank -- sums of the pass's partial slots, decision.  Slab-decomposed grid (round 5): SLAB 1 leaves the sums of THIS
// rank's planes in ps->red (and its largest magnitude in its own entry of ps->mm, zeros in the others) for ONE all-reduce, SLAB 2
// decides from the all-reduced values -- counts are sums of exact integers, the largest magnitude a maximum: every rank takes the
// decisions a single rank would take, bit for bit.
template <typename T, int STAGE, int SLAB = 0>
__global__ __launch_bounds__(1024) void k_card_decide(const double* __restrict__ partials,
                                                      const T* __restrict__ maxpart, ProjScalars<T>* ps, long long k,
                                                      long long true_len, int world = 1, int rank = 0, double cap_max = CARD_CAP) {
  if (STAGE == 1 && !(ps->need && ps->refine)) return;        // (the same verdict on every rank)
  __shared__ double red[PREP_SLOTS];
  __shared__ T smax[16];
  T vmax = T(0);
  if (SLAB != 2) {
    reduce_slots<1024>(partials, red);
    if (STAGE == 0) {
      for (int i = threadIdx.x; i < NB; i += 1024) vmax = maxpart[i] > vmax ? maxpart[i] : vmax;
      vmax = wave_max<T>(vmax);
      if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = vmax;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (SLAB != 2 && STAGE == 0)
    for (int i = 0; i < 16; ++i) vmax = smax[i] > vmax ? smax[i] : vmax;
  if (SLAB == 1) {
    for (int i = 0; i < PREP_SLOTS; ++i) ps->red[i] = red[i];
    ps->ovf = 0;
    for (int r = 0; r < 2 * world; ++r) ps->mm[r] = 0;
    if (STAGE == 0) ps->mm[2 * rank] = (double)vmax;
    return;
  }
  if (SLAB == 2) {
    for (int i = 0; i < PREP_SLOTS; ++i) red[i] = ps->red[i];
    for (int r = 0; r < world; ++r) vmax = (T)ps->mm[2 * r] > vmax ? (T)ps->mm[2 * r] : vmax;
  }
  if (STAGE == 0) {
    ps->vmax = vmax;
    ps->asum = red[0];
    ps->need = 0;
    ps->refine = 0;
    ps->spec_ok = 0;
    ps->n_compact = 0;
    ps->quota = 0x7fffffffffffffffll;
    const double nnz = red[2];
    if (k >= true_len || (double)k >= nnz) { ps->tau = T(0); return; }   // everything non-zero survives
    if (k <= 0) { ps->tau = (T)INFINITY; return; }                       // nothing survives
    ps->need = 1;
    ps->lo = 0; ps->c_lo = nnz;
    ps->hi = (double)vmax; ps->c_hi = 0;
  }
  double lo = ps->lo, hi = ps->hi, clo = ps->c_lo, chi = ps->c_hi;
  for (int j = 0; j < L1_K; ++j) {
    const double t = ps->t[j];
    if (!(t < INFINITY) || !(t > lo) || !(t < hi)) continue;
    const double C = red[3 + L1_K + j];
    if (C >= (double)k) { lo = t; clo = C; }
    else if (t < hi) { hi = t; chi = C; }
  }
  // second sweep: a probe accepted as `lo` early may lie above one accepted as `hi` later -- cannot happen,
  // C is non-increasing in t, so the accepted lo's are all below the accepted hi's.
  ps->lo = lo; ps->hi = hi; ps->c_lo = clo; ps->c_hi = chi;
  // (slab-decomposed: what the bracket holds over ALL ranks has to fit one rank's exchange segment -- all of it may sit on one rank)
  if (clo - chi > cap_max && hi > lo) {
    ps->refine = 1;
    for (int j = 0; j < L1_K; ++j) ps->t[j] = (double)(T)(lo + (hi - lo) * (double)(j + 1) / (double)(L1_K + 1));
  } else {
    ps->refine = 0;
  }
}

// Exact selection among the gathered (magnitude, index) pairs: tau = the (k - C(hi))-th largest of them
// (binary search on the bit pattern, which is monotone for non-negative floats), then the index cut among
// the entries equal to tau (stable sortperm: lowest indices win, project_cardinality!.jl:18-19).
template <typename T>
__global__ __launch_bounds__(1024) void k_card_select(ProjScalars<T>* ps, long long k, const T* __restrict__ compact) {
  typedef typename std::conditional<sizeof(T) == 4, unsigned int, unsigned long long>::type U;
  __shared__ double scnt[16];
  __shared__ double sh_val;
  auto count_block = [&](auto pred) -> double {     // number of gathered entries satisfying pred (block wide)
    const long long n = (long long)ps->n_compact;
    double c = 0;
    for (long long e = threadIdx.x; e < n; e += 1024) c += pred(e) ? 1.0 : 0.0;
    c = wave_sum(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int i = 0; i < 16; ++i) s += scnt[i]; sh_val = s; }
    __syncthreads();
    return sh_val;
  };
  if (ps->need) {
    const double kk = (double)k - ps->c_hi;          // rank inside the bracket, >= 1
    const U* keys = reinterpret_cast<const U*>(compact);
    U ans = 0;
    for (int bit = (int)sizeof(U) * 8 - 2; bit >= 0; --bit) {     // sign bit is 0
      const U cand = ans | ((U)1 << bit);
      const double c = count_block([&](long long e) { return keys[e] >= cand; });
      if (c >= kk) ans = cand;
    }
    const double c_gt = count_block([&](long long e) { return keys[e] > ans; });
    const double c_eq = count_block([&](long long e) { return keys[e] == ans; });
    const double quota = kk - c_gt;                  // how many entries equal to tau survive (>= 1)
    long long cut = 0x7fffffffffffffffll;
    if (c_eq > quota) {                              // partial tie: smallest index I with #{eq, idx <= I} >= quota
      const long long* idx = ps->cidx;
      long long lo = -1, hi = 0x3fffffffffffffffll;
      while (hi - lo > 1) {
        const long long mid = lo + (hi - lo) / 2;
        const double c = count_block([&](long long e) { return keys[e] == ans && idx[e] <= mid; });
        if (c >= quota) hi = mid; else lo = mid;
      }
      cut = hi;
    }
    if (threadIdx.x == 0) {
      T tau;
      memcpy(&tau, &ans, sizeof(T));
      ps->tau = tau;
      ps->quota = cut;
      ps->dbg[0] = (double)ps->n_compact;
    }
  }
  if (threadIdx.x == 0) {
    const T tp = ps->need ? ps->tau : ps->tau_prev;
    if (tp > T(0) && tp < (T)INFINITY) {
      ps->tau_prev = tp;
      const double m[L1_K] = {0.5, 0.9, 0.99, 0.999, 1.001, 1.01, 1.1, 2.0};
      for (int j = 0; j < L1_K; ++j) ps->t[j] = (double)(T)((double)tp * m[j]);
    }
    ps->n_compact = 0;
  }
}

// Slab-decomposed cardinality search: every rank has gathered the (magnitude, padded index) pairs of ITS planes inside the final
// bracket.  k_card_pack puts them into the rank's segment of the exchange buffer -- a count (-1: more than the segment holds),
// the indices, the magnitudes -- and after the all-gather k_card_unpack strings the segments together again, in rank order; the
// selection that follows (k_card_select: binary search on bit patterns, index cut among the ties) does not depend on the order
// of the pairs, so every rank arrives at the tau and the index cut of a single rank.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_card_pack(const ProjScalars<T>* ps, const T* __restrict__ compact, char* __restrict__ seg,
                                                     long long cap) {
  const long long n = ps->need ? (long long)ps->n_compact : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<double*>(seg)[0] = n <= cap ? (double)n : -1.0;
  if (n > cap) return;
  long long* si = reinterpret_cast<long long*>(seg + 16);
  T* sv = reinterpret_cast<T*>(seg + 16 + cap * 8);
  const long long* idx = ps->cidx;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) { si[i] = idx[i]; sv[i] = compact[i]; }
}
template <typename T>
__global__ __launch_bounds__(1024) void k_card_unpack(ProjScalars<T>* ps, T* __restrict__ compact, const char* __restrict__ seg0,
                                                      long long seg_bytes, int world, long long cap, int* host_ovf) {
  if (!ps->need) return;
  long long off = 0;
  bool ovf = false;
  for (int r = 0; r < world; ++r) ovf |= reinterpret_cast<const double*>(seg0 + (long long)r * seg_bytes)[0] < 0;
  if (ovf) {                                     // (every rank reads the same headers: the same verdict everywhere)
    if (threadIdx.x == 0) {
      ps->gather_overflow = 1;
      ps->n_compact = 0;
      if (host_ovf) __hip_atomic_store(host_ovf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  long long* idx = ps->cidx;
  for (int r = 0; r < world; ++r) {
    const char* seg = seg0 + (long long)r * seg_bytes;
    const long long n = (long long)reinterpret_cast<const double*>(seg)[0];
    const long long* si = reinterpret_cast<const long long*>(seg + 16);
    const T* sv = reinterpret_cast<const T*>(seg + 16 + cap * 8);
    for (long long i = threadIdx.x; i < n; i += 1024) { idx[off + i] = si[i]; compact[off + i] = sv[i]; }
    off += n;
  }
  __syncthreads();
  if (threadIdx.x == 0) ps->n_compact = (unsigned long long)off;
}

// ---------------------------------------------------------------------------------------------
// Slab-decomposed grid: every rank has gathered the magnitudes of ITS planes that fall into the bracket.  k_gather_pack puts
// them, behind a header (count, and the rank's (S, C) above the bracket from the fallback compaction), into the rank's
// segment of the exchange buffer; after the all-gather k_gather_unpack strings the segments together in `compact` again, in
// rank order, adds up the headers and leaves everything as k_l1_solve expects it from a single rank -- every rank then solves
// the same problem and arrives at the same theta, bit for bit (the sum of the gathered values is order-independent).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gather_pack(ProjScalars<T>* ps, const T* __restrict__ compact,
                                                       const double* __restrict__ partials, T* __restrict__ seg, long long gcap) {
  const bool active = ps->need != 0;
  const long long n = active ? (long long)ps->n_compact : 0;
  if (blockIdx.x == 0) {
    double sa = 0, ca = 0;
    if (active && !ps->spec_ok) {               // block partials of the compaction pass of this rank
      sa = block_sum_partials(partials + (long long)SL_ABOVE_S * NB);
      ca = block_sum_partials(partials + (long long)SL_ABOVE_C * NB);
    }
    if (threadIdx.x == 0) {
      double* h = reinterpret_cast<double*>(seg);
      h[0] = n <= gcap ? (double)n : -1.0;      // -1: more than the segment holds
      h[1] = sa;
      h[2] = ca;
    }
  }
  const long long m = n <= gcap ? n : 0;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < m; i += (long long)gridDim.x * BLOCK) seg[GATHER_HDR + i] = compact[i];
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gather_unpack(ProjScalars<T>* ps, T* __restrict__ compact, double* __restrict__ partials,
                                                         const T* __restrict__ gseg0, long long chunk, int world, long long compact_len,
                                                         int* host_ovf) {
  // gseg0: this set's segment in rank 0's chunk of the exchange buffer; rank r's is `chunk` elements further per rank
  long long off = 0;
  double sa = 0, ca = 0;
  bool bad = false;
  for (int r = 0; r < world; ++r) {
    const double* h = reinterpret_cast<const double*>(gseg0 + (long long)r * chunk);
    const long long n = (long long)h[0];
    if (n < 0 || off + n > compact_len) { bad = true; break; }
    sa += h[1];
    ca += h[2];
    const T* v = gseg0 + (long long)r * chunk + GATHER_HDR;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) compact[off + i] = v[i];
    off += n;
  }
  if (blockIdx.x == 0) {
    // (S, C) above the bracket, as the block partials k_l1_solve adds up: the totals in entry 0, zeros behind
    for (int i = threadIdx.x; i < NB; i += BLOCK) {
      partials[(long long)SL_ABOVE_S * NB + i] = i == 0 ? sa : 0.0;
      partials[(long long)SL_ABOVE_C * NB + i] = i == 0 ? ca : 0.0;
    }
    if (threadIdx.x == 0) {
      ps->n_compact = bad ? 0ull : (unsigned long long)off;
      // every rank reads the same headers, so every rank takes the same verdict: theta = NaN for this search (k_l1_solve) and
      // the host's pinned word raised -- the engine returns an error from the y/l update on every rank alike
      ps->gather_overflow = bad ? 1 : 0;
      if (bad && host_ovf) __hip_atomic_store(host_ovf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Slab-decomposed grid, SPECULATIVE EXCHANGE: one all-gather instead of (all-reduce, all-reduce, ..., all-gather) whenever the
// speculative range of the first pass holds theta -- which is the rule once rho and gamma have settled.  After its first pass a
// rank knows its share of the probe sums and has gathered its magnitudes inside the speculative range; k_spec_pack puts both
// into the rank's FAST segment (header: the PREP_SLOTS sums, overflow flag, largest / smallest non-zero magnitude, count; then
// the values, at most `cap` of them).  After the all-gather every rank adds the headers up in rank order (k_spec_decide: the
// same numbers in the same order on every rank, so the same decision, bit for bit), takes the first-pass decision on the sums
// and -- when the range held theta -- strings the gathered values together (k_spec_unpack) and solves.  Otherwise (theta left
// the range, the LDS buffers or a fast segment overflowed) the search falls back to its refinement rounds and the full-size
// exchange; the host learns which from one pinned word per set and enqueues the collectives of the fallback only then.
constexpr int FH_OVF = PREP_SLOTS, FH_MAX = PREP_SLOTS + 1, FH_MIN = PREP_SLOTS + 2, FH_CNT = PREP_SLOTS + 3;

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_spec_pack(const ProjScalars<T>* ps, const double* __restrict__ reg, int rank, int is_l1,
                                                     const T* __restrict__ compact, T* __restrict__ seg, long long cap) {
  // reg: what k_slot_sums<0> left for this rank -- sums | ovf | (max, min) in the rank's own two entries
  const long long n = is_l1 ? (long long)ps->n_compact : 0;
  if (blockIdx.x == 0 && threadIdx.x < PREP_SLOTS + 4) {
    double* h = reinterpret_cast<double*>(seg);
    const int i = threadIdx.x;
    double v;
    if (i <= FH_OVF) v = reg[i];
    else if (i == FH_MAX) v = reg[PREP_SLOTS + 1 + 2 * rank];
    else if (i == FH_MIN) v = reg[PREP_SLOTS + 1 + 2 * rank + 1];
    else v = n <= cap ? (double)n : -1.0;                  // -1: more than the fast segment holds
    h[i] = v;
  }
  const long long m = n <= cap ? n : 0;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < m; i += (long long)gridDim.x * BLOCK) seg[fast_hdr<T>() + i] = compact[i];
}

template <typename T>
__global__ __launch_bounds__(64) void k_spec_decide(ProjScalars<T>* ps, DecideArgs da, int world, double* __restrict__ reg,
                                                    const T* __restrict__ fseg0, long long fchunk, unsigned seq, unsigned* verdict) {
  // reg (the set's region of the staging buffer) receives the summed header: the stages of the fallback read it there
  __shared__ double sreg[PREP_SLOTS + 1 + 2 * SIPX_MAX_WORLD];
  __shared__ double scount;
  const int i = threadIdx.x;
  if (i <= FH_OVF) {
    double v = 0;
    for (int r = 0; r < world; ++r) v += reinterpret_cast<const double*>(fseg0 + (long long)r * fchunk)[i];
    sreg[i] = v;
  }
  if (i == FH_CNT) {
    double tot = 0;
    bool bad = false;
    for (int r = 0; r < world; ++r) {
      const double* h = reinterpret_cast<const double*>(fseg0 + (long long)r * fchunk);
      sreg[PREP_SLOTS + 1 + 2 * r] = h[FH_MAX];
      sreg[PREP_SLOTS + 1 + 2 * r + 1] = h[FH_MIN];
      if (h[FH_CNT] < 0) bad = true; else tot += h[FH_CNT];
    }
    scount = bad ? -1.0 : tot;
  }
  __syncthreads();
  if (i == 0 && scount < 0) sreg[FH_OVF] += 1.0;           // a fast segment overflowed: as if the speculation had
  __syncthreads();
  if (i < PREP_SLOTS + 1 + 2 * world) reg[i] = sreg[i];
  if (i != 0) return;
  decide_body<T, 0>(ps, da.prox, (T)da.pmin, (T)da.pmax, da.true_len, da.nospec, da.capdiv, world, da.cap_max, sreg);
  const bool settled = !(da.prox == PX_L1 && ps->need && !ps->spec_ok);
  if (da.prox == PX_L1 && ps->need && ps->spec_ok) ps->n_compact = (unsigned long long)scount;     // what k_spec_unpack strings together
  ps->gather_overflow = 0;
  // bit 0: the fallback has to run; bit 1: it starts with refinement rounds (else the bracket is gathered at once)
  const unsigned word = (seq << 2) | (settled ? 0u : 1u) | ((!settled && ps->refine) ? 2u : 0u);
  __hip_atomic_store(verdict, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_spec_unpack(const ProjScalars<T>* ps, T* __restrict__ compact, const T* __restrict__ fseg0,
                                                       long long fchunk, int world) {
  if (!(ps->need && ps->spec_ok)) return;
  long long off = 0;
  for (int r = 0; r < world; ++r) {
    const T* seg = fseg0 + (long long)r * fchunk;
    const long long n = (long long)reinterpret_cast<const double*>(seg)[FH_CNT];
    const T* v = seg + fast_hdr<T>();
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) compact[off + i] = v[i];
    off += n;
  }
}

// The same two steps for ALL sets of a slab-decomposed iteration in two launches (a rank's share of the grid is small when the
// ranks are many, and the iteration is then bound by the number of launches the host can issue: twenty-one small kernels of
// three searches become two): k_spec_sums_pack = k_slot_sums<0> + k_spec_pack of every set (grid.y = set; a workgroup per
// partial slot writes its sum straight into the header of the rank's fast segment, one takes the extrema, the overflow flag and
// the count, the others copy the gathered magnitudes); k_spec_finish = k_spec_decide + k_spec_unpack + the solve, one workgroup
// per set.
constexpr int SPEC_COPY_WG = 12;
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_spec_sums_pack(SpecPackArgs<T> A) {
  const SpecPackSet<T>& S = A.s[blockIdx.y];
  double* h = reinterpret_cast<double*>(S.seg);
  const int b = blockIdx.x;
  if (b < PREP_SLOTS) {
    const double v = block_sum_partials(S.partials + (long long)b * NB);
    if (threadIdx.x == 0) h[b] = v;
    return;
  }
  const long long n = S.is_l1 ? (long long)S.ps->n_compact : 0;
  if (b == PREP_SLOTS) {
    __shared__ T smax[BLOCK / 64], smin[BLOCK / 64];
    T vmax = T(0), vmin = (T)INFINITY;
    for (int i = threadIdx.x; i < NB; i += BLOCK) {
      vmax = S.maxpart[i] > vmax ? S.maxpart[i] : vmax;
      const T mn = S.maxpart[NB + i];               // 0 = entry beyond the pass's grid, or a workgroup that saw no non-zero magnitude
      vmin = (mn > T(0) && mn < vmin) ? mn : vmin;
    }
    vmax = wave_max<T>(vmax);
    vmin = -wave_max<T>(-vmin);
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = vmax; smin[threadIdx.x >> 6] = vmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < BLOCK / 64; ++i) { vmax = smax[i] > vmax ? smax[i] : vmax; vmin = smin[i] < vmin ? smin[i] : vmin; }
      h[FH_OVF] = S.ps->spec_overflow ? 1.0 : 0.0;
      h[FH_MAX] = (double)vmax;
      h[FH_MIN] = (vmin < (T)INFINITY) ? (double)vmin : 0.0;          // 0 = this rank saw no non-zero magnitude
      h[FH_CNT] = (A.local || n <= A.cap) ? (double)n : -1.0;         // -1: more than the fast segment holds
    }
    return;
  }
  if (A.local) return;
  const long long m = n <= A.cap ? n : 0;
  for (long long i = (long long)(b - PREP_SLOTS - 1) * BLOCK + threadIdx.x; i < m; i += (long long)SPEC_COPY_WG * BLOCK)
    S.seg[fast_hdr<T>() + i] = S.compact[i];
}
template <typename T>
void K<T>::spec_sums_pack(hipStream_t s, const SpecPackArgs<T>& A) {
  if (A.nsets < 1 || A.nsets > SPEC_MAX_SETS) throw std::runtime_error("spec_sums_pack: set count out of range");
  ObsScope obs_(KID_SLOT_SUMS, s, 0.0);
  hipLaunchKernelGGL((k_spec_sums_pack<T>), dim3(PREP_SLOTS + 1 + (A.local ? 0 : SPEC_COPY_WG), A.nsets), dim3(BLOCK), 0, s, A);
  SIPX_HIP(hipGetLastError());
}

template <typename T>
__global__ __launch_bounds__(SIPX_SOLVE_NT) void k_spec_finish(SpecFinishArgs<T> A) {
  const SpecFinishSet<T>& S = A.s[blockIdx.x];
  ProjScalars<T>* ps = S.ps;
  __shared__ double sreg[PREP_SLOTS + 1 + 2 * SIPX_MAX_WORLD];
  __shared__ double scount;
  const int i = threadIdx.x, world = A.world;
  if (i <= FH_OVF) {
    double v = 0;
    for (int r = 0; r < world; ++r) v += reinterpret_cast<const double*>(S.fseg0 + (long long)r * A.fchunk)[i];
    sreg[i] = v;
  }
  if (i == FH_CNT) {
    double tot = 0;
    bool bad = false;
    for (int r = 0; r < world; ++r) {
      const double* h = reinterpret_cast<const double*>(S.fseg0 + (long long)r * A.fchunk);
      sreg[PREP_SLOTS + 1 + 2 * r] = h[FH_MAX];
      sreg[PREP_SLOTS + 1 + 2 * r + 1] = h[FH_MIN];
      if (h[FH_CNT] < 0) bad = true; else tot += h[FH_CNT];
    }
    scount = bad ? -1.0 : tot;
  }
  __syncthreads();
  if (i == 0 && scount < 0) sreg[FH_OVF] += 1.0;           // a fast segment overflowed: as if the speculation had
  __syncthreads();
  if (i < PREP_SLOTS + 1 + 2 * world) S.reg[i] = sreg[i];
  if (i == 0) {
    decide_body<T, 0>(ps, S.da.prox, (T)S.da.pmin, (T)S.da.pmax, S.da.true_len, S.da.nospec, S.da.capdiv, world, S.da.cap_max, sreg);
    const bool settled = !(S.da.prox == PX_L1 && ps->need && !ps->spec_ok);
    if (S.da.prox == PX_L1 && ps->need && ps->spec_ok) ps->n_compact = (unsigned long long)scount;
    ps->gather_overflow = 0;
    const unsigned word = (A.seq << 2) | (settled ? 0u : 1u) | ((!settled && ps->refine) ? 2u : 0u);
    __hip_atomic_store(S.verdict, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_block();
  __syncthreads();
  if (S.da.prox != PX_L1) return;
  if (ps->need && ps->spec_ok && !A.local) {                // the values every rank gathered inside the range, strung together in rank order
    long long off = 0;
    for (int r = 0; r < world; ++r) {
      const T* seg = S.fseg0 + (long long)r * A.fchunk;
      const long long n = (long long)reinterpret_cast<const double*>(seg)[FH_CNT];
      const T* v = seg + fast_hdr<T>();
      for (long long k = threadIdx.x; k < n; k += SIPX_SOLVE_NT) S.compact[off + k] = v[k];
      off += n;
    }
  }
  __threadfence_block();
  __syncthreads();
  l1_solve_body<T>(ps, S.radius, S.compact, S.partials, S.da.true_len, A.hw_max, A.lean_on, S.host_want, A.coop_min, 1, 1, 0);
}
template <typename T>
void K<T>::spec_finish(hipStream_t s, SpecFinishArgs<T>& A) {
  if (A.nsets < 1 || A.nsets > SPEC_MAX_SETS) throw std::runtime_error("spec_finish: set count out of range");
  static const double capdiv = [] { const char* e = getenv("SIPX_L1_CAPDIV"); return e ? atof(e) : 64.0; }();
  for (int j = 0; j < A.nsets; ++j) A.s[j].da.capdiv = capdiv;
  A.hw_max = l1_hw_max();
  A.lean_on = l1_lean_on();
  A.coop_min = solve_coop_min();
  ObsScope obs_(KID_L1_SOLVE, s, 0.0);
  hipLaunchKernelGGL((k_spec_finish<T>), dim3(A.nsets), dim3(SIPX_SOLVE_NT), 0, s, A);
  SIPX_HIP(hipGetLastError());
}

// v = x_hat - l/rho: where the multiplier term dominates, theta moves like 1/rho when rho is changed.  Re-centre the
// probes of the coming call on the scaled prediction (and widen the range: the prediction is good to a few percent).
template <typename T>
__global__ void k_ps_rescale(ProjScalars<T>* ps, double factor, double hw_max) {
  if (!(ps->theta_prev > 0)) return;
  ps->theta_prev *= factor;
  ps->hw = hw_max;
  ps->rescaled = 1;
  if (ps->resc_bad) ps->want_sample = 1;   // the last such prediction missed the range: sample
  for (int k = 0; k < L1_K; ++k) ps->t[k] = (double)(T)(ps->theta_prev * (1.0 + ps->hw * l1_probe_mult(k)));
  ps->spec_lo = ps->t[L1_WIN_LO];
  ps->spec_hi = ps->t[L1_WIN_HI];
}
template <typename T>
void K<T>::ps_rescale(hipStream_t s, ProjScalars<T>* ps, double factor) {
  ObsScope obs_(KID_PS_RESCALE, s, 0.0);
  hipLaunchKernelGGL((k_ps_rescale<T>), dim3(1), dim3(1), 0, s, ps, factor, l1_hw_max());
  SIPX_HIP(hipGetLastError());
}

template <typename T>
__global__ void k_ps_rescale_multi(RescaleMulti<T> A, double hw_max) {
  ProjScalars<T>* ps = A.ps[blockIdx.x];
  const double factor = A.factor[blockIdx.x];
  if (!(ps->theta_prev > 0)) return;
  ps->theta_prev *= factor;
  ps->hw = hw_max;
  ps->rescaled = 1;
  if (ps->resc_bad) ps->want_sample = 1;   // the last such prediction missed the range: sample
  for (int k = 0; k < L1_K; ++k) ps->t[k] = (double)(T)(ps->theta_prev * (1.0 + ps->hw * l1_probe_mult(k)));
  ps->spec_lo = ps->t[L1_WIN_LO];
  ps->spec_hi = ps->t[L1_WIN_HI];
}
template <typename T>
void K<T>::ps_rescale_multi(hipStream_t s, const RescaleMulti<T>& A) {
  if (A.n < 1) return;
  ObsScope obs_(KID_PS_RESCALE, s, 0.0);
  hipLaunchKernelGGL((k_ps_rescale_multi<T>), dim3(A.n), dim3(1), 0, s, A, l1_hw_max());
  SIPX_HIP(hipGetLastError());
}
// The sampled prediction of every sampling set of a slab-decomposed iteration: stage 10 = each rank's share of the sample of
// all sets in one launch (grid.y = set), stage 11 = the decisions on the all-reduced histograms in one launch.  Whether the
// sample is taken, its stride and the capacity it plans for depend on the whole grid and the number of ranks only (see chain_stage).
template <typename T>
void K<T>::sample_multi(int stage, hipStream_t s, const Grid& g, const SampleMulti<T>& A, long long runs, const ChainHooks* hk) {
  if (A.ns < 1 || g.n[0] % 4 != 0) return;
  // hk == nullptr: one rank, the decision is taken inside the sampling kernel (SampleSet::reg == nullptr), stage 10 only
  const int world = hk ? hk->world : 1;
  const double cap_max = hk ? (double)hk->gcap : 0.0;
  const long long nchunks = (range_len(g) / 4 + SAMPLE_RUN - 1) / SAMPLE_RUN;
  const long long nchunks_all = (g.N / 4 + SAMPLE_RUN - 1) / SAMPLE_RUN;
  const long long target = runs > 0 ? runs : (g.N >= (1ll << 26) ? 32768 : 16384);
  const long long stride = nchunks_all / target;
  const long long per_rank = hk ? (g.N / world + 3) / 4 : range_len(g) / 4;
  double gcap = 0.2 * (double)fit_grid(per_rank, SIPX_PASS_GRID) * (double)SPEC_CAP * (double)world;
  if (hk && gcap > 0.8 * cap_max) gcap = 0.8 * cap_max;
  if (hk && hk->fcap > 0 && gcap > 0.8 * (double)hk->fcap * (double)world) gcap = 0.8 * (double)hk->fcap * (double)world;
  if (!(stride >= 4 && (g.N >= (1ll << 24) || runs > 0))) return;
  const long long nsamp = nchunks / stride;            // may be 0 on a short slab
  ObsScope obs_(stage != 11 ? KID_SAMPLE : KID_DECIDE, s, 0.0);
  if (stage != 11)
    hipLaunchKernelGGL((k_sample_multi<T, 4>), dim3((unsigned)(nsamp < 1 ? 1 : (nsamp < SAMPLE_WG ? nsamp : SAMPLE_WG)), A.ns), dim3(SAMPLE_NT), 0, s, g, A,
                       nchunks, nsamp, (unsigned int)stride, l1_hw_max(), l1_lean_on(), gcap);
  else
    hipLaunchKernelGGL((k_sample_decide2_multi<T>), dim3(A.ns), dim3(SAMPLE_NT), 0, s, A, l1_hw_max(), l1_lean_on(), gcap);
  SIPX_HIP(hipGetLastError());
}

// algorithmic bytes of one sweep of k_pass: x (+ y, l of every block unless the vector is s = A x itself), or the stored array
template <typename T>
static double pass_bytes(const Grid& g, const SetArgs<T>& a, int v_is_s, int src, long long len, bool stores) {
  if (src == 0) return (double)len * sizeof(T);
  const double n = (double)range_len(g), nb = a.nblk > 0 ? a.nblk : 1;
  return (n + (v_is_s ? 0.0 : 2.0 * nb * n) + (stores ? nb * n : 0.0)) * sizeof(T);
}
constexpr int pass_kid(int mode) {
  return mode == M_FIRST ? KID_PASS_FIRST : mode == M_LEAN ? KID_PASS_LEAN : mode == M_PROBE ? KID_PASS_PROBE
       : mode == M_COMPACT ? KID_PASS_COMPACT : mode == M_DIST ? KID_PASS_DIST : KID_PASS_STORE;
}

// One search = four stages; on a slab-decomposed grid a collective sits between consecutive stages (and the engine runs the
// stages of ALL its sets in lock step, so that ONE all-reduce / all-gather serves them all):
//   0  [sampled prediction,] first pass (full or lean) + sums of its partial slots            -> all-reduce of the region
//   1  bracket decision; l1: gated refinement pass + sums                                        -> all-reduce of the region
//   2  l1: decision, gated compaction pass, the rank's gathered magnitudes into its segment      -> all-gather of the segments
//   3  l1: segments strung together, solve
// reg: where the sums go (ps->red itself, or the set's region of the staging buffer); gseg0 / chunk: the set's segment in
// rank 0's chunk of the exchange buffer and the distance to the next rank's.
template <typename T, int SRC>
static void chain_stage(int stage, hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, const T* varr, long long len,
                        ProjScalars<T>* ps, double* partials, T* maxpart, T* compact, long long true_len, SampleCtl ctl,
                        const ChainHooks* hk, long long compact_len, double* reg, T* gseg0, long long chunk) {
  const int world = hk ? hk->world : 1, rank = hk ? hk->rank : 0;
  const double cap_max = hk ? (double)hk->gcap : 0.0;      // what ALL ranks gather together fits one rank's segment
  const bool vec = SRC == 1 && g.n[0] % 4 == 0;
  static const double capdiv = [] { const char* e = getenv("SIPX_L1_CAPDIV"); return e ? atof(e) : 64.0; }();
  const DecideArgs da0{a.prox, (a.flags & F_NOSPEC) ? 1 : 0, (double)a.plo, (double)a.phi, capdiv, cap_max, true_len};
  const DecideArgs da1{a.prox, 0, (double)a.plo, (double)a.phi, capdiv, cap_max, true_len};
#define SIPX_PASS(MODE)                                                                                            \
  do {                                                                                                             \
    ObsScope obs_(pass_kid(MODE), s, pass_bytes<T>(g, a, v_is_s, SRC, len, false));                                \
    if (vec)                                                                                                       \
      hipLaunchKernelGGL((k_pass<T, 4, MODE, SRC>), dim3(fit_grid(range_len(g) / 4, SIPX_PASS_GRID)), dim3(BLOCK), 0, s, g, a, v_is_s, varr, len, ps, compact, \
                         partials, maxpart);                                                                       \
    else                                                                                                           \
      hipLaunchKernelGGL((k_pass<T, 1, MODE, SRC>), dim3(fit_grid(SRC == 0 ? len : range_len(g), SIPX_PASS_GRID)), dim3(BLOCK), 0, s, g, a, v_is_s, varr, len, ps, compact, \
                         partials, maxpart);                                                                       \
  } while (0)
  if (stage == 0 || stage == 10 || stage == 11) {
    // stage 0: sampled prediction with the decision inside the kernel (one rank).  Slab-decomposed: stage 10 = this rank's
    // share of the sample into the set's region `reg` of the sample staging buffer (2 SAMPLE_BINS + 3 doubles), an all-reduce
    // of the caller, stage 11 = the decision on the summed histogram.
    if (SRC == 1 && vec && ctl.enable && a.prox == PX_L1 && (!v_is_s || (!hk && stage == 0))) {      // (v = A x itself: one rank only)
      // about a million grid points (times the operator's blocks); not worth it when that is more than a quarter of the vector.
      // Slab-decomposed grid: WHETHER the sample is taken, its stride and the capacity it plans for are functions of the
      // whole grid and the number of ranks only -- a rank with a short (or empty) slab must take the same decisions as the
      // others, or the all-reduced histogram mixes searches probing at different thresholds.  Such a rank launches the
      // kernels all the same: with nothing to sample they write the zeros its share of the all-reduce has to hold.
      const long long nchunks = (range_len(g) / 4 + SAMPLE_RUN - 1) / SAMPLE_RUN;  // runs of 64 grid points on this rank
      const long long nchunks_all = hk ? (g.N / 4 + SAMPLE_RUN - 1) / SAMPLE_RUN : nchunks;
      const long long target = ctl.runs > 0 ? ctl.runs : (g.N >= (1ll << 26) ? 32768 : 16384);   // one or two million sampled points, over all ranks
      const long long stride = nchunks_all / target;
      const long long per_rank = hk ? (g.N / world + 3) / 4 : range_len(g) / 4;     // vectors of a full slab
      double gcap = 0.2 * (double)fit_grid(per_rank, SIPX_PASS_GRID) * (double)SPEC_CAP * (double)world;   // a fifth of the LDS buffers of the pass
      if (hk && gcap > 0.8 * cap_max) gcap = 0.8 * cap_max;
      if (hk && hk->fcap > 0 && gcap > 0.8 * (double)hk->fcap * (double)world) gcap = 0.8 * (double)hk->fcap * (double)world;   // the fast segments of the speculative exchange (margin: the ranks' shares are not equal)
      // (below 2^24 grid points the extra launch costs more than the sweeps it saves: 2048^2 loses 3 %; a test may force it)
      if (stride >= 4 && (g.N >= (1ll << 24) || ctl.runs > 0)) {
        const long long nsamp = nchunks / stride;            // may be 0 on a short slab
        ObsScope obs_(stage != 11 ? KID_SAMPLE : KID_DECIDE, s, 0.0);
        if (stage != 11)
          hipLaunchKernelGGL((k_sample<T, 4>), dim3((unsigned)(nsamp < 1 ? 1 : (nsamp < SAMPLE_WG ? nsamp : SAMPLE_WG))), dim3(SAMPLE_NT), 0, s, g, a, ps,
                             partials, nchunks, nsamp, (unsigned int)stride, true_len, l1_hw_max(), l1_lean_on(), gcap,
                             stage == 10 ? reg : (double*)nullptr, v_is_s);
        else
          hipLaunchKernelGGL((k_sample_decide2<T>), dim3(1), dim3(SAMPLE_NT), 0, s, ps, reg, a.phi, true_len, l1_hw_max(), l1_lean_on(), gcap);
      }
    }
    if (stage != 0) {
      SIPX_HIP(hipGetLastError());
      return;
    }
    SIPX_PASS(M_FIRST);
    if (a.prox == PX_L1 && !(a.flags & F_NOSPEC) && !ctl.lean_done) SIPX_PASS(M_LEAN);      // (lean_done: k_lean_multi took this set's lean pass)
    // one rank: the last workgroup of the sums takes the decision (no k_decide launch); slab-decomposed: an all-reduce of the
    // caller sits between the two
    ObsScope obs_(KID_SLOT_SUMS, s, 0.0);
    if (hk) hipLaunchKernelGGL((k_slot_sums<T, 0, false>), dim3(PREP_SLOTS + 1), dim3(BLOCK), 0, s, partials, maxpart, ps, rank, world, reg, da0);
    else hipLaunchKernelGGL((k_slot_sums<T, 0, true>), dim3(PREP_SLOTS + 1), dim3(BLOCK), 0, s, partials, maxpart, ps, rank, world, reg, da0);
  } else if (stage == 13) {     // speculative exchange, batched form: the first pass only (its sums and the packing: K::spec_sums_pack for all sets)
    if (!(ctl.lean_known && ctl.lean_done)) SIPX_PASS(M_FIRST);      // (known to be lean and served by k_lean_multi: the full pass would return at once)
    if (a.prox == PX_L1 && !(a.flags & F_NOSPEC) && !ctl.lean_done) SIPX_PASS(M_LEAN);
  } else if (stage == 5) {      // speculative exchange: this rank's sums and speculatively gathered magnitudes into its fast segment
    ObsScope obs_(KID_GATHER, s, 0.0);
    hipLaunchKernelGGL((k_spec_pack<T>), dim3(16), dim3(BLOCK), 0, s, ps, reg, rank, a.prox == PX_L1 ? 1 : 0, compact, gseg0 + (long long)rank * chunk,
                       hk->fcap);
  } else if (stage == 6) {      // ... after the all-gather: decision on the summed headers, the values strung together, the solve
    {
      ObsScope obs_(KID_DECIDE, s, 0.0);
      hipLaunchKernelGGL((k_spec_decide<T>), dim3(1), dim3(64), 0, s, ps, da0, world, reg, gseg0, chunk, ctl.seq, ctl.verdict);
    }
    if (a.prox == PX_L1) {
      {
        ObsScope obs_(KID_GATHER, s, 0.0);
        hipLaunchKernelGGL((k_spec_unpack<T>), dim3(16), dim3(BLOCK), 0, s, ps, compact, gseg0, chunk, world);
      }
      ObsScope obs_(KID_L1_SOLVE, s, 0.0);
      hipLaunchKernelGGL((k_l1_solve<T>), dim3(1), dim3(SIPX_SOLVE_NT), 0, s, ps, a.phi, compact, partials, true_len, l1_hw_max(), l1_lean_on(),
                         ctl.host_want, solve_coop_min(), 1);
    }
  } else if (stage == 9) {      // fallback of a speculative exchange: decision on the all-reduced sums of a refinement round + verdict
    if (a.prox == PX_L1) {
      ObsScope obs_(KID_DECIDE, s, 0.0);
      hipLaunchKernelGGL((k_decide_round<T>), dim3(1), dim3(64), 0, s, ps, a.prox, a.plo, a.phi, true_len, capdiv, world, cap_max, reg, ctl.seq,
                         ctl.verdict);
    }
  } else if (stage == 12) {     // ... its bracket is final: gated compaction pass, the rank's magnitudes into its segment
    if (a.prox == PX_L1) {
      SIPX_PASS(M_COMPACT);
      ObsScope obs_(KID_GATHER, s, 0.0);
      hipLaunchKernelGGL((k_gather_pack<T>), dim3(64), dim3(BLOCK), 0, s, ps, compact, partials, gseg0 + (long long)rank * chunk, hk->gcap);
    }
  } else if (stage == 1 || stage == 8) {      // (8: a refinement round whose decision was taken already -- k_spec_decide, k_decide_round)
    if (hk && stage == 1) {
      ObsScope obs_(KID_DECIDE, s, 0.0);
      hipLaunchKernelGGL((k_decide<T, 0>), dim3(1), dim3(64), 0, s, ps, a.prox, a.plo, a.phi, true_len, da0.nospec, capdiv,
                         world, cap_max, reg);
    }
    if (a.prox == PX_L1) {
      SIPX_PASS(M_PROBE);
      ObsScope obs_(KID_SLOT_SUMS, s, 0.0);
      if (hk) hipLaunchKernelGGL((k_slot_sums<T, 1, false>), dim3(PREP_SLOTS), dim3(BLOCK), 0, s, partials, maxpart, ps, rank, world, reg, da1);
      else hipLaunchKernelGGL((k_slot_sums<T, 1, true>), dim3(PREP_SLOTS), dim3(BLOCK), 0, s, partials, maxpart, ps, rank, world, reg, da1);
    }
  } else if (stage == 4) {      // slab-decomposed: one more gated refinement round (decision on the all-reduced sums, probe pass, sums)
    if (a.prox == PX_L1 && hk) {
      {
        ObsScope obs_(KID_DECIDE, s, 0.0);
        hipLaunchKernelGGL((k_decide<T, 1>), dim3(1), dim3(64), 0, s, ps, a.prox, a.plo, a.phi, true_len, 0, capdiv, world, cap_max, reg);
      }
      SIPX_PASS(M_PROBE);
      ObsScope obs_(KID_SLOT_SUMS, s, 0.0);
      hipLaunchKernelGGL((k_slot_sums<T, 1, false>), dim3(PREP_SLOTS), dim3(BLOCK), 0, s, partials, maxpart, ps, rank, world, reg, da1);
    }
  } else if (stage == 2) {
    if (a.prox == PX_L1) {
      if (hk) {
        ObsScope obs_(KID_DECIDE, s, 0.0);
        hipLaunchKernelGGL((k_decide<T, 1>), dim3(1), dim3(64), 0, s, ps, a.prox, a.plo, a.phi, true_len, 0, capdiv, world, cap_max, reg);
      }
      SIPX_PASS(M_COMPACT);
      if (hk) {
        ObsScope obs_(KID_GATHER, s, 0.0);
        hipLaunchKernelGGL((k_gather_pack<T>), dim3(64), dim3(BLOCK), 0, s, ps, compact, partials, gseg0 + (long long)rank * chunk, hk->gcap);
      }
    }
  } else if (a.prox == PX_L1) {
    if (hk) {
      ObsScope obs_(KID_GATHER, s, 0.0);
      hipLaunchKernelGGL((k_gather_unpack<T>), dim3(64), dim3(BLOCK), 0, s, ps, compact, partials, gseg0, chunk, world, compact_len,
                         ctl.host_ovf);
    }
    ObsScope obs_(KID_L1_SOLVE, s, 0.0);
    hipLaunchKernelGGL((k_l1_solve<T>), dim3(hk ? 1 : SOLVE_G), dim3(SIPX_SOLVE_NT), 0, s, ps, a.phi, compact, partials, true_len, l1_hw_max(),
                       l1_lean_on(), ctl.host_want, solve_coop_min(), 0);
  }
#undef SIPX_PASS
  SIPX_HIP(hipGetLastError());
}

template <typename T, int SRC>
static void launch_chain(hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, const T* varr, long long len,
                         ProjScalars<T>* ps, double* partials, T* maxpart, T* compact, long long true_len,
                         SampleCtl ctl = SampleCtl(), const ChainHooks* hk = nullptr, long long compact_len = 0) {
  if (a.prox == PX_CARD) {
    const bool vec = SRC == 1 && g.n[0] % 4 == 0;
#define SIPX_PASS(MODE)                                                                                            \
  do {                                                                                                             \
    ObsScope obs_(pass_kid(MODE), s, pass_bytes<T>(g, a, v_is_s, SRC, len, false));                                \
    if (vec)                                                                                                       \
      hipLaunchKernelGGL((k_pass<T, 4, MODE, SRC>), dim3(fit_grid(range_len(g) / 4, SIPX_PASS_GRID)), dim3(BLOCK), 0, s, g, a, v_is_s, varr, len, ps, compact, \
                         partials, maxpart);                                                                       \
    else                                                                                                           \
      hipLaunchKernelGGL((k_pass<T, 1, MODE, SRC>), dim3(fit_grid(SRC == 0 ? len : range_len(g), SIPX_PASS_GRID)), dim3(BLOCK), 0, s, g, a, v_is_s, varr, len, ps, compact, \
                         partials, maxpart);                                                                       \
  } while (0)
    const long long kc = (long long)a.phi;
    if (hk) {
      // slab-decomposed grid (round 5): every rank passes over its planes; one all-reduce per decision makes the probe counts
      // global, one all-gather strings the (magnitude, index) pairs inside the final bracket together; the same rounds on every rank
      const int world = hk->world, rank = hk->rank;
      const long long seg_T = hk->gcap + GATHER_HDR, seg_bytes = seg_T * (long long)sizeof(T);
      const long long cap = std::min<long long>(1ll << 16, (seg_bytes - 16) / (8 + (long long)sizeof(T)));
      if (cap < 1) throw std::runtime_error("the exchange segments are too small for a cardinality search");
      char* gb = static_cast<char*>(hk->gbuf);
      double* reg = ps->red;
      const size_t nreg = (size_t)(PREP_SLOTS + 1 + 2 * world);
      SIPX_PASS(M_FIRST);
      { ObsScope obs_(KID_CARD, s, 0.0); hipLaunchKernelGGL((k_card_decide<T, 0, 1>), dim3(1), dim3(1024), 0, s, partials, maxpart, ps, kc, true_len, world, rank, (double)std::min<long long>((long long)CARD_CAP, cap)); }
      hk->allreduce_sum(hk->user, reg, nreg, s);
      { ObsScope obs_(KID_CARD, s, 0.0); hipLaunchKernelGGL((k_card_decide<T, 0, 2>), dim3(1), dim3(1024), 0, s, partials, maxpart, ps, kc, true_len, world, rank, (double)std::min<long long>((long long)CARD_CAP, cap)); }
      for (int r = 0; r < CARD_REFINES; ++r) {
        SIPX_PASS(M_PROBE);
        { ObsScope obs_(KID_CARD, s, 0.0); hipLaunchKernelGGL((k_card_decide<T, 1, 1>), dim3(1), dim3(1024), 0, s, partials, maxpart, ps, kc, true_len, world, rank, (double)std::min<long long>((long long)CARD_CAP, cap)); }
        hk->allreduce_sum(hk->user, reg, (size_t)PREP_SLOTS, s);
        { ObsScope obs_(KID_CARD, s, 0.0); hipLaunchKernelGGL((k_card_decide<T, 1, 2>), dim3(1), dim3(1024), 0, s, partials, maxpart, ps, kc, true_len, world, rank, (double)std::min<long long>((long long)CARD_CAP, cap)); }
      }
      SIPX_PASS(M_COMPACT);
      {
        ObsScope obs_(KID_GATHER, s, 0.0);
        hipLaunchKernelGGL((k_card_pack<T>), dim3(16), dim3(BLOCK), 0, s, ps, compact, gb + (long long)rank * seg_bytes, cap);
      }
      hk->allgather(hk->user, gb, (size_t)seg_T, sizeof(T) == 8 ? 1 : 0, s);
      {
        ObsScope obs_(KID_GATHER, s, 0.0);
        hipLaunchKernelGGL((k_card_unpack<T>), dim3(1), dim3(1024), 0, s, ps, compact, gb, seg_bytes, world, cap, ctl.host_ovf);
      }
      { ObsScope obs_(KID_CARD, s, 0.0); hipLaunchKernelGGL((k_card_select<T>), dim3(1), dim3(1024), 0, s, ps, kc, compact); }
      SIPX_HIP(hipGetLastError());
      return;
    }
    SIPX_PASS(M_FIRST);
    { ObsScope obs_(KID_CARD, s, 0.0); hipLaunchKernelGGL((k_card_decide<T, 0>), dim3(1), dim3(1024), 0, s, partials, maxpart, ps, kc, true_len); }
    for (int r = 0; r < CARD_REFINES; ++r) {
      SIPX_PASS(M_PROBE);
      ObsScope obs_(KID_CARD, s, 0.0);
      hipLaunchKernelGGL((k_card_decide<T, 1>), dim3(1), dim3(1024), 0, s, partials, maxpart, ps, kc, true_len);
    }
    SIPX_PASS(M_COMPACT);
    { ObsScope obs_(KID_CARD, s, 0.0); hipLaunchKernelGGL((k_card_select<T>), dim3(1), dim3(1024), 0, s, ps, kc, compact); }
    SIPX_HIP(hipGetLastError());
#undef SIPX_PASS
    return;
  }
  // the four stages back to back; on a slab-decomposed grid with this search's own collectives in between
  static_assert(L1_REFINES == 1, "one refinement stage");
  const int world = hk ? hk->world : 1;
  const long long chunk = hk ? hk->gcap + GATHER_HDR : 0;
  T* gb = hk ? static_cast<T*>(hk->gbuf) : nullptr;
  double* reg = ps->red;                       // red, ovf, mm: adjacent doubles
  chain_stage<T, SRC>(0, s, g, a, v_is_s, varr, len, ps, partials, maxpart, compact, true_len, ctl, hk, compact_len, reg, gb, chunk);
  if (hk) hk->allreduce_sum(hk->user, reg, (size_t)(PREP_SLOTS + 1 + 2 * world), s);
  chain_stage<T, SRC>(1, s, g, a, v_is_s, varr, len, ps, partials, maxpart, compact, true_len, ctl, hk, compact_len, reg, gb, chunk);
  if (hk && a.prox == PX_L1) hk->allreduce_sum(hk->user, reg, (size_t)PREP_SLOTS, s);
  for (int r = 1; hk && a.prox == PX_L1 && r < L1_REFINES_SLAB; ++r) {      // (searches with collectives of their own are rare: all rounds)
    chain_stage<T, SRC>(4, s, g, a, v_is_s, varr, len, ps, partials, maxpart, compact, true_len, ctl, hk, compact_len, reg, gb, chunk);
    hk->allreduce_sum(hk->user, reg, (size_t)PREP_SLOTS, s);
  }
  chain_stage<T, SRC>(2, s, g, a, v_is_s, varr, len, ps, partials, maxpart, compact, true_len, ctl, hk, compact_len, reg, gb, chunk);
  if (hk && a.prox == PX_L1) hk->allgather(hk->user, gb, (size_t)chunk, sizeof(T) == 8 ? 1 : 0, s);
  chain_stage<T, SRC>(3, s, g, a, v_is_s, varr, len, ps, partials, maxpart, compact, true_len, ctl, hk, compact_len, reg, gb, chunk);
  SIPX_HIP(hipGetLastError());
}

template <typename T>
void K<T>::search_tail(int stage, hipStream_t s, const SetArgs<T>& a, ProjScalars<T>* ps, double* partials, T* maxpart, T* compact,
                       long long true_len, SampleCtl ctl, double* reg) {
  static const double capdiv = [] { const char* e = getenv("SIPX_L1_CAPDIV"); return e ? atof(e) : 64.0; }();
  if (a.prox != PX_L1) return;
  if (stage == 1) {
    const DecideArgs da1{a.prox, 0, (double)a.plo, (double)a.phi, capdiv, 0.0, true_len};
    ObsScope obs_(KID_SLOT_SUMS, s, 0.0);
    hipLaunchKernelGGL((k_slot_sums<T, 1, true>), dim3(PREP_SLOTS), dim3(BLOCK), 0, s, partials, maxpart, ps, 0, 1, reg ? reg : ps->red, da1);
  } else {
    ObsScope obs_(KID_L1_SOLVE, s, 0.0);
    hipLaunchKernelGGL((k_l1_solve<T>), dim3(SOLVE_G), dim3(SIPX_SOLVE_NT), 0, s, ps, a.phi, compact, partials, true_len, l1_hw_max(), l1_lean_on(),
                       ctl.host_want, solve_coop_min(), 0);
  }
  SIPX_HIP(hipGetLastError());
}

template <typename T>
void K<T>::proj_scalars_set(hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, ProjScalars<T>* ps,
                            double* partials, T* maxpart, T* compact, long long true_len, SampleCtl ctl, const ChainHooks* hooks) {
  SetArgs<T> b = a;
  b.ps = ps;
  launch_chain<T, 1>(s, g, b, v_is_s, nullptr, 0, ps, partials, maxpart, compact, true_len, ctl, hooks,
                     ctl.compact_cap > 0 ? ctl.compact_cap : (long long)a.nblk_or1() * g.N);
}
template <typename T>
void K<T>::proj_scalars_stage(int stage, hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, ProjScalars<T>* ps,
                              double* partials, T* maxpart, T* compact, long long true_len, SampleCtl ctl, const ChainHooks* hooks,
                              double* reg, T* gseg0, long long chunk) {
  if (a.prox == PX_CARD) throw std::runtime_error("the cardinality search has no staged form");
  SetArgs<T> b = a;
  b.ps = ps;
  chain_stage<T, 1>(stage, s, g, b, v_is_s, nullptr, 0, ps, partials, maxpart, compact, true_len, ctl, hooks,
                    ctl.compact_cap > 0 ? ctl.compact_cap : (long long)a.nblk_or1() * g.N, reg ? reg : ps->red, gseg0, chunk);
}
template <typename T>
void K<T>::proj_scalars_arr(hipStream_t s, long long len, const T* v, int prox, T pmin, T pmax, ProjScalars<T>* ps,
                            double* partials, T* maxpart, T* compact, long long true_len) {
  SetArgs<T> a = {};
  a.prox = prox;
  a.plo = pmin;
  a.phi = pmax;
  a.ps = ps;
  Grid g = {};
  g.n[0] = len; g.n[1] = 1; g.n[2] = 1; g.N = len; g.st[0] = 1; g.st[1] = len; g.st[2] = len;
  launch_chain<T, 0>(s, g, a, 0, v, len, ps, partials, maxpart, compact, true_len);
}
// ... of a stored array that is this rank's share of a vector spread over the ranks of a slab-decomposed solve (the coefficients
// of the slab-decomposed DFT, dist_dft.hip): the chain with the slab collectives of `hooks` between its stages -- all-reduced
// probe sums, all-gathered bracket -- so that every rank arrives at the same scalars.  true_len: entries over ALL ranks.
template <typename T>
void K<T>::proj_scalars_arr_slab(hipStream_t s, long long len, const T* v, int prox, T pmin, T pmax, ProjScalars<T>* ps,
                                 double* partials, T* maxpart, T* compact, long long true_len, const ChainHooks* hooks,
                                 long long compact_len, int* host_ovf) {
  SetArgs<T> a = {};
  a.prox = prox;
  a.plo = pmin;
  a.phi = pmax;
  a.ps = ps;
  Grid g = {};
  g.n[0] = len > 0 ? len : 1; g.n[1] = 1; g.n[2] = 1; g.N = len; g.st[0] = 1; g.st[1] = g.n[0]; g.st[2] = g.n[0];
  SampleCtl ctl;
  ctl.host_ovf = host_ovf;
  launch_chain<T, 0>(s, g, a, 0, v, len, ps, partials, maxpart, compact, true_len, ctl, hooks, compact_len);
}
template <typename T>
void K<T>::store_v(hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, T* out) {
  ObsScope obs_(KID_PASS_STORE, s, pass_bytes<T>(g, a, v_is_s, 1, 0, true));
  if (g.n[0] % 4 == 0)
    hipLaunchKernelGGL((k_pass<T, 4, M_STORE, 1>), dim3(NB_7), dim3(BLOCK), 0, s, g, a, v_is_s, (const T*)nullptr, 0ll,
                       (ProjScalars<T>*)nullptr, out, (double*)nullptr, (T*)nullptr);
  else
    hipLaunchKernelGGL((k_pass<T, 1, M_STORE, 1>), dim3(NB_7), dim3(BLOCK), 0, s, g, a, v_is_s, (const T*)nullptr, 0ll,
                       (ProjScalars<T>*)nullptr, out, (double*)nullptr, (T*)nullptr);
  SIPX_HIP(hipGetLastError());
}
template <typename T>
void K<T>::proj_dist_set(hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, const ProjScalars<T>* ps,
                         double* dst) {
  SetArgs<T> b = a;
  b.ps = ps;
  ProjScalars<T>* psm = const_cast<ProjScalars<T>*>(ps);
  ObsScope obs_(KID_PASS_DIST, s, pass_bytes<T>(g, a, v_is_s, 1, 0, false));
  if (g.n[0] % 4 == 0)
    hipLaunchKernelGGL((k_pass<T, 4, M_DIST, 1>), dim3(NB_7), dim3(BLOCK), 0, s, g, b, v_is_s, (const T*)nullptr, 0ll, psm,
                       (T*)nullptr, dst, (T*)nullptr);
  else
    hipLaunchKernelGGL((k_pass<T, 1, M_DIST, 1>), dim3(NB_7), dim3(BLOCK), 0, s, g, b, v_is_s, (const T*)nullptr, 0ll, psm,
                       (T*)nullptr, dst, (T*)nullptr);
  SIPX_HIP(hipGetLastError());
}

#define SIPX_INST(T)                                                                                              \
  template void K<T>::ps_init(hipStream_t, ProjScalars<T>*, long long*);                                         \
  template void K<T>::ps_rescale(hipStream_t, ProjScalars<T>*, double);                                          \
  template void K<T>::store_v(hipStream_t, const Grid&, const SetArgs<T>&, int, T*);                                                     \
  template void K<T>::proj_scalars_set(hipStream_t, const Grid&, const SetArgs<T>&, int, ProjScalars<T>*, double*, \
                                       T*, T*, long long, SampleCtl, const ChainHooks*);                                        \
  template void K<T>::lean_multi(hipStream_t, const Grid&, const LeanMulti<T>&);                                                        \
  template void K<T>::pass_multi(int, hipStream_t, const Grid&, const LeanMulti<T>&, int);                                              \
  template void K<T>::search_tail(int, hipStream_t, const SetArgs<T>&, ProjScalars<T>*, double*, T*, T*, long long, SampleCtl, double*);   \
  template void K<T>::spec_sums_pack(hipStream_t, const SpecPackArgs<T>&);                                                              \
  template void K<T>::ps_rescale_multi(hipStream_t, const RescaleMulti<T>&);                                                            \
  template void K<T>::sample_multi(int, hipStream_t, const Grid&, const SampleMulti<T>&, long long, const ChainHooks*);                 \
  template void K<T>::spec_finish(hipStream_t, SpecFinishArgs<T>&);                                                                     \
  template void K<T>::proj_scalars_stage(int, hipStream_t, const Grid&, const SetArgs<T>&, int, ProjScalars<T>*, double*, T*, T*,   \
                                         long long, SampleCtl, const ChainHooks*, double*, T*, long long);                                                     \
  template void K<T>::proj_scalars_arr(hipStream_t, long long, const T*, int, T, T, ProjScalars<T>*, double*, T*, \
                                       T*, long long);                                                           \
  template void K<T>::proj_scalars_arr_slab(hipStream_t, long long, const T*, int, T, T, ProjScalars<T>*, double*, T*, T*, long long, \
                                            const ChainHooks*, long long, int*);                                     \
  template void K<T>::proj_dist_set(hipStream_t, const Grid&, const SetArgs<T>&, int, const ProjScalars<T>*, double*);
SIPX_INST(float)
SIPX_INST(double)

}  // namespace sipx
