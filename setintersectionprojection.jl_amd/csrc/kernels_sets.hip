// Per-set kernels of the PARSDMM iteration body: matrix-free difference stencils fused with
// the y/l update, the proximal maps, the residual / Barzilai-Borwein / feasibility reductions.
//
// Replaces (reference file:line):
//   rhs_compose                         src/rhs_compose.jl:24-36
//   update_y_l (per set)                src/update_y_l.jl:36-101   (Blas_active=false formulas :64-78)
//   l_hat + snapshots + 5 BB differences  src/PARSDMM.jl:164-180,192-206, src/adapt_rho_gamma.jl:41-53
//   prox_l2s!, prox_l1!, project_bounds!, project_l1/l2/annulus (apply step)  src/prox_*.jl, src/projectors/*.jl
//   obj / evol_x logging                src/PARSDMM.jl:140,145
//
// Data layout: every transform-domain vector (y_i, l_i, ...) of a difference operator is stored
// PADDED to the grid: block q of the operator occupies [q*N, (q+1)*N) and element g of a block
// sits at the grid point whose forward difference it is; the last hyper-plane along the
// difference direction is a pad that is kept exactly zero.  Row r of the reference's matrix
// maps monotonically onto the valid points, so index order (tie breaking) is preserved, every
// access is 16-byte aligned and stencil neighbours are plain +-stride offsets.
#include <stdexcept>
#include <string>

#include "sipx_device.h"
#ifndef SIPX_F64_VEC
#define SIPX_F64_VEC 2
#endif

namespace sipx {

// ---------------------------------------------------------------------------------------------
// rhs = sum_i A_i'(rho_i y_i + l_i): every owned set in one pass, sets added in order
// (rhs_compose.jl:24-31); algorithmic bytes (sum_i 2 M_i + N) * w.
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_rhs(Grid G, RhsArgs<T> a, T* __restrict__ rhs, int accumulate) {
  long long v0, nvec;
  vec_range<V>(G, v0, nvec);
  for (long long vi = v0 + (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long g = vi * V;
    const Coord c = coords(G, g);
    T out[V];
    if (accumulate) {
      const Vec<T, V> r0 = ldv<T, V>(rhs + g);
#pragma unroll
      for (int k = 0; k < V; ++k) out[k] = r0.v[k];
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) out[k] = T(0);
    }
    for (int si = 0; si < a.nsets; ++si) {
      const RhsSet<T>& S = a.s[si];
      const T rho = S.rho;
      T t[V];
#pragma unroll
      for (int k = 0; k < V; ++k) t[k] = T(0);
      if (S.nblk == 0) {
        const Vec<T, V> yv = ldv<T, V>(S.y + g), lv = ldv<T, V>(S.l + g);
#pragma unroll
        for (int k = 0; k < V; ++k) t[k] = rho * yv.v[k] + lv.v[k];
      } else {
        for (int q = 0; q < S.nblk; ++q) {
          const T* yb = S.y + (long long)q * G.N;
          const T* lb = S.l + (long long)q * G.N;
          auto wv = [&](long long e) {
            const Vec<T, V> yv = ldv_u<T, V>(yb + e), lv = ldv_u<T, V>(lb + e);
            Vec<T, V> w;
#pragma unroll
            for (int k = 0; k < V; ++k) w.v[k] = rho * yv.v[k] + lv.v[k];
            return w;
          };
          adj_dir_acc<T, V>(G, g, c, S.dir[q], S.ih[q], t, wv);
        }
      }
#pragma unroll
      for (int k = 0; k < V; ++k) out[k] = out[k] + t[k];
    }
    Vec<T, V> o;
#pragma unroll
    for (int k = 0; k < V; ++k) o.v[k] = out[k];
    stv<T, V>(rhs + g, o);
  }
}
// The same sum marched along z (3-D grids, round 3): a workgroup owns a tile of (V LX) x TY points of a plane and walks a chunk
// of planes.  w = rho y + l of every block is formed once per point; the adjoint stencils take w[g - stride] from the lane next
// door (x; one element from memory at a wave / tile edge), from the thread one row up through LDS (y; the row in front of the
// tile from memory) and from the thread's own value of the previous plane (z; the plane in front of a chunk from memory) --
// where k_rhs re-reads y and l at g - stride through the caches (1.32 x its algorithmic bytes at 512^3).  Same products, same
// order (per set: blocks in order, t += ih w[g - st] then t += -ih w[g] under the same masks; sets added in order): the same bits.
constexpr int RM_NT = 256, RM_MAXB = 8, RM_MAXY = 3;
template <typename T>
struct RhsMarchBlk {
  const T *y, *l;       // base of the block (already offset by q N)
  T rho, ih;
  int dir;              // -1: identity set (one "block"), 0 / 1 / 2: difference along x / y / z
  int yslot;            // LDS slot of a y-difference block
  int last;             // last block of its set: the set's sum is added to the result
};
template <typename T>
struct RhsMarchArgs {
  int nb;
  RhsMarchBlk<T> b[RM_MAXB];
};
template <typename T, int V>
__global__ __launch_bounds__(RM_NT) void k_rhs_march(Grid G, RhsMarchArgs<T> a, T* __restrict__ rhs, long long zlo, long long zhi, int lgLX,
                                                     int tiles_x, int tiles_y, int zchunk, long long items) {
  __shared__ T ybuf[2][RM_MAXY][V][RM_NT];
  const int tid = threadIdx.x, LX = 1 << lgLX, tx = tid & (LX - 1), ty = tid >> lgLX, TY = RM_NT >> lgLX;
  const long long n1 = G.n[0], n2 = G.n[1], st1 = G.st[1], st2 = G.st[2];
  const long long tiles = (long long)tiles_x * tiles_y;
  for (long long item = blockIdx.x; item < items; item += gridDim.x) {
    const long long zc = item / tiles, tile = item - zc * tiles;
    const int tile_y = (int)(tile / tiles_x), tile_x = (int)(tile - (long long)tile_y * tiles_x);
    const long long i0 = ((long long)tile_x * LX + tx) * V, j = (long long)tile_y * TY + ty;
    const bool active = i0 < n1 && j < n2;
    const long long k0 = zlo + zc * zchunk, k1 = (k0 + zchunk < zhi) ? k0 + zchunk : zhi;
    const long long go = active ? i0 + st1 * j : 0;
    __syncthreads();                                   // the previous item's LDS traffic is over
    T zprev[RM_MAXB][V];                               // w of the previous plane (z-difference blocks)
#pragma unroll
    for (int b = 0; b < RM_MAXB; ++b) {
#pragma unroll
      for (int k = 0; k < V; ++k) zprev[b][k] = T(0);
      if (b < a.nb && a.b[b].dir == 2 && active) {     // (the vectors carry a front halo: plane k0 - 1 of plane 0 reads zeros, masked anyway)
        const long long e = st2 * (k0 - 1) + go;
        const Vec<T, V> yv = ldv_u<T, V>(a.b[b].y + e), lv = ldv_u<T, V>(a.b[b].l + e);
#pragma unroll
        for (int k = 0; k < V; ++k) zprev[b][k] = a.b[b].rho * yv.v[k] + lv.v[k];
      }
    }
    for (long long kz = k0; kz < k1; ++kz) {
      const int par = (int)(kz & 1);
      const long long e0 = st2 * kz + go;
      T w[RM_MAXB][V];
#pragma unroll
      for (int b = 0; b < RM_MAXB; ++b) {
#pragma unroll
        for (int k = 0; k < V; ++k) w[b][k] = T(0);
        if (b < a.nb && active) {
          const Vec<T, V> yv = ldv<T, V>(a.b[b].y + e0), lv = ldv<T, V>(a.b[b].l + e0);
#pragma unroll
          for (int k = 0; k < V; ++k) w[b][k] = a.b[b].rho * yv.v[k] + lv.v[k];
        }
        if (b < a.nb && a.b[b].dir == 1) {
#pragma unroll
          for (int k = 0; k < V; ++k) ybuf[par][a.b[b].yslot][k][tid] = w[b][k];
        }
      }
      __syncthreads();
      if (active) {
        T out[V], t[V];
#pragma unroll
        for (int k = 0; k < V; ++k) out[k] = t[k] = T(0);
#pragma unroll
        for (int b = 0; b < RM_MAXB; ++b) {
          if (b < a.nb) {
            const RhsMarchBlk<T>& B = a.b[b];
            if (B.dir < 0) {
#pragma unroll
              for (int k = 0; k < V; ++k) t[k] = w[b][k];
            } else {
              const T ih = B.ih, nih = -B.ih;
              T wp[V];
              int cd, nd;
              if (B.dir == 0) {
                T left = __shfl_up(w[b][V - 1], 1, 64);
                if (tx == 0 || (tid & 63) == 0) left = B.rho * B.y[e0 - 1] + B.l[e0 - 1];       // (front halo: in bounds; masked at i = 0)
                wp[0] = left;
#pragma unroll
                for (int k = 1; k < V; ++k) wp[k] = w[b][k - 1];
                cd = (int)i0; nd = (int)n1;
              } else if (B.dir == 1) {
                if (ty > 0) {
#pragma unroll
                  for (int k = 0; k < V; ++k) wp[k] = ybuf[par][B.yslot][k][tid - LX];
                } else {
                  const Vec<T, V> yv = ldv_u<T, V>(B.y + e0 - st1), lv = ldv_u<T, V>(B.l + e0 - st1);
#pragma unroll
                  for (int k = 0; k < V; ++k) wp[k] = B.rho * yv.v[k] + lv.v[k];
                }
                cd = (int)j; nd = (int)n2;
              } else {
#pragma unroll
                for (int k = 0; k < V; ++k) wp[k] = zprev[b][k];
                cd = (int)kz; nd = (int)G.n[2];
              }
#pragma unroll
              for (int k = 0; k < V; ++k) {
                const int ck = cd + (B.dir == 0 ? k : 0);
                const T t1 = t[k] + ih * wp[k];
                t[k] = (ck > 0) ? t1 : t[k];
                const T t2 = t[k] + nih * w[b][k];
                t[k] = (ck < nd - 1) ? t2 : t[k];
              }
            }
            if (B.last) {
#pragma unroll
              for (int k = 0; k < V; ++k) { out[k] = out[k] + t[k]; t[k] = T(0); }
            }
          }
        }
        Vec<T, V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] = out[k];
        stv<T, V>(rhs + e0, o);
      }
#pragma unroll
      for (int b = 0; b < RM_MAXB; ++b)
#pragma unroll
        for (int k = 0; k < V; ++k) zprev[b][k] = w[b][k];
    }
  }
}
// false: the march does not apply (2-D grid, too many blocks, a grid too small to fill the chip that way): k_rhs
template <typename T>
static bool try_rhs_march(hipStream_t s, const Grid& g, const RhsArgs<T>& a, T* rhs) {
  const int sw = env_knobs().rhs_march;                 // SIPX_RHS_MARCH 0: never; 2: also on small grids, chunks of SIPX_RHS_MARCH_ZCHUNK planes (tests)
  constexpr int V = sizeof(T) == 8 ? 2 : 4;
  const long long n1 = g.n[0], n2 = g.n[1], n3 = g.n[2], st2 = n1 * n2;
  if (sw == 0 || n3 < 2 || n1 % V != 0 || a.nsets < 1) return false;
  const long long e0 = g.e0, e1 = g.e1 < 0 ? g.N : g.e1;
  if (e0 % st2 != 0 || e1 % st2 != 0 || e1 <= e0) return false;
  RhsMarchArgs<T> m;
  m.nb = 0;
  int ny = 0;
  for (int i = 0; i < a.nsets; ++i) {
    const RhsSet<T>& S = a.s[i];
    const int nb = S.nblk > 0 ? S.nblk : 1;
    if (m.nb + nb > RM_MAXB) return false;
    for (int q = 0; q < nb; ++q) {
      RhsMarchBlk<T>& B = m.b[m.nb++];
      B.y = S.y + (long long)q * g.N; B.l = S.l + (long long)q * g.N;
      B.rho = S.rho; B.ih = S.nblk > 0 ? S.ih[q] : T(0);
      B.dir = S.nblk > 0 ? S.dir[q] : -1;
      B.yslot = 0;
      if (B.dir == 1) { if (ny == RM_MAXY) return false; B.yslot = ny++; }
      B.last = q == nb - 1 ? 1 : 0;
    }
  }
  const long long nvx = n1 / V;
  int lg = 0;
  while ((1 << lg) < nvx && lg < 6) ++lg;
  const int LX = 1 << lg, TY = RM_NT / LX;
  const int tiles_x = (int)((nvx + LX - 1) / LX), tiles_y = (int)((n2 + TY - 1) / TY);
  const long long tiles = (long long)tiles_x * tiles_y, planes = (e1 - e0) / st2;
  long long zchunk = planes * tiles / 4096;
  if (zchunk < 16) zchunk = 16;
  if (sw == 2 && env_knobs().rhs_march_zchunk > 0) zchunk = env_knobs().rhs_march_zchunk;
  if (zchunk > planes) zchunk = planes;
  const long long nchunks = (planes + zchunk - 1) / zchunk, items = tiles * nchunks;
  // (up to 2^24 grid points y and l of the neighbouring planes are still in the Infinity Cache when k_rhs re-reads them: 256^3
  //  140 us against 153 us marched; 512^3 1335 -> 1137 us)
  if ((items < 768 || g.N <= (1ll << 24)) && sw != 2) return false;
  const int grid = (int)(items < 2 * NB ? items : 2 * NB);
  hipLaunchKernelGGL((k_rhs_march<T, V>), dim3(grid), dim3(RM_NT), 0, s, g, m, rhs, e0 / st2, e1 / st2, lg, tiles_x, tiles_y, (int)zchunk, items);
  return true;
}

template <typename T>
void K<T>::rhs_compose(hipStream_t s, const Grid& g, const RhsArgs<T>& a, T* rhs, int accumulate) {
  double vecs = 1.0 + (accumulate ? 1.0 : 0.0);      // (sum_i 2 M_i + N) w: y_i, l_i of every set read, rhs written (rhs_compose.jl:24-36)
  for (int i = 0; i < a.nsets; ++i) vecs += 2.0 * (a.s[i].nblk > 0 ? a.s[i].nblk : 1);
  ObsScope obs(KID_RHS, s, vecs * (double)range_len(g) * sizeof(T));
  if (!accumulate && try_rhs_march<T>(s, g, a, rhs)) {
    SIPX_HIP(hipGetLastError());
    return;
  }
  if (g.n[0] % 4 == 0)
    hipLaunchKernelGGL((k_rhs<T, 4>), dim3(fit_grid(range_len(g) / 4, NB)), dim3(BLOCK), 0, s, g, a, rhs, accumulate);
  else
    hipLaunchKernelGGL((k_rhs<T, 1>), dim3(fit_grid(range_len(g), NB)), dim3(BLOCK), 0, s, g, a, rhs, accumulate);
  SIPX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// The fused y/l update of one set (update_y_l.jl:36-101).
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_yl(Grid G, SetArgs<T> a, double* __restrict__ partials) {
  double acc[YL_SLOTS];
#pragma unroll
  for (int k = 0; k < YL_SLOTS; ++k) acc[k] = 0;
  const bool ident = a.nblk == 0;
  const int nb = ident ? 1 : a.nblk;
  const bool relax = !(a.gamma == T(1));
  const T gam = a.gamma, omg = T(1) - a.gamma;
  const ProxCtx<T> pc = make_prox<T>(a.prox, a.plo, a.phi, a.rho, a.ps);
  const bool elementwise = (a.prox == PX_BOUNDS || a.prox == PX_BOUNDS_VEC || a.prox == PX_PROX_L1);
  const bool feas = (a.flags & F_FEAS) && elementwise;
  const bool first = (a.flags & F_FIRST) != 0;
  const bool bb = (a.flags & F_BB) && !first;
  const bool dist = a.prox == PX_DIST;
  long long v0, nvec;
  vec_range<V>(G, v0, nvec);
  for (long long vi = v0 + (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long g = vi * V;
    const bool summed = g >= G.s0;         // false on the recomputed plane in front of the rank's own slab
    const Coord c = coords(G, g);
    const Vec<T, V> xc = ldv<T, V>(a.x + g);
    for (int q = 0; q < nb; ++q) {
      const long long e = (long long)q * G.N + g;
      T s[V];
      bool valid[V];
      if (ident) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
          s[k] = xc.v[k];
          valid[k] = true;
        }
      } else {
        fwd_dir<T, V>(G, a.x, xc, g, c, a.dir[q], a.ih[q], s, valid);
      }
      const Vec<T, V> yv = ldv_nt<T, V>(a.y + e), lv = ldv_nt<T, V>(a.l + e);   // last use of the old iterate: streaming loads (+1.5 %)
      Vec<T, V> vv = zerov<T, V>(), lbv = zerov<T, V>(), ubv = zerov<T, V>(), mv = zerov<T, V>();
      if (a.vsrc) vv = ldv<T, V>(a.v + e);
      if (a.prox == PX_BOUNDS_VEC) {
        lbv = ldv<T, V>(a.lb + e);
        ubv = ldv<T, V>(a.ub + e);
      }
      if (dist) mv = ldv<T, V>(a.m + g);
      Vec<T, V> yn, ln, dyv, lh;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const T yo = yv.v[k], lo = lv.v[k];
        const T xh = relax ? (gam * s[k] + omg * yo) : s[k];           // update_y_l.jl:72
        const T v = a.vsrc ? vv.v[k] : (xh - lo * a.rho1);              // :67 / :74
        T y1 = (a.vsrc == 2) ? vv.v[k] : prox_apply<T>(pc, v, lbv.v[k], ubv.v[k], mv.v[k], e + k);  // :68 / :75
        if (!valid[k]) y1 = T(0);
        const T rp = y1 - s[k];                                         // r_pri = -s + y   :69 / :76
        const T l1 = relax ? (lo + a.rho * (y1 - xh)) : (lo + a.rho * rp);  // :70 / :77
        yn.v[k] = y1;
        ln.v[k] = l1;
        dyv.v[k] = y1 - yo;                                             // x_hat = y - y_old  :82
        lh.v[k] = lo + a.rho * (yo - s[k]);                             // l_hat = l_old + rho(-s + y_old)  PARSDMM.jl:173
        if (summed) acc[SL_RPRI] += (double)rp * (double)rp;
        if (ident && summed) acc[SL_DY] += (double)dyv.v[k] * (double)dyv.v[k];
        if (feas && valid[k] && summed) {                               // update_y_l.jl:90-99
          const T ps = prox_apply<T>(pc, s[k], lbv.v[k], ubv.v[k], T(0), e + k);
          const T d = ps - s[k];
          acc[SL_FE] += (double)d * (double)d;
          acc[SL_SS] += (double)s[k] * (double)s[k];
        }
      }
      if (dist && summed) {                                              // PARSDMM.jl:140,145
        const Vec<T, V> xo = ldv<T, V>(a.xold + g);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const T d = xc.v[k] - mv.v[k], ev = xo.v[k] - xc.v[k];
          acc[SL_OBJ] += (double)d * (double)d;
          acc[SL_EVO] += (double)ev * (double)ev;
          acc[SL_XX] += (double)xc.v[k] * (double)xc.v[k];
        }
      }
      if (bb) {                                                          // adapt_rho_gamma.jl:41-53
        // the snapshot arrays are touched once every rho_update_frequency iterations: streaming loads and stores keep them
        // from evicting the vectors the next kernels re-read (+2 % at 256^3)
        const Vec<T, V> a0 = ldv_nt<T, V>(a.lh0 + e), b0 = ldv_nt<T, V>(a.y0 + e), c0 = ldv_nt<T, V>(a.s0 + e),
                        d0 = ldv_nt<T, V>(a.l0 + e);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          if (!summed) continue;
          const T dlh = lh.v[k] - a0.v[k], dH = s[k] - c0.v[k], dl = ln.v[k] - d0.v[k], dG = -(yn.v[k] - b0.v[k]);
          acc[SL_HL] += (double)dH * (double)dlh;
          acc[SL_HH] += (double)dH * (double)dH;
          acc[SL_LH] += (double)dlh * (double)dlh;
          acc[SL_DL] += (double)dl * (double)dl;
          acc[SL_GG] += (double)dG * (double)dG;
          acc[SL_GL] += (double)dG * (double)dl;
        }
      }
      if (bb || first) {                                                 // PARSDMM.jl:174-177, 200-203
        Vec<T, V> sv;
#pragma unroll
        for (int k = 0; k < V; ++k) sv.v[k] = s[k];
        stv_nt<T, V>(a.lh0 + e, lh);       // y_0 <- y and l_0 <- l need no copy: on these iterations the update below
        stv_nt<T, V>(a.s0 + e, sv);        // is written into the snapshot arrays themselves (engine, update_y_l)
      }
      stv_nt<T, V>(a.yo + e, yn);        // streaming stores: +1 % at 256^3 over cached ones
      stv_nt<T, V>(a.lo + e, ln);
      if (!ident || (a.flags & F_STORE_DY)) stv<T, V>(a.dy + e, dyv);
    }
  }
  block_reduce_store<YL_SLOTS>(acc, partials, 0);
}
// Algorithmic bytes of one y/l update: x (and its stencil neighbours, cached) + y, l read + y, l written = (N + 4 M) w;
// + M w for y - y_old of a difference operator (read again by k_adj_norm); Barzilai-Borwein iterations read the four
// snapshot arrays and rewrite two of them (+6 M; the first iteration only writes the two: +2 M); the distance term reads
// m and x_old (+2 N); per-element bounds read lb, ub (+2 M); a materialised v / y arrives from an array (+M).
template <typename T>
double yl_bytes(const Grid& g, const SetArgs<T>& a) {
  const double n = (double)range_len(g), nb = a.nblk > 0 ? a.nblk : 1, M = nb * n;
  double v = n + 4.0 * M;
  if (a.nblk > 0 || (a.flags & F_STORE_DY)) v += M;
  const bool first = (a.flags & F_FIRST) != 0, bb = (a.flags & F_BB) && !first;
  if (bb) v += 6.0 * M;
  else if (first) v += 2.0 * M;
  if (a.prox == PX_DIST) v += 2.0 * n;
  if (a.prox == PX_BOUNDS_VEC) v += 2.0 * M;
  if (a.vsrc) v += M;
  return v * sizeof(T);
}
template <typename T>
void K<T>::yl(hipStream_t s, const Grid& g, const SetArgs<T>& a, double* partials) {
  ObsScope obs(KID_YL, s, yl_bytes<T>(g, a));
  constexpr int VW = sizeof(T) == 8 ? SIPX_F64_VEC : 4;      // four doubles per thread spill registers at 3 waves per SIMD
  if (g.n[0] % 4 == 0)
    hipLaunchKernelGGL((k_yl<T, VW>), dim3(fit_grid(range_len(g) / VW, NB)), dim3(BLOCK), 0, s, g, a, partials);
  else
    hipLaunchKernelGGL((k_yl<T, 1>), dim3(fit_grid(range_len(g), NB)), dim3(BLOCK), 0, s, g, a, partials);
  SIPX_HIP(hipGetLastError());
}

// ||A'(y - y_old)||^2 for difference operators (r_dual, update_y_l.jl:84).
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_adj_norm(Grid G, SetArgs<T> a, double* __restrict__ partials) {
  double acc[1] = {0};
  long long v0, nvec;
  vec_range<V>(G, v0, nvec);
  for (long long vi = v0 + (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long g = vi * V;
    const Coord c = coords(G, g);
    T t[V];
#pragma unroll
    for (int k = 0; k < V; ++k) t[k] = T(0);
    for (int q = 0; q < a.nblk; ++q) {
      const T* wb = a.dy + (long long)q * G.N;
      auto wv = [&](long long e) { return ldv_u<T, V>(wb + e); };
      adj_dir_acc<T, V>(G, g, c, a.dir[q], a.ih[q], t, wv);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) acc[0] += (double)t[k] * (double)t[k];
  }
  block_reduce_store<1>(acc, partials, 0);
}
template <typename T>
void K<T>::adj_norm(hipStream_t s, const Grid& g, const SetArgs<T>& a, double* partials) {
  ObsScope obs(KID_ADJ_NORM, s, (double)a.nblk * (double)range_len(g) * sizeof(T));
  if (g.n[0] % 4 == 0)
    hipLaunchKernelGGL((k_adj_norm<T, 4>), dim3(fit_grid(range_len(g) / 4, NB)), dim3(BLOCK), 0, s, g, a, partials);
  else
    hipLaunchKernelGGL((k_adj_norm<T, 1>), dim3(fit_grid(range_len(g), NB)), dim3(BLOCK), 0, s, g, a, partials);
  SIPX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Plain operator applications (padded output / padded input): s = A x, t = A' v.
struct OpArgs {
  int nblk;
  int dir[3];
};
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_fwd(Grid G, OpArgs o, T ih0, T ih1, T ih2, const T* __restrict__ x,
                                               T* __restrict__ out) {
  const T ihs[3] = {ih0, ih1, ih2};
  const long long nvec = G.N / V;
  for (long long vi = (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long g = vi * V;
    const Coord c = coords(G, g);
    const Vec<T, V> xc = ldv<T, V>(x + g);
    if (o.nblk == 0) {
      stv<T, V>(out + g, xc);
      continue;
    }
    for (int q = 0; q < o.nblk; ++q) {
      T s[V];
      bool valid[V];
      fwd_dir<T, V>(G, x, xc, g, c, o.dir[q], ihs[q], s, valid);
      Vec<T, V> sv;
#pragma unroll
      for (int k = 0; k < V; ++k) sv.v[k] = s[k];
      stv<T, V>(out + (long long)q * G.N + g, sv);
    }
  }
}
template <typename T>
void K<T>::fwd(hipStream_t s, const Grid& g, int nblk, const int* dir, const T* ih, const T* x, T* out) {
  OpArgs o;
  o.nblk = nblk;
  for (int q = 0; q < 3; ++q) o.dir[q] = q < nblk ? dir[q] : 0;
  const T i0 = nblk > 0 ? ih[0] : T(0), i1 = nblk > 1 ? ih[1] : T(0), i2 = nblk > 2 ? ih[2] : T(0);
  if (g.n[0] % 4 == 0)
    hipLaunchKernelGGL((k_fwd<T, 4>), dim3(NB), dim3(BLOCK), 0, s, g, o, i0, i1, i2, x, out);
  else
    hipLaunchKernelGGL((k_fwd<T, 1>), dim3(NB), dim3(BLOCK), 0, s, g, o, i0, i1, i2, x, out);
  SIPX_HIP(hipGetLastError());
}

template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_adj(Grid G, OpArgs o, T ih0, T ih1, T ih2, const T* __restrict__ v,
                                               T* __restrict__ out) {
  const T ihs[3] = {ih0, ih1, ih2};
  const long long nvec = G.N / V;
  for (long long vi = (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long g = vi * V;
    const Coord c = coords(G, g);
    if (o.nblk == 0) {
      stv<T, V>(out + g, ldv<T, V>(v + g));
      continue;
    }
    T t[V];
#pragma unroll
    for (int k = 0; k < V; ++k) t[k] = T(0);
    for (int q = 0; q < o.nblk; ++q) {
      const T* wb = v + (long long)q * G.N;
      auto wv = [&](long long e) { return ldv_u<T, V>(wb + e); };
      adj_dir_acc<T, V>(G, g, c, o.dir[q], ihs[q], t, wv);
    }
    Vec<T, V> tv;
#pragma unroll
    for (int k = 0; k < V; ++k) tv.v[k] = t[k];
    stv<T, V>(out + g, tv);
  }
}
template <typename T>
void K<T>::adj(hipStream_t s, const Grid& g, int nblk, const int* dir, const T* ih, const T* v, T* out) {
  OpArgs o;
  o.nblk = nblk;
  for (int q = 0; q < 3; ++q) o.dir[q] = q < nblk ? dir[q] : 0;
  const T i0 = nblk > 0 ? ih[0] : T(0), i1 = nblk > 1 ? ih[1] : T(0), i2 = nblk > 2 ? ih[2] : T(0);
  if (g.n[0] % 4 == 0)
    hipLaunchKernelGGL((k_adj<T, 4>), dim3(NB), dim3(BLOCK), 0, s, g, o, i0, i1, i2, v, out);
  else
    hipLaunchKernelGGL((k_adj<T, 1>), dim3(NB), dim3(BLOCK), 0, s, g, o, i0, i1, i2, v, out);
  SIPX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// ||P(v) - v||^2 and ||v||^2 over a padded vector (set feasibility, update_y_l.jl:92-94,
// PARSDMM_initialize.jl:98) and the in-place application v = P(v).  Pads (v == 0 by construction,
// flagged through `valid`) are skipped.
template <typename T>
struct ProjArgs {
  int nblk;
  int dir[3];
  int prox;
  T plo, phi;
  const T *lb, *ub;
  const ProjScalars<T>* ps;
};
template <typename T>
__device__ __forceinline__ bool is_valid(const Grid& G, const ProjArgs<T>& a, long long e) {
  if (a.nblk == 0) return true;
  const long long q = e / G.N;
  const Coord c = coords(G, e - q * G.N);
  const int dir = a.dir[q];
  return coord_of(c, dir) < G.n[dir] - 1;
}
template <typename T, int APPLY>
__global__ __launch_bounds__(BLOCK) void k_proj(Grid G, ProjArgs<T> a, long long len, T* __restrict__ v,
                                                double* __restrict__ partials) {
  double acc[2] = {0, 0};
  const ProxCtx<T> pc = make_prox<T>(a.prox, a.plo, a.phi, T(0), a.ps);
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < len; e += (long long)gridDim.x * BLOCK) {
    if (!is_valid<T>(G, a, e)) continue;
    const T x = v[e];
    const T lb = a.prox == PX_BOUNDS_VEC ? a.lb[e] : T(0), ub = a.prox == PX_BOUNDS_VEC ? a.ub[e] : T(0);
    const T p = prox_apply<T>(pc, x, lb, ub, T(0), e);
    if (APPLY) {
      v[e] = p;
    } else {
      const T d = p - x;
      acc[0] += (double)d * (double)d;
      acc[1] += (double)x * (double)x;
    }
  }
  if (!APPLY) block_reduce_store<2>(acc, partials, 0);
}

// ---------------------------------------------------------------------------------------------
// obj / evol_x reductions when no distance-term kernel carries them (feasibility_only).
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_log3(long long N, const T* __restrict__ x, const T* __restrict__ m,
                                                const T* __restrict__ xold, double* __restrict__ partials) {
  double acc[3] = {0, 0, 0};
  const long long nvec = N / V;
  for (long long vi = (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const Vec<T, V> xv = ldv<T, V>(x + vi * V), ov = ldv<T, V>(xold + vi * V);
    const Vec<T, V> mv = m ? ldv<T, V>(m + vi * V) : xv;        // m == nullptr: only the evol_x sums are wanted
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const T d = xv.v[k] - mv.v[k], e = ov.v[k] - xv.v[k];
      acc[0] += (double)d * (double)d;
      acc[1] += (double)e * (double)e;
      acc[2] += (double)xv.v[k] * (double)xv.v[k];
    }
  }
  block_reduce_store<3>(acc, partials, SL_OBJ);
}
template <typename T>
void K<T>::log3(hipStream_t s, long long N, const T* x, const T* m, const T* xold, double* partials) {
  ObsScope obs(KID_LOG3, s, (m ? 3.0 : 2.0) * (double)N * sizeof(T));
  if (N % 4 == 0 && aligned16(x, m, xold))
    hipLaunchKernelGGL((k_log3<T, 4>), dim3(NB), dim3(BLOCK), 0, s, N, x, m, xold, partials);
  else
    hipLaunchKernelGGL((k_log3<T, 1>), dim3(NB), dim3(BLOCK), 0, s, N, x, m, xold, partials);
  SIPX_HIP(hipGetLastError());
}

// Reference row order <-> padded layout of one operator block (difference along `dir`, or dir < 0 for the identity):
// row r = (i, j, k) column-major over the block's own extents (n with n[dir]-1) lives at padded entry i + n1 (j + n2 k).
// PACK: rows[r] = pad[e];  otherwise pad[e] = rows[r] (the pads keep their zeros).
template <typename T, bool PACK>
__global__ __launch_bounds__(BLOCK) void k_rows(Grid G, int dir, long long nrows, T* __restrict__ rows, T* __restrict__ pad) {
  long long d0 = G.n[0], d1 = G.n[1];
  if (dir == 0) d0 -= 1;
  if (dir == 1) d1 -= 1;
  for (long long r = (long long)blockIdx.x * BLOCK + threadIdx.x; r < nrows; r += (long long)gridDim.x * BLOCK) {
    const long long i = r % d0, t = r / d0, j = t % d1, k = t / d1;
    const long long e = i + G.n[0] * (j + G.n[1] * k);
    if (PACK) rows[r] = pad[e];
    else pad[e] = rows[r];
  }
}
template <typename T>
void K<T>::rows_pack(hipStream_t s, const Grid& g, int dir, long long nrows, const T* pad, T* rows) {
  hipLaunchKernelGGL((k_rows<T, true>), dim3(NB), dim3(BLOCK), 0, s, g, dir, nrows, rows, const_cast<T*>(pad));
  SIPX_HIP(hipGetLastError());
}
template <typename T>
void K<T>::rows_unpack(hipStream_t s, const Grid& g, int dir, long long nrows, const T* rows, T* pad) {
  hipLaunchKernelGGL((k_rows<T, false>), dim3(NB), dim3(BLOCK), 0, s, g, dir, nrows, const_cast<T*>(rows), pad);
  SIPX_HIP(hipGetLastError());
}

// ---- caller-supplied sparse operator (constraint.custom_TD_OP): rows have a handful of entries, one thread per row / column.
// The accumulation orders are those of Julia's CSC kernels: A*x adds the products of a row in ascending column order,
// A'*w those of a column in ascending row order (SparseArrays mul!).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_csr_spmv(long long M, const long long* __restrict__ rowptr, const long long* __restrict__ col,
                                                    const T* __restrict__ val, const T* __restrict__ x, T* __restrict__ out) {
  for (long long r = (long long)blockIdx.x * BLOCK + threadIdx.x; r < M; r += (long long)gridDim.x * BLOCK) {
    T acc = T(0);
    for (long long k = rowptr[r]; k < rowptr[r + 1]; ++k) acc = acc + val[k] * x[col[k]];
    out[r] = acc;
  }
}
template <typename T, int MODE>
__global__ __launch_bounds__(BLOCK) void k_csc_adj(long long N, const long long* __restrict__ colptr, const long long* __restrict__ row,
                                                   const T* __restrict__ val, const T* __restrict__ y, const T* __restrict__ l,
                                                   T rho, T* __restrict__ out, int accumulate, double* __restrict__ partials) {
  double acc[1] = {0};
  for (long long j = (long long)blockIdx.x * BLOCK + threadIdx.x; j < N; j += (long long)gridDim.x * BLOCK) {
    T t = T(0);
    for (long long k = colptr[j]; k < colptr[j + 1]; ++k) {
      const long long r = row[k];
      const T w = MODE == 0 ? (rho * y[r] + l[r]) : y[r];        // MODE 1: y carries y - y_old
      t = t + val[k] * w;
    }
    if (MODE == 0) out[j] = accumulate ? out[j] + t : t;
    else acc[0] += (double)t * (double)t;
  }
  if (MODE == 1) block_reduce_store<1>(acc, partials, 0);
}
template <typename T>
void K<T>::csr_spmv(hipStream_t s, long long M, const long long* rowptr, const long long* col, const T* val, const T* x, T* out) {
  hipLaunchKernelGGL((k_csr_spmv<T>), dim3(NB), dim3(BLOCK), 0, s, M, rowptr, col, val, x, out);
  SIPX_HIP(hipGetLastError());
}
template <typename T>
void K<T>::csc_adj_rhs(hipStream_t s, long long N, const long long* colptr, const long long* row, const T* val, const T* y,
                       const T* l, T rho, T* out, int accumulate) {
  hipLaunchKernelGGL((k_csc_adj<T, 0>), dim3(NB), dim3(BLOCK), 0, s, N, colptr, row, val, y, l, rho, out, accumulate, (double*)nullptr);
  SIPX_HIP(hipGetLastError());
}
template <typename T>
void K<T>::csc_adj_norm(hipStream_t s, long long N, const long long* colptr, const long long* row, const T* val, const T* dy,
                        double* partials) {
  hipLaunchKernelGGL((k_csc_adj<T, 1>), dim3(NB), dim3(BLOCK), 0, s, N, colptr, row, val, dy, (const T*)nullptr, T(0), (T*)nullptr, 0,
                     partials);
  SIPX_HIP(hipGetLastError());
}

// w = u + v (Minkowski mode: TD_OP_sum[i] * x = A (u + v))
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_sum_uv(long long N, const T* __restrict__ u, const T* __restrict__ v,
                                                  T* __restrict__ w) {
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < N; e += (long long)gridDim.x * BLOCK) w[e] = u[e] + v[e];
}
template <typename T>
void K<T>::sum_uv(hipStream_t s, long long N, const T* u, const T* v, T* w) {
  hipLaunchKernelGGL((k_sum_uv<T>), dim3(NB), dim3(BLOCK), 0, s, N, u, v, w);
  SIPX_HIP(hipGetLastError());
}

template <typename T>
static ProjArgs<T> make_proj_args(int nblk, const int* dir, int prox, T plo, T phi, const T* lb, const T* ub,
                                  const ProjScalars<T>* ps) {
  ProjArgs<T> a;
  a.nblk = nblk;
  for (int q = 0; q < 3; ++q) a.dir[q] = (dir && q < nblk) ? dir[q] : 0;
  a.prox = prox;
  a.plo = plo;
  a.phi = phi;
  a.lb = lb;
  a.ub = ub;
  a.ps = ps;
  return a;
}

template <typename T>
void proj_dist_grid(hipStream_t s, const Grid& g, int nblk, const int* dir, long long len, const T* v, int prox, T plo,
                    T phi, const T* lb, const T* ub, const ProjScalars<T>* ps, double* partials) {
  ProjArgs<T> a = make_proj_args<T>(nblk, dir, prox, plo, phi, lb, ub, ps);
  hipLaunchKernelGGL((k_proj<T, 0>), dim3(NB), dim3(BLOCK), 0, s, g, a, len, const_cast<T*>(v), partials);
  SIPX_HIP(hipGetLastError());
}
template <typename T>
void proj_apply_grid(hipStream_t s, const Grid& g, int nblk, const int* dir, long long len, T* v, int prox, T plo,
                     T phi, const T* lb, const T* ub, const ProjScalars<T>* ps) {
  ProjArgs<T> a = make_proj_args<T>(nblk, dir, prox, plo, phi, lb, ub, ps);
  hipLaunchKernelGGL((k_proj<T, 1>), dim3(NB), dim3(BLOCK), 0, s, g, a, len, v, (double*)nullptr);
  SIPX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Nearest-neighbour grid transfer of the multilevel scheme: sample position of fine index k (0-based) on an
// axis is 1 + k (nc-1)/(nf-1) (1-based, range(1, stop=nc, length=nf)); BSpline(Constant()) rounds it to the
// nearest grid point.  Exact integer arithmetic, no floating point.
struct RsArgs {
  long long nc[3], nf[3];
};
// nearest grid point of position 1 + k (nc-1)/(nf-1) on 1..nc, half-way positions go up
// (Interpolations.jl 0.13 rounds with floor(x + 1/2) inside the axis), in exact integers
__device__ __forceinline__ long long nn_index(long long k, long long nc, long long nf) {
  if (nf <= 1 || nc <= 1) return 0;
  const long long num = k * (nc - 1), den = nf - 1;
  return (2 * num + den) / (2 * den);
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_resample(RsArgs a, const T* __restrict__ in, T* __restrict__ out) {
  const long long tot = a.nf[0] * a.nf[1] * a.nf[2];
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < tot; e += (long long)gridDim.x * BLOCK) {
    const long long i = e % a.nf[0], jk = e / a.nf[0], j = jk % a.nf[1], k = jk / a.nf[1];
    const long long ci = nn_index(i, a.nc[0], a.nf[0]), cj = nn_index(j, a.nc[1], a.nf[1]), ck = nn_index(k, a.nc[2], a.nf[2]);
    out[e] = in[ci + a.nc[0] * (cj + a.nc[1] * ck)];
  }
}
template <typename T>
void resample_nn(hipStream_t s, const long long* nc, const long long* nf, const T* in, T* out) {
  RsArgs a;
  for (int q = 0; q < 3; ++q) { a.nc[q] = nc[q]; a.nf[q] = nf[q]; }
  hipLaunchKernelGGL((k_resample<T>), dim3(NB), dim3(BLOCK), 0, s, a, in, out);
  SIPX_HIP(hipGetLastError());
}

// The same transfer between two PADDED arrays (every block of a set's vector is laid out over the whole grid: entry (i, j, k) of a
// chunk with dims cf <= nf sits at i + nf0 (j + nf1 k)), for the fine grid points e0 <= g < e1 only: what a rank of a slab-decomposed
// level stores.  Entry by entry the copy resample_nn makes on the chunks in row order (the pads are left alone).
struct RsPadArgs {
  long long nc[3], nf[3];      // the grids
  long long cc[3], cf[3];      // dims of the chunk on either level (the grid's, minus one along the direction of a difference operator)
  long long e0, e1;
};
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_resample_padded(RsPadArgs a, const T* __restrict__ in, T* __restrict__ out) {
  for (long long e = a.e0 + (long long)blockIdx.x * BLOCK + threadIdx.x; e < a.e1; e += (long long)gridDim.x * BLOCK) {
    const long long i = e % a.nf[0], jk = e / a.nf[0], j = jk % a.nf[1], k = jk / a.nf[1];
    if (i >= a.cf[0] || j >= a.cf[1] || k >= a.cf[2]) continue;
    const long long ci = nn_index(i, a.cc[0], a.cf[0]), cj = nn_index(j, a.cc[1], a.cf[1]), ck = nn_index(k, a.cc[2], a.cf[2]);
    out[e] = in[ci + a.nc[0] * (cj + a.nc[1] * ck)];
  }
}
template <typename T>
void resample_nn_padded(hipStream_t s, const long long* nc, const long long* nf, const long long* cc, const long long* cf, long long e0,
                        long long e1, const T* in, T* out) {
  if (e1 <= e0) return;
  RsPadArgs a;
  for (int q = 0; q < 3; ++q) { a.nc[q] = nc[q]; a.nf[q] = nf[q]; a.cc[q] = cc[q]; a.cf[q] = cf[q]; }
  a.e0 = e0; a.e1 = e1;
  hipLaunchKernelGGL((k_resample_padded<T>), dim3(fit_grid(e1 - e0, NB)), dim3(BLOCK), 0, s, a, in, out);
  SIPX_HIP(hipGetLastError());
}

// A set on TV / D2D / D3D: the reference cuts the ROW vector of l, y into a D_x-, a D_y- and a D_z-sized chunk, in that order,
// whatever order the operator's blocks have, and resamples each chunk as an array of its shape (interpolate_y_l.jl:21-30,53-57;
// the engine's blocks are ordered z, y, x, so a chunk straddles blocks).  The same copy, padded blocks to padded block, for the
// grid points e0 <= g < e1 of fine block b: padded entry -> row of the fine vector -> chunk and position in it -> nearest
// position of the coarse chunk -> row of the coarse vector -> coarse block and padded entry.
struct RsRowsArgs {
  long long nc[3], nf[3];
  int nblk, b;
  int dir[3];                  // direction of block q's difference operator (both levels)
  long long cs_c[4], cs_f[4];  // first row of chunk q (q = 0, 1, 2: the D_x-, D_y-, D_z-sized one) on either level
  long long ro_c[4], ro_f[4];  // first row of block q
  long long cstride;           // distance between the coarse blocks in `in`
  long long e0, e1;
};
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_resample_rows(RsRowsArgs a, const T* __restrict__ in, T* __restrict__ out) {
  const int db = a.dir[a.b];
  long long fb[3] = {a.nf[0], a.nf[1], a.nf[2]};
  fb[db] -= 1;
  for (long long e = a.e0 + (long long)blockIdx.x * BLOCK + threadIdx.x; e < a.e1; e += (long long)gridDim.x * BLOCK) {
    const long long i = e % a.nf[0], jk = e / a.nf[0], j = jk % a.nf[1], k = jk / a.nf[1];
    if (i >= fb[0] || j >= fb[1] || k >= fb[2]) continue;                       // (a pad of the block)
    const long long r = a.ro_f[a.b] + i + fb[0] * (j + fb[1] * k);             // row of the fine vector
    int q = 0;
    while (q + 1 < a.nblk && r >= a.cs_f[q + 1]) ++q;                          // its chunk: dims n - e_q
    long long sf[3] = {a.nf[0], a.nf[1], a.nf[2]}, sc[3] = {a.nc[0], a.nc[1], a.nc[2]};
    sf[q] -= 1; sc[q] -= 1;
    const long long lf = r - a.cs_f[q];
    const long long ii = lf % sf[0], jjkk = lf / sf[0], jj = jjkk % sf[1], kk = jjkk / sf[1];
    const long long ci = nn_index(ii, sc[0], sf[0]), cj = nn_index(jj, sc[1], sf[1]), ck = nn_index(kk, sc[2], sf[2]);
    const long long rc = a.cs_c[q] + ci + sc[0] * (cj + sc[1] * ck);          // row of the coarse vector
    int p = 0;
    while (p + 1 < a.nblk && rc >= a.ro_c[p + 1]) ++p;                         // its block: dims n - e_dir[p]
    long long cb[3] = {a.nc[0], a.nc[1], a.nc[2]};
    cb[a.dir[p]] -= 1;
    const long long lc = rc - a.ro_c[p];
    const long long pi = lc % cb[0], pjk = lc / cb[0], pj = pjk % cb[1], pk = pjk / cb[1];
    out[e] = in[(long long)p * a.cstride + pi + a.nc[0] * (pj + a.nc[1] * pk)];
  }
}
template <typename T>
void resample_nn_rows(hipStream_t s, const long long* nc, const long long* nf, int nblk, const int* dir, int b, long long cstride,
                      long long e0, long long e1, const T* in, T* out) {
  if (e1 <= e0) return;
  RsRowsArgs a;
  for (int q = 0; q < 3; ++q) { a.nc[q] = nc[q]; a.nf[q] = nf[q]; a.dir[q] = q < nblk ? dir[q] : 0; }
  a.nblk = nblk; a.b = b; a.cstride = cstride; a.e0 = e0; a.e1 = e1;
  a.cs_c[0] = a.cs_f[0] = a.ro_c[0] = a.ro_f[0] = 0;
  for (int q = 0; q < nblk; ++q) {
    long long sc = 1, sf = 1, bc = 1, bf = 1;
    for (int d = 0; d < 3; ++d) {
      sc *= nc[d] - (d == q ? 1 : 0); sf *= nf[d] - (d == q ? 1 : 0);
      bc *= nc[d] - (d == dir[q] ? 1 : 0); bf *= nf[d] - (d == dir[q] ? 1 : 0);
    }
    a.cs_c[q + 1] = a.cs_c[q] + sc; a.cs_f[q + 1] = a.cs_f[q] + sf;
    a.ro_c[q + 1] = a.ro_c[q] + bc; a.ro_f[q + 1] = a.ro_f[q] + bf;
  }
  hipLaunchKernelGGL((k_resample_rows<T>), dim3(fit_grid(e1 - e0, NB)), dim3(BLOCK), 0, s, a, in, out);
  SIPX_HIP(hipGetLastError());
}

// x = (x*rho + m) / (rho + 1.0) through the same device function k_yl applies for the distance term (prox_l2s!.jl:3-6)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_prox_l2s(long long n, T* __restrict__ x, const T* __restrict__ m, T rho) {
  const ProxCtx<T> pc = make_prox<T>(PX_DIST, T(0), T(0), rho, nullptr);
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < n; e += (long long)gridDim.x * BLOCK)
    x[e] = prox_apply<T>(pc, x[e], T(0), T(0), m[e], e);
}
template <typename T>
void prox_l2s_dev(hipStream_t s, long long n, T* x, const T* m, T rho) {
  hipLaunchKernelGGL((k_prox_l2s<T>), dim3(fit_grid(n, NB)), dim3(BLOCK), 0, s, n, x, m, rho);
  SIPX_HIP(hipGetLastError());
}

#define SIPX_INST(T)                                                                                              \
  template void prox_l2s_dev<T>(hipStream_t, long long, T*, const T*, T);                                         \
  template void resample_nn<T>(hipStream_t, const long long*, const long long*, const T*, T*);                   \
  template void resample_nn_rows<T>(hipStream_t, const long long*, const long long*, int, const int*, int, long long, long long, \
                                    long long, const T*, T*);                                                     \
  template void resample_nn_padded<T>(hipStream_t, const long long*, const long long*, const long long*, const long long*, long long, \
                                      long long, const T*, T*);                                                   \
  template void K<T>::sum_uv(hipStream_t, long long, const T*, const T*, T*);                                      \
  template void K<T>::csr_spmv(hipStream_t, long long, const long long*, const long long*, const T*, const T*, T*);  \
  template void K<T>::csc_adj_rhs(hipStream_t, long long, const long long*, const long long*, const T*, const T*,    \
                                  const T*, T, T*, int);                                                             \
  template void K<T>::csc_adj_norm(hipStream_t, long long, const long long*, const long long*, const T*, const T*,   \
                                   double*);                                                                         \
  template void K<T>::rows_pack(hipStream_t, const Grid&, int, long long, const T*, T*);                           \
  template void K<T>::rows_unpack(hipStream_t, const Grid&, int, long long, const T*, T*);                         \
  template void K<T>::rhs_compose(hipStream_t, const Grid&, const RhsArgs<T>&, T*, int);                         \
  template void K<T>::yl(hipStream_t, const Grid&, const SetArgs<T>&, double*);                                  \
  template void K<T>::adj_norm(hipStream_t, const Grid&, const SetArgs<T>&, double*);                            \
  template void K<T>::fwd(hipStream_t, const Grid&, int, const int*, const T*, const T*, T*);                    \
  template void K<T>::adj(hipStream_t, const Grid&, int, const int*, const T*, const T*, T*);                    \
  template void K<T>::log3(hipStream_t, long long, const T*, const T*, const T*, double*);                       \
  template void proj_dist_grid<T>(hipStream_t, const Grid&, int, const int*, long long, const T*, int, T, T,     \
                                  const T*, const T*, const ProjScalars<T>*, double*);                           \
  template void proj_apply_grid<T>(hipStream_t, const Grid&, int, const int*, long long, T*, int, T, T, const T*, \
                                   const T*, const ProjScalars<T>*);
SIPX_INST(float)
SIPX_INST(double)

}  // namespace sipx
