// k_yl_multi: the y/l update of EVERY set, the residual sums and -- when rho cannot change before the next iteration -- the
// right-hand side of that iteration, in ONE sweep over the grid.
//
// Replaces (reference file:line), on iterations without Barzilai-Borwein sums, snapshots or feasibility estimates:
//   update_y_l for all sets            src/update_y_l.jl:36-88   (one k_yl launch per set before: x was read p times)
//   r_dual = rho ||A'(y - y_old)||     src/update_y_l.jl:82-84   (k_adj_norm: y - y_old was written, then read back)
//   rhs_compose of the NEXT iteration  src/rhs_compose.jl:24-36  (k_rhs: every y_i, l_i was read back right after being written)
// Algorithmic bytes: x, m, x_old read once (3 N), y_i, l_i read and written once (4 M_i each), rhs written (N) --
// 24 N w for the headline list (p = 5, M_i ~ N) against 30 + 3 + 11 = 44 N w of the three separate passes.
//
// The adjoint stencils of the residual norm and of the right-hand side need the NEW values of a set at the grid neighbours
// g - stride.  A workgroup owns a tile of (4 LX) x TY grid points of a plane and marches along the last dimension:
//   -x neighbour: the lane to the left (one shuffle); at a tile edge inside the grid the one point is recomputed;
//   -y neighbour: the thread one row up, through LDS (double buffered by plane parity: one barrier per plane); the row in
//                 front of the tile is recomputed by the tile's first row of threads -- for the blocks that difference along
//                 y only, 1 / TY of one block's work;
//   -z neighbour: the same thread's values of the previous plane, kept in a private LDS slot; the plane in front of a chunk
//                 of planes is recomputed once per chunk (this is also how a rank of a slab-decomposed solve obtains the last
//                 plane of the rank below, bit for bit, without an exchange).
// Recomputation is exact because the prox is element-wise once its scalars (theta, scale, tau) are known.  Every element
// goes through the arithmetic of k_yl / k_rhs / k_adj_norm in the same order (-ffp-contract=off), so y, l and rhs are
// bit-identical to the separate kernels; the float64 sums differ in their summation order only.
#include <stdexcept>
#include <string>

#include "sipx_device.h"

namespace sipx {

constexpr int MULTI_NT = 256;
constexpr int MULTI_YB = 2;       // blocks that difference along the tile's row dimension (LDS exchange slots)
constexpr int MULTI_ZB = 2;       // blocks that difference along the march dimension (private LDS slots)

// s = A x at V consecutive points of a line, then the element-wise update of update_y_l.jl:64-78 (the code of k_yl)
template <typename T, int V>
__device__ __forceinline__ void multi_block_update(const MultiBlk<T>& B, const ProxCtx<T>& pc, const Vec<T, V>& xc, const Vec<T, V>& xn,
                                                   const bool (&valid)[V], const Vec<T, V>& yv, const Vec<T, V>& lv, const Vec<T, V>& lbv,
                                                   const Vec<T, V>& ubv, const Vec<T, V>& mv, long long e, Vec<T, V>& yn, Vec<T, V>& ln,
                                                   T (&s)[V], T (&rp)[V]) {
  const bool relax = !(B.gamma == T(1));
  const T gam = B.gamma, omg = T(1) - B.gamma, nih = -B.ih;
#pragma unroll
  for (int k = 0; k < V; ++k) {
    if (B.dir < 0) {
      s[k] = xc.v[k];
    } else {
      const T d = nih * xc.v[k] + B.ih * xn.v[k];              // fwd_dir: the two products of a CSC row in column order
      s[k] = valid[k] ? d : T(0);
    }
    const T yo = yv.v[k], lo = lv.v[k];
    const T xh = relax ? (gam * s[k] + omg * yo) : s[k];        // update_y_l.jl:72
    const T v = xh - lo * B.rho1;                               // :67 / :74
    T y1 = prox_apply<T>(pc, v, lbv.v[k], ubv.v[k], mv.v[k], e + k);   // :68 / :75
    if (!valid[k]) y1 = T(0);
    rp[k] = y1 - s[k];                                          // :69 / :76
    ln.v[k] = relax ? (lo + B.rho * (y1 - xh)) : (lo + B.rho * rp[k]);   // :70 / :77
    yn.v[k] = y1;
  }
}

template <typename T, int V, int NBLK>
__global__ __launch_bounds__(MULTI_NT) void k_yl_multi(Grid G, MultiArgs<T> a, int lgLX, int tiles_x, int tiles_y, int zchunk,
                                                       long long items, long long jlo, long long jhi) {
  __shared__ T ybuf[2][MULTI_YB][2][V][MULTI_NT];       // [plane parity][y block][w | dy][element][thread]
  __shared__ T zbuf[MULTI_ZB][2][V][MULTI_NT];          // [z block][w | dy][element][thread]: the previous plane, private
  const int tid = threadIdx.x, LX = 1 << lgLX, tx = tid & (LX - 1), ty = tid >> lgLX, TY = MULTI_NT >> lgLX;
  const long long n1 = G.n[0], n2 = G.n[1], n3 = G.n[2], st1 = G.st[1], st2 = G.st[2], N = G.N;
  const bool fuse_rhs = a.rhs != nullptr;

  ProxCtx<T> pc[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b)
    if (b < a.nblk) pc[b] = make_prox<T>(a.b[b].prox, a.b[b].plo, a.b[b].phi, a.b[b].rho, a.b[b].ps);
  double acc_rp[NBLK], acc_du[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) acc_rp[b] = acc_du[b] = 0;
  double acc_obj = 0, acc_evo = 0, acc_xx = 0;

  const long long tiles = (long long)tiles_x * tiles_y;
  for (long long item = blockIdx.x; item < items; item += gridDim.x) {
    const long long zc = item / tiles, tile = item - zc * tiles;
    const int tile_y = (int)(tile / tiles_x), tile_x = (int)(tile - (long long)tile_y * tiles_x);
    const long long i0 = ((long long)tile_x * LX + tx) * V, j = jlo + (long long)tile_y * TY + ty;
    const bool active = i0 < n1 && j < jhi;
    const long long k0 = a.zlo + zc * zchunk, k1 = (k0 + zchunk < a.zhi) ? k0 + zchunk : a.zhi;
    const long long gj = active ? i0 + st1 * j : 0;           // inactive threads shadow point 0 of the plane (nothing stored)
    // masks that do not change along the march
    bool vx[V], mxm[V];                                        // forward-difference row exists / its left neighbour row exists (x)
#pragma unroll
    for (int k = 0; k < V; ++k) { vx[k] = (i0 + k) < n1 - 1; mxm[k] = (i0 + k) > 0; }
    const bool vy = j < n2 - 1, mym = j > 0;
    __syncthreads();                                           // the previous item's LDS traffic is over

    // the previous plane of the blocks that difference along the march dimension (recomputed at the chunk's first plane)
    {
      int zi = 0;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        if (b >= a.nblk || a.b[b].dir != 2) continue;
        const MultiBlk<T>& B = a.b[b];
        Vec<T, V> w = zerov<T, V>(), d = zerov<T, V>();
        if (active && k0 > 0) {
          const long long g = gj + st2 * (k0 - 1);
          const Vec<T, V> xc = ldv<T, V>(a.x + g), xn = ldv_u<T, V>(a.x + g + st2);
          const Vec<T, V> yv = ldv<T, V>(B.y + g), lv = ldv<T, V>(B.l + g);
          Vec<T, V> lbv = zerov<T, V>(), ubv = zerov<T, V>();
          if (B.prox == PX_BOUNDS_VEC) { lbv = ldv<T, V>(B.lb + g); ubv = ldv<T, V>(B.ub + g); }
          bool valid[V];
#pragma unroll
          for (int k = 0; k < V; ++k) valid[k] = true;        // plane k0 - 1 <= n3 - 2
          Vec<T, V> yn, ln;
          T s[V], rp[V];
          multi_block_update<T, V>(B, pc[b], xc, xn, valid, yv, lv, lbv, ubv, zerov<T, V>(), g, yn, ln, s, rp);
#pragma unroll
          for (int k = 0; k < V; ++k) { w.v[k] = B.rho * yn.v[k] + ln.v[k]; d.v[k] = yn.v[k] - yv.v[k]; }
        }
        if (zi < MULTI_ZB) {
#pragma unroll
          for (int k = 0; k < V; ++k) { zbuf[zi][0][k][tid] = w.v[k]; zbuf[zi][1][k][tid] = d.v[k]; }
        }
        ++zi;
      }
    }

    Vec<T, V> xnext = active ? ldv<T, V>(a.x + gj + st2 * k0) : zerov<T, V>();
    for (long long kz = k0; kz < k1; ++kz) {
      const int par = (int)(kz & 1);
      const long long g = gj + st2 * kz;
      const Vec<T, V> xc = xnext;
      const bool vz = kz < n3 - 1, mzm = kz > 0;
      Vec<T, V> xpx = zerov<T, V>(), xpy = zerov<T, V>();
      if (active) {
        xnext = ldv_u<T, V>(a.x + g + st2);                    // x carries an end halo of a plane: unconditional
        xpx = ldv_u<T, V>(a.x + g + 1);
        xpy = ldv_u<T, V>(a.x + g + st1);
      }
      // ---- phase A: the update of every block; w = rho y + l and y - y_old stay in registers -------------------------
      T wv[NBLK][V], dv[NBLK][V];
      int yi = 0;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
#pragma unroll
        for (int k = 0; k < V; ++k) wv[b][k] = dv[b][k] = T(0);
        if (b >= a.nblk) continue;
        const MultiBlk<T>& B = a.b[b];
        if (active) {
          const long long e = g;                                // (block bases are already offset by q N)
          const Vec<T, V> yv = ldv_nt<T, V>(B.y + e), lv = ldv_nt<T, V>(B.l + e);      // last use of the old iterate
          Vec<T, V> lbv = zerov<T, V>(), ubv = zerov<T, V>(), mv = zerov<T, V>();
          if (B.prox == PX_BOUNDS_VEC) { lbv = ldv<T, V>(B.lb + e); ubv = ldv<T, V>(B.ub + e); }
          if (B.dist) mv = ldv<T, V>(a.m + g);
          bool valid[V];
#pragma unroll
          for (int k = 0; k < V; ++k) valid[k] = B.dir < 0 ? true : (B.dir == 0 ? vx[k] : (B.dir == 1 ? vy : vz));
          const Vec<T, V>& xn = B.dir == 0 ? xpx : (B.dir == 1 ? xpy : xnext);
          Vec<T, V> yn, ln;
          T s[V], rp[V];
          multi_block_update<T, V>(B, pc[b], xc, xn, valid, yv, lv, lbv, ubv, mv, B.set >= 0 ? (long long)(B.y - a.b[b].y) + e : e, yn, ln, s, rp);
          stv_nt<T, V>(B.yo + e, yn);
          stv_nt<T, V>(B.lo + e, ln);
#pragma unroll
          for (int k = 0; k < V; ++k) {
            wv[b][k] = B.rho * yn.v[k] + ln.v[k];              // rhs_compose.jl:28-30 on the new iterate
            dv[b][k] = yn.v[k] - yv.v[k];                      // x_hat = y - y_old  update_y_l.jl:82
            acc_rp[b] += (double)rp[k] * (double)rp[k];
            if (B.dir < 0) acc_du[b] += (double)dv[b][k] * (double)dv[b][k];
          }
          if (B.dist) {                                         // PARSDMM.jl:140,145
            const Vec<T, V> xo = ldv<T, V>(a.xold + g);
#pragma unroll
            for (int k = 0; k < V; ++k) {
              const T dd = xc.v[k] - mv.v[k], ev = xo.v[k] - xc.v[k];
              acc_obj += (double)dd * (double)dd;
              acc_evo += (double)ev * (double)ev;
              acc_xx += (double)xc.v[k] * (double)xc.v[k];
            }
          }
        }
        if (B.dir == 1) {
          if (yi < MULTI_YB) {
#pragma unroll
            for (int k = 0; k < V; ++k) { ybuf[par][yi][0][k][tid] = wv[b][k]; ybuf[par][yi][1][k][tid] = dv[b][k]; }
          }
          ++yi;
        }
      }
      __syncthreads();
      // ---- phase B: adjoint stencils on the new values -> r_dual sums and the right-hand side -------------------------
      T out[V], tr[V], td[V];
#pragma unroll
      for (int k = 0; k < V; ++k) out[k] = tr[k] = td[k] = T(0);
      yi = 0;
      int zi = 0;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        if (b >= a.nblk) continue;
        const MultiBlk<T>& B = a.b[b];
        if (B.first) {
#pragma unroll
          for (int k = 0; k < V; ++k) tr[k] = td[k] = T(0);
        }
        if (B.dir < 0) {                                        // identity: t = rho y + l (k_rhs), no neighbour
#pragma unroll
          for (int k = 0; k < V; ++k) tr[k] = wv[b][k];
        } else {
          T pw[V], pd[V];                                       // the values at g - stride
          bool mm[V], mc[V];                                    // that row exists / the row at g exists
          if (B.dir == 0) {
            // the lane to the left holds the point in front of this vector; at the left edge of a tile inside the grid the
            // one point is recomputed
            T lw = __shfl_up(wv[b][V - 1], 1, 64), ld = __shfl_up(dv[b][V - 1], 1, 64);
            if (tx == 0 && active && i0 > 0) {
              const long long e1 = g - 1;
              Vec<T, 1> x1, x2, y1v, l1v, lb1 = zerov<T, 1>(), ub1 = zerov<T, 1>(), yn1, ln1;
              x1.v[0] = a.x[e1]; x2.v[0] = xc.v[0];
              y1v.v[0] = B.y[e1]; l1v.v[0] = B.l[e1];
              if (B.prox == PX_BOUNDS_VEC) { lb1.v[0] = B.lb[e1]; ub1.v[0] = B.ub[e1]; }
              const bool v1[1] = {true};
              T s1[1], rp1[1];
              multi_block_update<T, 1>(B, pc[b], x1, x2, v1, y1v, l1v, lb1, ub1, zerov<T, 1>(), e1, yn1, ln1, s1, rp1);
              lw = B.rho * yn1.v[0] + ln1.v[0];
              ld = yn1.v[0] - y1v.v[0];
            }
#pragma unroll
            for (int k = 0; k < V; ++k) {
              pw[k] = k == 0 ? lw : wv[b][k - 1];
              pd[k] = k == 0 ? ld : dv[b][k - 1];
              mm[k] = mxm[k];
              mc[k] = vx[k];
            }
          } else if (B.dir == 1) {
            if (ty > 0) {
#pragma unroll
              for (int k = 0; k < V; ++k) {
                pw[k] = yi < MULTI_YB ? ybuf[par][yi][0][k][tid - LX] : T(0);
                pd[k] = yi < MULTI_YB ? ybuf[par][yi][1][k][tid - LX] : T(0);
              }
            } else {
#pragma unroll
              for (int k = 0; k < V; ++k) pw[k] = pd[k] = T(0);
              if (active && mym) {                              // the row in front of the tile: recomputed
                const long long e1 = g - st1;
                const Vec<T, V> x1 = ldv<T, V>(a.x + e1), yv = ldv<T, V>(B.y + e1), lv = ldv<T, V>(B.l + e1);
                Vec<T, V> lbv = zerov<T, V>(), ubv = zerov<T, V>(), yn, ln;
                if (B.prox == PX_BOUNDS_VEC) { lbv = ldv<T, V>(B.lb + e1); ubv = ldv<T, V>(B.ub + e1); }
                bool valid[V];
#pragma unroll
                for (int k = 0; k < V; ++k) valid[k] = true;
                T s1[V], rp1[V];
                multi_block_update<T, V>(B, pc[b], x1, xc, valid, yv, lv, lbv, ubv, zerov<T, V>(), e1, yn, ln, s1, rp1);
#pragma unroll
                for (int k = 0; k < V; ++k) { pw[k] = B.rho * yn.v[k] + ln.v[k]; pd[k] = yn.v[k] - yv.v[k]; }
              }
            }
#pragma unroll
            for (int k = 0; k < V; ++k) { mm[k] = mym; mc[k] = vy; }
            ++yi;
          } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
              pw[k] = zi < MULTI_ZB ? zbuf[zi][0][k][tid] : T(0);
              pd[k] = zi < MULTI_ZB ? zbuf[zi][1][k][tid] : T(0);
              mm[k] = mzm;
              mc[k] = vz;
            }
            if (zi < MULTI_ZB) {
#pragma unroll
              for (int k = 0; k < V; ++k) { zbuf[zi][0][k][tid] = wv[b][k]; zbuf[zi][1][k][tid] = dv[b][k]; }
            }
            ++zi;
          }
          // adj_dir_acc: t += ih w[g - st] (if that row exists); t += (-ih) w[g] (if row g exists)
          const T ih = B.ih, nih = -B.ih;
#pragma unroll
          for (int k = 0; k < V; ++k) {
            const T r1 = tr[k] + ih * pw[k];
            tr[k] = mm[k] ? r1 : tr[k];
            const T r2 = tr[k] + nih * wv[b][k];
            tr[k] = mc[k] ? r2 : tr[k];
            const T d1 = td[k] + ih * pd[k];
            td[k] = mm[k] ? d1 : td[k];
            const T d2 = td[k] + nih * dv[b][k];
            td[k] = mc[k] ? d2 : td[k];
          }
        }
        if (B.last) {
#pragma unroll
          for (int k = 0; k < V; ++k) {
            out[k] = out[k] + tr[k];                            // sets added in order (rhs_compose.jl:24-31)
            if (B.dir >= 0 && active) acc_du[b] += (double)td[k] * (double)td[k];
          }
        }
      }
      if (fuse_rhs && active) {
        Vec<T, V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] = out[k];
        stv<T, V>(a.rhs + g, o);
      }
    }
  }
  // ---- sums: blocks of one set fold into the set's first block, then one block-wide reduction per slot ----------------
  // flat slot index into the engine's partial array: set * SET_SLOTS + {SL_RPRI, SL_DY | SL_ADJ}, the distance term's three
  constexpr int K = 2 * NBLK + 3;
  double acc[K];
  int slots[K];
  const int scratch = a.nblk > 0 ? a.b[0].set * SET_SLOTS + SL_HL : SL_HL;      // unused entries: a slot nobody reads on these iterations
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    acc[2 * b] = acc[2 * b + 1] = 0;
    slots[2 * b] = slots[2 * b + 1] = scratch;
  }
  int dist_set = -1;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    if (b >= a.nblk) continue;
    if (a.b[b].dist) dist_set = a.b[b].set;
#pragma unroll
    for (int b2 = 0; b2 < NBLK; ++b2) {          // the set's last block carries r_dual, its first collects r_pri
      if (b2 < a.nblk && a.b[b2].set == a.b[b].set && a.b[b2].first) acc[2 * b2] += acc_rp[b];
    }
    if (a.b[b].last) {
      acc[2 * b + 1] = acc_du[b];
      slots[2 * b + 1] = a.b[b].set * SET_SLOTS + (a.b[b].dir < 0 ? SL_DY : SL_ADJ);
    }
    if (a.b[b].first) slots[2 * b] = a.b[b].set * SET_SLOTS + SL_RPRI;
  }
  acc[2 * NBLK] = acc_obj; acc[2 * NBLK + 1] = acc_evo; acc[2 * NBLK + 2] = acc_xx;
  const int ds = dist_set >= 0 ? dist_set : (a.nblk > 0 ? a.b[0].set : 0);
  slots[2 * NBLK] = ds * SET_SLOTS + (dist_set >= 0 ? SL_OBJ : SL_LH);
  slots[2 * NBLK + 1] = ds * SET_SLOTS + (dist_set >= 0 ? SL_EVO : SL_DL);
  slots[2 * NBLK + 2] = ds * SET_SLOTS + (dist_set >= 0 ? SL_XX : SL_GG);
  // (several entries may name the scratch slot: they are written one after another by different threads of the epilogue --
  // harmless, nobody reads it)
  __shared__ double sm[K][MULTI_NT / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) sm[k][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < MULTI_NT / 64; ++i) s += sm[threadIdx.x][i];
    int slot = slots[0];
#pragma unroll
    for (int k = 1; k < K; ++k) slot = (int)threadIdx.x == k ? slots[k] : slot;
    double* row = a.partials + (long long)slot * NB;
    row[blockIdx.x] = s;
    for (int jj = blockIdx.x + gridDim.x; jj < NB; jj += gridDim.x) row[jj] = 0.0;
  }
}

template <typename T, int V, int NBLK>
static void launch_multi(hipStream_t s, const Grid& g, const MultiArgs<T>& a, double bytes) {
  // tile geometry: LX lanes of V points along x (a power of two, at most a wave), TY = 256 / LX rows
  const long long nvx = g.n[0] / V;
  int lg = 0;
  while ((1 << lg) < nvx && lg < 6) ++lg;
  const int LX = 1 << lg, TY = MULTI_NT / LX;
  const bool three = g.n[2] > 1;
  // rows of the plane this launch covers: all of them, or (2-D slab decomposition) the rank's rows
  long long jlo = 0, jhi = g.n[1], zlo = a.zlo, zhi = a.zhi;
  if (!three) { jlo = a.zlo; jhi = a.zhi; zlo = 0; zhi = 1; }
  if (jhi <= jlo || zhi <= zlo) return;
  const int tiles_x = (int)((nvx + LX - 1) / LX), tiles_y = (int)((jhi - jlo + TY - 1) / TY);
  const long long tiles = (long long)tiles_x * tiles_y;
  // chunks of planes: enough work items to fill the chip several times over, chunks long enough that the plane recomputed in
  // front of each stays a small share (<= 1 / 8 of one block's work)
  const long long planes = zhi - zlo;
  long long want = (4ll * NB_7 + tiles - 1) / tiles;          // chunks per tile column for ~4 items per workgroup slot
  if (want < 1) want = 1;
  long long zchunk = (planes + want - 1) / want;
  if (zchunk < 8) zchunk = planes < 8 ? planes : 8;
  const long long nchunks = (planes + zchunk - 1) / zchunk;
  const long long items = tiles * nchunks;
  const int grid = (int)(items < NB_7 ? items : NB_7);
  MultiArgs<T> b = a;
  b.zlo = zlo; b.zhi = zhi;
  ObsScope obs(KID_YL_MULTI, s, bytes);
  hipLaunchKernelGGL((k_yl_multi<T, V, NBLK>), dim3(grid), dim3(MULTI_NT), 0, s, g, b, lg, tiles_x, tiles_y, (int)zchunk, items, jlo, jhi);
  SIPX_HIP(hipGetLastError());
}

template <typename T>
bool K<T>::yl_multi(hipStream_t s, const Grid& g, const MultiArgs<T>& a) {
  constexpr int V = sizeof(T) == 8 ? 2 : 4;
  if (a.nblk < 1 || a.nblk > MULTI_MAXB || g.n[0] % V != 0) return false;
  int ny = 0, nz = 0;
  for (int b = 0; b < a.nblk; ++b) {
    ny += a.b[b].dir == 1;
    nz += a.b[b].dir == 2;
    if (a.b[b].dir == 2 && g.n[2] <= 1) return false;
  }
  if (ny > MULTI_YB || nz > MULTI_ZB) return false;
  // algorithmic bytes: x, (m, x_old for the distance term) read; y, l of every block read and written; rhs written
  const bool three = g.n[2] > 1;
  const double pts = (double)(a.zhi - a.zlo) * (three ? (double)g.st[2] : (double)g.st[1]);
  double vecs = 1.0 + (a.rhs ? 1.0 : 0.0);
  for (int b = 0; b < a.nblk; ++b) vecs += 4.0 + (a.b[b].dist ? 2.0 : 0.0) + (a.b[b].prox == PX_BOUNDS_VEC ? 2.0 : 0.0);
  const double bytes = vecs * pts * sizeof(T);
  if (a.nblk <= 4) launch_multi<T, V, 4>(s, g, a, bytes);
  else if (a.nblk <= 6) launch_multi<T, V, 6>(s, g, a, bytes);
  else launch_multi<T, V, 8>(s, g, a, bytes);
  return true;
}

template bool K<float>::yl_multi(hipStream_t, const Grid&, const MultiArgs<float>&);
template bool K<double>::yl_multi(hipStream_t, const Grid&, const MultiArgs<double>&);

}  // namespace sipx
