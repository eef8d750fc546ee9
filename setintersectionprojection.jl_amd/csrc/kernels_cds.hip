// CDS (compressed diagonal storage) kernels + the fused CG building blocks.
//
// Replaces (reference file:line):
//   CDS_MVp_MT / CDS_MVp_MT_subfunc  src/CDS_MVp_MT.jl:9-25, src/CDS_MVp_MT_subfunc.jl:6-20
//   Ax_CDS_MT (fill! + MVp)          src/argmin_x.jl:72-78
//   cg loop body                     src/cg.jl:82-115
//   CDS_scaled_add!                  src/CDS_scaled_add!.jl:8-26
// The reference sweeps the output once PER DIAGONAL (4dN words of traffic); here one pass
// reads every band once and writes y once: (d+2)*N*w algorithmic bytes.  Per row the products
// are added band after band in Q_offsets order starting from 0, i.e. exactly the reference's
// accumulation order, and -ffp-contract=off keeps mul and add separate like Julia does.
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "sipx_device.h"
// Workgroups of the kernels that write the CG partials: three per compute unit (768 on MI355X).  Measured against 512 ...
// 1792: whole multiples of the CU count win over the in-between sizes, and 3 per CU is +1 % at 256^3, +4 % at 512^3 over
// the 7 per CU the other stencil kernels use.
#define SIPX_CG_GRID launch_blocks(3)
#ifndef SIPX_F64_VEC
#define SIPX_F64_VEC 2
#endif

namespace sipx {

// acc[k] = sum_b R[r+k, b] * x[r+k+off_b], bands in order, rows clipped like CDS_MVp.jl:17-23.
// BRANCH-FREE: x carries a halo of max|off| elements on both sides (engine allocation), so every band
// is one unconditional band load + one unconditional (element-aligned) x load; rows whose column falls
// outside [0,N) are masked by a select, which keeps the reference's "skip" semantics bit for bit.
// All 2d loads of a row group are independent and issue back to back (measured +30% over the
// bounds-checked version, profiles/r01_spmv_designspace_*.txt).
// How the vector of a product is read at position c (element aligned): a stored array, or -- CG iterations from the second
// on, on grids where it pays -- p_{k+1} = r_{k+1} + beta p_k formed on the fly from the two stored arrays (see k_cds_fused).
template <typename T, int V>
struct LoadStored {
  const T* __restrict__ x;
  __device__ __forceinline__ Vec<T, V> operator()(long long c) const { return ldv_u<T, V>(x + c); }
};
template <typename T, int V>
struct LoadFused {
  const T* __restrict__ r;
  const T* __restrict__ p;
  T beta;
  __device__ __forceinline__ Vec<T, V> operator()(long long c) const {
    const Vec<T, V> rv = ldv_u<T, V>(r + c), pv = ldv_u<T, V>(p + c);
    Vec<T, V> o;
#pragma unroll
    for (int k = 0; k < V; ++k) o.v[k] = rv.v[k] + beta * pv.v[k];      // the arithmetic of k_cg_update_p (cg.jl:114)
    return o;
  }
};

template <typename T, int V, int D, typename XL>
__device__ __forceinline__ void cds_rows(long long N, const T* __restrict__ R, const CdsArgs& a, const XL& xl, long long r,
                                         T (&acc)[V]) {
  const int d = D ? D : a.d;
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = T(0);
  if constexpr (D > 0) {
    Vec<T, V> rv[D], xv[D];
#pragma unroll
    for (int b = 0; b < D; ++b) {
      if (a.sym) {
        // the positive bands are read twice (here and |o| rows further on): ordinary loads keep them in cache; the
        // negative ones are the partner shifted by o (rows r + o < 0 are masked below, the address stays inside Q)
        rv[b] = a.off[b] >= 0 ? ldv<T, V>(R + (long long)b * N + r)
                              : ldv_u<T, V>(R + (long long)a.partner[b] * N + r + a.off[b]);
      } else {
        rv[b] = ldv_nt<T, V>(R + (long long)b * N + r);
      }
      xv[b] = xl(r + a.off[b]);
    }
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const long long c = r + a.off[b];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const T t = acc[k] + rv[b].v[k] * xv[b].v[k];
        acc[k] = (c + k >= 0 && c + k < N) ? t : acc[k];
      }
    }
  } else {
    for (int b = 0; b < d; ++b) {
      const long long c = r + a.off[b];
      const Vec<T, V> rv = !a.sym ? ldv_nt<T, V>(R + (long long)b * N + r)
                                  : (a.off[b] >= 0 ? ldv<T, V>(R + (long long)b * N + r)
                                                   : ldv_u<T, V>(R + (long long)a.partner[b] * N + c));
      const Vec<T, V> xv = xl(c);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const T t = acc[k] + rv.v[k] * xv.v[k];
        acc[k] = (c + k >= 0 && c + k < N) ? t : acc[k];
      }
    }
  }
}

// MODE 0: y = Qx.  MODE 1: Ap = Qp and partial(p.Ap) (cg.jl:85-88).
// MODE 2: r = b - Qx, p = r, x_old = x, partials ||r||^2, ||b||^2 (argmin_x.jl:34 + cg.jl:52-58 + PARSDMM.jl:106).
template <typename T, int V, int D, int MODE>
__global__ __launch_bounds__(BLOCK) void k_cds(long long N, long long r0, long long r1, const T* __restrict__ R, CdsArgs a,
                                               const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ b,
                                               T* __restrict__ pout, T* __restrict__ xold, double* __restrict__ partials,
                                               const int* __restrict__ done) {
  if (MODE == 1 && *done) return;
  // rows [r0, r1) of the N x N matrix (the whole matrix, or the z-slab this rank owns in the sharded x-step): indices and
  // boundary masks stay global, only the sweep is restricted
  const long long nvec = r1 / V;
  double acc0 = 0, acc1 = 0;
  for (long long vi = r0 / V + (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long r = vi * V;
    T s[V];
    cds_rows<T, V, D>(N, R, a, LoadStored<T, V>{x}, r, s);
    if (MODE == 0) {
      Vec<T, V> o;
#pragma unroll
      for (int k = 0; k < V; ++k) o.v[k] = s[k];
      stv_nt<T, V>(y + r, o);
    } else if (MODE == 1) {
      const Vec<T, V> pv = ldv<T, V>(x + r);
      Vec<T, V> o;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        o.v[k] = s[k];
        acc0 += (double)pv.v[k] * (double)s[k];
      }
      stv_nt<T, V>(y + r, o);
    } else {
      const Vec<T, V> bv = ldv<T, V>(b + r);
      const Vec<T, V> xv = ldv<T, V>(x + r);
      Vec<T, V> o;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        o.v[k] = bv.v[k] - s[k];
        acc0 += (double)o.v[k] * (double)o.v[k];
        acc1 += (double)bv.v[k] * (double)bv.v[k];
      }
      stv<T, V>(y + r, o);
      if (pout) stv<T, V>(pout + r, o);
      if (xold) stv<T, V>(xold + r, xv);
    }
  }
  if (MODE == 1) {
    double acc[1] = {acc0};
    block_reduce_store<1>(acc, partials, 0);
  } else if (MODE == 2) {
    double acc[2] = {acc0, acc1};
    block_reduce_store<2>(acc, partials, 0);
  }
}

// ---------------------------------------------------------------------------------------------
// Z-MARCHING product for the 7-band matrix of a 3-D grid (what north_star calls LDS-staged diagonal bands).  A workgroup owns
// a tile of (V LX) x TY grid points of a plane and walks z: x of the planes k-1, k, k+1 and the +n1n2 band of plane k-1 stay
// in registers, the +-1 neighbours come from the lanes next door (shuffles), the +-n1 neighbours and the +n1 band of the row
// above go through LDS (one barrier per plane) -- every band value and every x crosses the fabric ONCE (plus the rows in
// front of / behind a tile and one plane per chunk of planes), where k_cds relies on L2 / Infinity Cache for the shifted
// re-reads (1.32 x the minimum at 512^3).  Products are added in the band order of Q_offsets with k_cds's masks, so the result
// is bit-identical (tested, and compared bit for bit in tools/spmv_bench.hip: 512^3 670 -> 577 us, 256^3 80 -> 70 us).
// role of a band: 0 diag, 1: -1, 2: +1, 3: -n1, 4: +n1, 5: -n1n2, 6: +n1n2
constexpr int march_role(int ord, int b) {
  constexpr int o1[7] = {0, 1, 2, 3, 4, 5, 6}, o2[7] = {0, 5, 3, 1, 2, 4, 6};
  return ord == 1 ? o1[b] : o2[b];
}
// value of (A'A)[g, g+o] for a difference operator (identity when nblk == 0), accumulated in ascending row order of A
template <typename T>
__device__ __forceinline__ T ata_value(const Grid& G, int nblk, const int* dir, const T* ihs, long long o, const Coord& c) {
  T val = (nblk == 0 && o == 0) ? T(1) : T(0);   // identity: AtA = I (precompute_distribute.jl:44-45)
  for (int q = 0; q < nblk; ++q) {
    const int d = dir[q];
    const T ih = ihs[q], nih = -ih;
    const int cc = coord_of(c, d);
    const long long st = G.st[d];
    if (o == 0) {
      if (cc > 0) val = val + ih * ih;              // row g-st holds +ih in column g
      if (cc < G.n[d] - 1) val = val + nih * nih;   // row g holds -ih in column g
    } else if (o == st) {
      if (cc < G.n[d] - 1) val = val + nih * ih;    // row g: A[g,g]*A[g,g+st]
    } else if (o == -st) {
      if (cc > 0) val = val + ih * nih;             // row g-st: A[.,g]*A[.,g-st]
    }
  }
  return val;
}

// What the residual product needs to apply a pending Q update on the fly (MODE 4 of the march): the changed sets with their
// rho differences (the arithmetic of k_q_update, band values regenerated from the operator descriptors) and where the updated
// bands go -- a second copy of Q: the tile in front, the row above and the plane below are still read from the old one.
// (the plan of the update: see k_q_update_plan below for what it holds and why its products are the bits of k_q_update's)
constexpr int QP_MAXB = 9, QP_MAXT = 8, QP_TAB = 160;
struct QPlanTerm {
  int kind;          // 0: one value; 1: diagonal of a difference set -- table over the classes of its nblk directions, index
                     // sum_q class(dir[q]) 3^q; 2 / 3: band +stride / -stride of direction dir[0]: (row exists ? tab[0] : tab[1])
  int nblk;
  int dir[3];
  int tab;           // where its values start in QPlan::tab
};
template <typename T>
struct QPlan {
  int nbands;
  int col[QP_MAXB], nterms[QP_MAXB];
  QPlanTerm t[QP_MAXB][QP_MAXT];
  int ntab;
  T tab[QP_TAB];
};

template <typename T>
struct QUpd {
  QPlan<T> plan;  // alpha_i * (A_i'A_i)[g, g + o] per boundary class of g, the changed sets in order (Q_update!.jl:45-48)
  int pb[4];      // the plan's band for the stored bands with offsets 0, +1, +n1, +n1 n2 (-1: not touched by this update)
  T* qn[4];       // the four stored bands of the updated matrix
  Grid G;
};
struct NoExtra {};
// value of stored band `role` at the point with coordinates c after the pending update; tab: the plan's products (LDS copy)
template <typename T>
__device__ __forceinline__ T q_upd_val(const QUpd<T>& u, const T* __restrict__ tab, int role, const Coord& c, T qv) {
  const int b = u.pb[role];
  if (b < 0) return qv;
  const int cx = c.i == 0 ? 0 : (c.i == (int)u.G.n[0] - 1 ? 2 : 1), cy = c.j == 0 ? 0 : (c.j == (int)u.G.n[1] - 1 ? 2 : 1),
            cz = c.k == 0 ? 0 : (c.k == (int)u.G.n[2] - 1 ? 2 : 1);
  for (int ti = 0; ti < u.plan.nterms[b]; ++ti) {
    const QPlanTerm& t = u.plan.t[b][ti];
    if (t.kind == 0) {
      qv = qv + tab[t.tab];
    } else if (t.kind == 1) {
      int idx = 0, m3 = 1;
      for (int q = 0; q < t.nblk; ++q) {
        const int d = t.dir[q];
        idx += (d == 0 ? cx : (d == 1 ? cy : cz)) * m3;
        m3 *= 3;
      }
      qv = qv + tab[t.tab + idx];
    } else {
      const int d = t.dir[0], cl = d == 0 ? cx : (d == 1 ? cy : cz);
      qv = qv + (cl != (t.kind == 2 ? 2 : 0) ? tab[t.tab] : tab[t.tab + 1]);
    }
  }
  return qv;
}
template <typename T>
__device__ __forceinline__ T q_upd_val(const NoExtra&, const T*, int, const Coord&, T qv) { return qv; }
template <typename T>
__device__ __forceinline__ void load_plan_tab(const QUpd<T>& u, T* tab) {
  for (int i = threadIdx.x; i < u.plan.ntab; i += blockDim.x) tab[i] = u.plan.tab[i];
}
template <typename T>
__device__ __forceinline__ void load_plan_tab(const NoExtra&, T*) {}
template <typename T>
__device__ __forceinline__ const Grid& extra_grid(const QUpd<T>& u) { return u.G; }
__device__ __forceinline__ Grid extra_grid(const NoExtra&) { return Grid{}; }
template <typename T, int V>
__device__ __forceinline__ void store_bands(const QUpd<T>& u, long long r, const Vec<T, V>& r0, const Vec<T, V>& r1, const Vec<T, V>& r2,
                                            const Vec<T, V>& r3) {
  stv<T, V>(u.qn[0] + r, r0);
  stv<T, V>(u.qn[1] + r, r1);
  stv<T, V>(u.qn[2] + r, r2);
  stv<T, V>(u.qn[3] + r, r3);
}
template <typename T, int V>
__device__ __forceinline__ void store_bands(const NoExtra&, long long, const Vec<T, V>&, const Vec<T, V>&, const Vec<T, V>&, const Vec<T, V>&) {}
__device__ __forceinline__ void publish_ticket(unsigned long long* ticket, unsigned seq, int iter, int done);      // (defined with the CG scalar steps below)
constexpr int MARCH_NT = 512;
template <typename T, int V, int ORD, int MODE, typename X>
__global__ __launch_bounds__(MARCH_NT) void k_cds_march(long long n1, long long n2, long long n3, long long z0, long long z1, const T* __restrict__ R0,
                                                        const T* __restrict__ R1, const T* __restrict__ R2, const T* __restrict__ R3,
                                                        const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ b,
                                                        T* __restrict__ pout, T* __restrict__ xold, double* __restrict__ partials,
                                                        const int* __restrict__ done, int lgLX, int tiles_x, int tiles_y, int zchunk,
                                                        long long items, CgState<T>* __restrict__ st, CgState<T>* __restrict__ host,
                                                        unsigned long long* ticket, X extra) {
  constexpr bool UPD = MODE == 4;         // MODE 4 = MODE 2 (residual form) with the pending Q update applied on the fly
  if (MODE == 1 && *done) return;
  // MODE 3 (fused CG iteration, see k_cds_fused): the scalar step of iteration k (resvec, stop test, beta) and the product of
  // iteration k + 1 on p_{k+1} = r_{k+1} + beta p_k, formed wherever it is loaded (x = r_{k+1}, b = p_k; same arithmetic as
  // k_cg_update_p) and stored once into pout by the thread that owns the point; 8 N w instead of the 9 of product + p-update
  T beta = T(0);
  if (MODE == 3) {
    if (st->done) return;
    const double ss = block_sum_partials_n<MARCH_NT>(partials + NB);
    const T rr = (T)ss;
    const T res = (T)sqrt(ss) / st->nr0;      // cg.jl:100
    const bool conv = res <= st->tol;         // cg.jl:104-106
    beta = rr / st->gamma;                    // cg.jl:110
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->ss = ss;
      st->res_last = res;
      if (conv) { st->flag = 0; st->done = 1; } else { st->beta = beta; }
      st->rr = rr;
      *host = *st;
      publish_ticket(ticket, st->seq, st->iters, conv ? 1 : 0);
    }
    if (conv) return;
  }
  auto ldx = [&](long long at) -> Vec<T, V> {          // V entries of the vector the product is taken on
    Vec<T, V> v = ldv<T, V>(x + at);
    if (MODE == 3) {
      const Vec<T, V> pv = ldv<T, V>(b + at);
#pragma unroll
      for (int k = 0; k < V; ++k) v.v[k] = v.v[k] + beta * pv.v[k];
    }
    return v;
  };
  auto ldx1 = [&](long long at) -> T { return MODE == 3 ? x[at] + beta * b[at] : x[at]; };
  __shared__ T sx[2][V][MARCH_NT], sr[2][V][MARCH_NT];
  __shared__ T qtab[UPD ? QP_TAB : 1];
  if (UPD) {
    load_plan_tab<T>(extra, qtab);
    __syncthreads();
  }
  const int tid = threadIdx.x, LX = 1 << lgLX, tx = tid & (LX - 1), ty = tid >> lgLX, TY = MARCH_NT >> lgLX;
  const long long st1 = n1, st2 = n1 * n2, N = st2 * n3;
  const long long tiles = (long long)tiles_x * tiles_y;
  double acc0 = 0, acc1 = 0;
  for (long long item = blockIdx.x; item < items; item += gridDim.x) {
    const long long zc = item / tiles, tile = item - zc * tiles;
    const int tile_y = (int)(tile / tiles_x), tile_x = (int)(tile - (long long)tile_y * tiles_x);
    const long long i0 = ((long long)tile_x * LX + tx) * V, j = (long long)tile_y * TY + ty;
    const bool active = i0 < n1 && j < n2;
    const long long k0 = z0 + zc * zchunk, k1 = (k0 + zchunk < z1) ? k0 + zchunk : z1;
    const unsigned go = active ? (unsigned)(i0 + st1 * j) : 0u;
    __syncthreads();
    Vec<T, V> xm = zerov<T, V>(), x0 = zerov<T, V>(), rzm = zerov<T, V>();
    if (active) {
      xm = ldx(st2 * (k0 - 1) + go);                            // (x carries a halo of a plane on both sides)
      x0 = ldx(st2 * k0 + go);
      if (k0 > 0) {
        rzm = ldv<T, V>(R3 + st2 * (k0 - 1) + go);
        if (UPD) {
#pragma unroll
          for (int k = 0; k < V; ++k) rzm.v[k] = q_upd_val<T>(extra, qtab, 3, Coord{(int)(i0 + k), (int)j, (int)(k0 - 1)}, rzm.v[k]);
        }
      }
    }
    for (long long kz = k0; kz < k1; ++kz) {
      const int par = (int)(kz & 1);
      const long long pz = st2 * kz;
      Vec<T, V> xp = zerov<T, V>(), r0 = xp, r1 = xp, r2 = xp, r3 = xp, bv = xp;
      if (active) {
        xp = ldx(pz + st2 + go);
        r0 = ldv_nt<T, V>(R0 + pz + go);
        r1 = ldv_nt<T, V>(R1 + pz + go);
        r2 = ldv_nt<T, V>(R2 + pz + go);
        r3 = ldv_nt<T, V>(R3 + pz + go);
        if (MODE == 2 || MODE == 4) bv = ldv<T, V>(b + pz + go);
        if (UPD) {
#pragma unroll
          for (int k = 0; k < V; ++k) {
            const Coord ck{(int)(i0 + k), (int)j, (int)kz};
            r0.v[k] = q_upd_val<T>(extra, qtab, 0, ck, r0.v[k]);
            r1.v[k] = q_upd_val<T>(extra, qtab, 1, ck, r1.v[k]);
            r2.v[k] = q_upd_val<T>(extra, qtab, 2, ck, r2.v[k]);
            r3.v[k] = q_upd_val<T>(extra, qtab, 3, ck, r3.v[k]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < V; ++k) { sx[par][k][tid] = x0.v[k]; sr[par][k][tid] = r2.v[k]; }
      // the points next door along x: lanes, or (tile edge / wave edge) one element from memory
      T xl = __shfl_up(x0.v[V - 1], 1, 64), xr = __shfl_down(x0.v[0], 1, 64), rl = __shfl_up(r1.v[V - 1], 1, 64);
      const long long r = pz + go;               // row of element 0
      if (active) {
        if (tx == 0 || (tid & 63) == 0) {
          xl = ldx1(r - 1);
          rl = r > 0 ? R1[r - 1] : T(0);
          if (UPD && r > 0) rl = q_upd_val<T>(extra, qtab, 1, coords(extra_grid(extra), r - 1), rl);     // (still the old value in memory)
        }
        if (tx == LX - 1 || (tid & 63) == 63) xr = ldx1(r + V);
      }
      __syncthreads();
      if (active) {
        Vec<T, V> xu = zerov<T, V>(), xd = xu, ru = xu;    // x of the row above (j - 1) / below (j + 1), +n1 band of the row above
        if (ty > 0) {
#pragma unroll
          for (int k = 0; k < V; ++k) { xu.v[k] = sx[par][k][tid - LX]; ru.v[k] = sr[par][k][tid - LX]; }
        } else {
          xu = ldx(r - st1);
          if (r - st1 >= 0) {
            ru = ldv<T, V>(R2 + r - st1);
            if (UPD) {
#pragma unroll
              for (int k = 0; k < V; ++k) ru.v[k] = q_upd_val<T>(extra, qtab, 2, coords(extra_grid(extra), r - st1 + k), ru.v[k]);
            }
          }
        }
        if (ty < TY - 1 && j + 1 < n2) {
#pragma unroll
          for (int k = 0; k < V; ++k) xd.v[k] = sx[par][k][tid + LX];
        } else {
          xd = ldx(r + st1);
        }
        Vec<T, V> o4;
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const long long rr = r + k;
          // band value and vector entry of each role
          const T rv[7] = {r0.v[k], k == 0 ? rl : r1.v[k - 1], r1.v[k], ru.v[k], r2.v[k], rzm.v[k], r3.v[k]};
          const T xv[7] = {x0.v[k], k == 0 ? xl : x0.v[k - 1], k == V - 1 ? xr : x0.v[k + 1], xu.v[k], xd.v[k], xm.v[k], xp.v[k]};
          const long long co[7] = {0, -1, 1, -st1, st1, -st2, st2};
          T acc = T(0);
#pragma unroll
          for (int q = 0; q < 7; ++q) {
            const int role = march_role(ORD, q);
            const T t = acc + rv[role] * xv[role];
            const long long c = rr + co[role];
            acc = (c >= 0 && c < N) ? t : acc;
          }
          o4.v[k] = acc;
        }
        if (MODE == 0) {
          stv_nt<T, V>(y + r, o4);
        } else if (MODE == 1 || MODE == 3) {
#pragma unroll
          for (int k = 0; k < V; ++k) acc0 += (double)x0.v[k] * (double)o4.v[k];
          stv_nt<T, V>(y + r, o4);
          if (MODE == 3) stv<T, V>(pout + r, x0);
        } else {
          Vec<T, V> o;
#pragma unroll
          for (int k = 0; k < V; ++k) {
            o.v[k] = bv.v[k] - o4.v[k];
            acc0 += (double)o.v[k] * (double)o.v[k];
            acc1 += (double)bv.v[k] * (double)bv.v[k];
          }
          stv<T, V>(y + r, o);
          if (pout) stv<T, V>(pout + r, o);
          if (xold) stv<T, V>(xold + r, x0);
          if (UPD) store_bands<T, V>(extra, r, r0, r1, r2, r3);
        }
      }
      xm = x0; x0 = xp; rzm = r3;
    }
  }
  if (MODE != 0) {
    // block-wide sums of 512 threads into the partial slots (block_reduce_store is written for 256)
    __shared__ double sm[2][MARCH_NT / 64];
    const double v0 = wave_sum(acc0), v1 = wave_sum(acc1);
    __syncthreads();
    if ((tid & 63) == 0) { sm[0][tid >> 6] = v0; sm[1][tid >> 6] = v1; }
    __syncthreads();
    if (tid < ((MODE == 2 || MODE == 4) ? 2 : 1)) {
      double s = 0;
#pragma unroll
      for (int i = 0; i < MARCH_NT / 64; ++i) s += sm[tid][i];
      double* row = partials + (long long)tid * NB;
      row[blockIdx.x] = s;
      for (int jj = blockIdx.x + gridDim.x; jj < NB; jj += gridDim.x) row[jj] = 0.0;
    }
  }
}

// the march applies when the engine marked the matrix (CdsArgs::march) and the row range is whole planes; small grids, where
// a launch would not fill the chip, keep k_cds
template <typename T, int MODE>
static bool try_march(hipStream_t s, long long N, long long r0, long long r1, const T* R, const CdsArgs& a, const T* x, T* y, const T* b,
                      T* pout, T* xold, double* partials, const int* done, CgState<T>* st = nullptr, CgState<T>* host = nullptr,
                      unsigned long long* ticket = nullptr, const QUpd<T>* upd = nullptr) {
  // SIPX_CDS_MARCH=0: never; =2: also on grids too small to fill the chip that way, in chunks of SIPX_CDS_MARCH_ZCHUNK planes (tests)
  // (one copy for all instantiations, refreshed whenever a context is finalised: env_knobs)
  const int sw = env_knobs().cds_march;
  const long long zc_env = env_knobs().cds_march_zchunk;
  if (sw == 0 || !a.march || !a.sym || a.d != 7) return false;
  constexpr int V = sizeof(T) == 8 ? 2 : 4;
  const long long n1 = a.gn[0], n2 = a.gn[1], n3 = a.gn[2], st2 = n1 * n2;
  if (n1 % V != 0 || n1 * n2 * n3 != N || r0 % st2 != 0 || r1 % st2 != 0 || r1 <= r0) return false;
  const long long nvx = n1 / V;
  int lg = 0;
  while ((1 << lg) < nvx && lg < 6) ++lg;
  const int LX = 1 << lg, TY = MARCH_NT / LX;
  const int tiles_x = (int)((nvx + LX - 1) / LX), tiles_y = (int)((n2 + TY - 1) / TY);
  const long long tiles = (long long)tiles_x * tiles_y, planes = (r1 - r0) / st2;
  long long zchunk = planes * tiles / 2048;       // aim at ~2048 work items, chunks of at least 16 planes (measured)
  if (zchunk < 16) zchunk = 16;
  if (sw == 2 && zc_env > 0) zchunk = zc_env;
  if (zchunk > planes) zchunk = planes;
  const long long nchunks = (planes + zchunk - 1) / zchunk, items = tiles * nchunks;
  if (items < 384 && sw != 2) return false;
  const int grid = (int)(items < NB ? items : NB);
  const T *R0 = R + (long long)a.mb[0] * N, *R1 = R + (long long)a.mb[1] * N, *R2 = R + (long long)a.mb[2] * N, *R3 = R + (long long)a.mb[3] * N;
#define SIPX_MARCH(ORD, XT, XV)                                                                                                      \
  hipLaunchKernelGGL((k_cds_march<T, V, ORD, MODE, XT>), dim3(grid), dim3(MARCH_NT), 0, s, n1, n2, n3, r0 / st2, r1 / st2, R0, R1, R2, R3, x, y, b, pout, \
                     xold, partials, done, lg, tiles_x, tiles_y, (int)zchunk, items, st, host, ticket, XV)
  if constexpr (MODE == 4) {
    if (a.march == 1) SIPX_MARCH(1, QUpd<T>, *upd);
    else SIPX_MARCH(2, QUpd<T>, *upd);
  } else {
    if (a.march == 1) SIPX_MARCH(1, NoExtra, NoExtra{});
    else SIPX_MARCH(2, NoExtra, NoExtra{});
  }
#undef SIPX_MARCH
  return true;
}

template <typename T, int MODE>
static void launch_cds(hipStream_t s, long long N, long long r0, long long r1, const T* R, const CdsArgs& a, const T* x, T* y,
                       const T* b, T* pout, T* xold, double* partials, const int* done) {
  if (a.d < 1 || a.d > MAXD) throw std::runtime_error("cds: band count out of range");
  if (r0 < 0 || r1 > N || r0 > r1) throw std::runtime_error("cds: row range outside the matrix");
  // algorithmic bytes: SURVEY 8(d) counts all d bands + the vector read + the result written ((d+2) rows w; the residual form
  // also reads b and writes x_old: +2); what has to move: only the bands with a non-negative offset under the symmetric read
  int read_bands = a.d;
  if (a.sym) { read_bands = 0; for (int b = 0; b < a.d; ++b) read_bands += a.off[b] >= 0 ? 1 : 0; }
  const double rw = (double)(r1 - r0) * sizeof(T), extra = MODE == 2 ? 4.0 : 2.0;
  const double moved_extra = (MODE == 2 && !xold) ? 3.0 : extra;      // (x_old kept by buffer rotation: not written)
  ObsScope obs(MODE == 0 ? KID_CDS_SPMV : (MODE == 1 ? KID_CDS_DOT : KID_CDS_RESID), s, (a.d + extra) * rw, (read_bands + moved_extra) * rw);
  if (try_march<T, MODE>(s, N, r0, r1, R, a, x, y, b, pout, xold, partials, done)) {
    SIPX_HIP(hipGetLastError());
    return;
  }
#define SIPX_CDS(V, D) \
  hipLaunchKernelGGL((k_cds<T, V, D, MODE>), dim3(fit_grid((r1 - r0) / V, SIPX_CG_GRID)), dim3(BLOCK), 0, s, N, r0, r1, R, a, x, y, b, pout, xold, partials, done)
  // 16 bytes per thread and band: four floats or two doubles (four doubles leave the 7-band kernel 3 waves per SIMD)
  constexpr int VW = sizeof(T) == 8 ? SIPX_F64_VEC : 4;
  if (N % 4 == 0 && r0 % 4 == 0 && r1 % 4 == 0) {
    switch (a.d) {
      case 1: SIPX_CDS(VW, 1); break;
      case 3: SIPX_CDS(VW, 3); break;
      case 5: SIPX_CDS(VW, 5); break;
      case 7: SIPX_CDS(VW, 7); break;
      default: SIPX_CDS(VW, 0); break;
    }
  } else {
    SIPX_CDS(1, 0);
  }
#undef SIPX_CDS
  SIPX_HIP(hipGetLastError());
}

template <typename T>
void K<T>::spmv(hipStream_t s, const Grid&, long long N, const T* R, const CdsArgs& a, const T* x, T* y) {
  launch_cds<T, 0>(s, N, 0, N, R, a, x, y, nullptr, nullptr, nullptr, nullptr, nullptr);
}
template <typename T>
void K<T>::spmv_dot(hipStream_t s, long long N, long long r0, long long r1, const T* R, const CdsArgs& a, const T* p, T* Ap,
                    double* partials, const CgState<T>* st) {
  launch_cds<T, 1>(s, N, r0, r1, R, a, p, Ap, nullptr, nullptr, nullptr, partials, &st->done);
}
template <typename T>
void K<T>::resid(hipStream_t s, long long N, long long r0, long long r1, const T* R, const CdsArgs& a, const T* x, const T* b,
                 T* r, T* p, T* xold, double* partials) {
  launch_cds<T, 2>(s, N, r0, r1, R, a, x, r, b, p, xold, partials, nullptr);
}

// ---------------------------------------------------------------------------------------------
// Stencil form of Q = sum_i rho_i A_i'A_i for descriptor-generated sets (identity and forward differences):
// (Qx)_c = w0 x_c + sum_dir w_dir [ (c_dir > 0)(x_c - x_{c-st}) + (c_dir < n_dir-1)(x_c - x_{c+st}) ],
// w0 = sum of rho over identity sets, w_dir = sum of rho_i / h_dir^2 over the sets that difference along dir.
// Reads x only: 2 N w algorithmic bytes instead of (d+2) N w, and Q_update! becomes four scalars.
// NOT the reference's CDS arithmetic (coefficients are not rounded band by band, no update history):
// a separate mode (sipx_set_q_mode), agreement with the CDS mode is at rounding level, not bit for bit.
template <typename T, int V>
__device__ __forceinline__ void stencil_rows(const Grid& G, const StencilQ<T>& q, const T* __restrict__ x, long long g,
                                             const Vec<T, V>& xc, T (&acc)[V]) {
  const Coord c = coords(G, g);
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = q.w0 * xc.v[k];
#pragma unroll
  for (int dir = 0; dir < 3; ++dir) {
    if (!(q.mask & (1 << dir))) continue;
    const long long st = G.st[dir];
    const int nd = (int)G.n[dir], cd = coord_of(c, dir);
    const Vec<T, V> xm = ldv_u<T, V>(x + g - st), xp = ldv_u<T, V>(x + g + st);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int ck = cd + (dir == 0 ? k : 0);
      const T a = xc.v[k] - xm.v[k], b = xc.v[k] - xp.v[k];
      const T t = (ck > 0 ? a : T(0)) + (ck < nd - 1 ? b : T(0));
      acc[k] = acc[k] + q.w[dir] * t;
    }
  }
}

template <typename T, int V, int MODE>
__global__ __launch_bounds__(BLOCK) void k_sq(Grid G, StencilQ<T> q, const T* __restrict__ x, T* __restrict__ y,
                                              const T* __restrict__ b, T* __restrict__ pout, T* __restrict__ xold,
                                              double* __restrict__ partials, const int* __restrict__ done) {
  if (MODE == 1 && *done) return;
  const long long nvec = G.N / V;
  double acc0 = 0, acc1 = 0;
  for (long long vi = (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long r = vi * V;
    const Vec<T, V> xv = ldv<T, V>(x + r);
    T s[V];
    stencil_rows<T, V>(G, q, x, r, xv, s);
    Vec<T, V> o;
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < V; ++k) o.v[k] = s[k];
      stv_nt<T, V>(y + r, o);
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        o.v[k] = s[k];
        acc0 += (double)xv.v[k] * (double)s[k];
      }
      stv_nt<T, V>(y + r, o);
    } else {
      const Vec<T, V> bv = ldv<T, V>(b + r);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        o.v[k] = bv.v[k] - s[k];
        acc0 += (double)o.v[k] * (double)o.v[k];
        acc1 += (double)bv.v[k] * (double)bv.v[k];
      }
      stv<T, V>(y + r, o);
      if (pout) stv<T, V>(pout + r, o);
      if (xold) stv<T, V>(xold + r, xv);
    }
  }
  if (MODE == 1) {
    double acc[1] = {acc0};
    block_reduce_store<1>(acc, partials, 0);
  } else if (MODE == 2) {
    double acc[2] = {acc0, acc1};
    block_reduce_store<2>(acc, partials, 0);
  }
}

template <typename T, int MODE>
static void launch_sq(hipStream_t s, const Grid& G, const StencilQ<T>& q, const T* x, T* y, const T* b, T* pout, T* xold,
                      double* partials, const int* done) {
  ObsScope obs(MODE == 0 ? KID_SQ_SPMV : (MODE == 1 ? KID_SQ_DOT : KID_SQ_RESID), s, (MODE == 2 ? 4.0 : 2.0) * (double)G.N * sizeof(T));
  // the stencil form streams 2N values and wants the larger grid (43 us at 3 workgroups per CU, 32 us at 7; 256^3);
  // block_reduce_store clears the partial entries beyond a launch's grid, so it may differ from cg_update_xr's
  if (G.N % 4 == 0 && G.n[0] % 4 == 0)
    hipLaunchKernelGGL((k_sq<T, 4, MODE>), dim3(NB_7), dim3(BLOCK), 0, s, G, q, x, y, b, pout, xold, partials, done);
  else
    hipLaunchKernelGGL((k_sq<T, 1, MODE>), dim3(NB_7), dim3(BLOCK), 0, s, G, q, x, y, b, pout, xold, partials, done);
  SIPX_HIP(hipGetLastError());
}
template <typename T>
void K<T>::sq_spmv(hipStream_t s, const Grid& G, const StencilQ<T>& q, const T* x, T* y) {
  launch_sq<T, 0>(s, G, q, x, y, nullptr, nullptr, nullptr, nullptr, nullptr);
}
template <typename T>
void K<T>::sq_spmv_dot(hipStream_t s, const Grid& G, const StencilQ<T>& q, const T* p, T* Ap, double* partials,
                       const CgState<T>* st) {
  launch_sq<T, 1>(s, G, q, p, Ap, nullptr, nullptr, nullptr, partials, &st->done);
}
template <typename T>
void K<T>::sq_resid(hipStream_t s, const Grid& G, const StencilQ<T>& q, const T* x, const T* b, T* r, T* p, T* xold,
                    double* partials) {
  launch_sq<T, 2>(s, G, q, x, r, b, p, xold, partials, nullptr);
}

// ---------------------------------------------------------------------------------------------
// Q[:,c] += alpha * AtA_i[:,k]   (CDS_scaled_add!.jl:16-22; one band per launch)
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_q_axpy(long long N, T* __restrict__ q, const T* __restrict__ a, T alpha) {
  const long long nvec = N / V;
  for (long long vi = (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    Vec<T, V> qv = ldv<T, V>(q + vi * V);
    const Vec<T, V> av = ldv<T, V>(a + vi * V);
#pragma unroll
    for (int k = 0; k < V; ++k) qv.v[k] = qv.v[k] + alpha * av.v[k];
    stv<T, V>(q + vi * V, qv);
  }
}
template <typename T>
void K<T>::q_axpy(hipStream_t s, long long N, T* Qband, const T* Aband, T alpha) {
  if (N % 4 == 0)
    hipLaunchKernelGGL((k_q_axpy<T, 4>), dim3(NB), dim3(BLOCK), 0, s, N, Qband, Aband, alpha);
  else
    hipLaunchKernelGGL((k_q_axpy<T, 1>), dim3(NB), dim3(BLOCK), 0, s, N, Qband, Aband, alpha);
  SIPX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// AtA = A'A bands of a difference operator straight from the descriptor, in the accumulation
// order of Julia's sparse product (ascending row of A): what mat2CDS(TD_OP'*TD_OP) yields
// (PARSDMM_precompute_distribute.jl:44-59, mat2CDS.jl:7-32).
struct GenArgs {
  int nblk;
  int dir[3];
  int nband;
  long long off[7];
};
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gen_ata(Grid G, GenArgs a, T ih0, T ih1, T ih2, T* __restrict__ R) {
  const T ihs[3] = {ih0, ih1, ih2};
  for (long long g = (long long)blockIdx.x * BLOCK + threadIdx.x; g < G.N; g += (long long)gridDim.x * BLOCK) {
    const Coord c = coords(G, g);
    for (int b = 0; b < a.nband; ++b) R[(long long)b * G.N + g] = ata_value<T>(G, a.nblk, a.dir, ihs, a.off[b], c);
  }
}

// Fused Q update: every band of Q is read and written once; the changed sets are applied in order
// (Q_update!.jl:45-48), their band values regenerated on the fly unless explicit bands were supplied.
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_q_update(Grid G, long long r0, long long r1, CdsArgs q, QArgs<T> a, T* __restrict__ Q) {
  const long long nvec = r1 / V;      // rows [r0, r1): the whole of Q, or the rows this rank's slab of the x-step reads
  for (long long vi = r0 / V + (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long g = vi * V;
    Coord c[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { c[k] = coords(G, g); c[k].i += k; }
    for (int b = 0; b < q.d; ++b) {
      const long long o = q.off[b];
      if (q.sym && o < 0) continue;            // never read by the SpMV: rebuilt from the partner band on demand (k_mirror_bands)
      bool touched = false;
      Vec<T, V> qv;
      for (int si = 0; si < a.nsets; ++si) {
        const QSet<T>& S = a.s[si];
        int j = -1;
        for (int t = 0; t < S.nband; ++t) if (S.off[t] == o) j = t;
        if (j < 0) continue;
        if (!touched) { qv = ldv<T, V>(Q + (long long)b * G.N + g); touched = true; }
        if (S.ata) {
          const Vec<T, V> av = ldv_nt<T, V>(S.ata + (long long)j * G.N + g);
#pragma unroll
          for (int k = 0; k < V; ++k) qv.v[k] = qv.v[k] + S.alpha * av.v[k];
        } else {
#pragma unroll
          for (int k = 0; k < V; ++k) qv.v[k] = qv.v[k] + S.alpha * ata_value<T>(G, S.nblk, S.dir, S.ih, o, c[k]);
        }
      }
      if (touched) stv<T, V>(Q + (long long)b * G.N + g, qv);
    }
  }
}

// ---- the same update from a PLAN (round 4) ------------------------------------------------------------------------------------
// k_q_update regenerates every band value per element -- loops over sets, a search for the band among the set's offsets, the
// boundary tests of ata_value for every block -- and is bound by that arithmetic, not by the 2 N w per touched band it moves
// (512^3, all sets changed: 1.47 ms for 8 N w = 2.9 TB/s).  But the contribution of a set to a band takes only a handful of
// values: alpha_i * (A_i'A_i)[g, g + o] depends on g only through the CLASS of its coordinates along the set's difference
// directions (first / interior / last point of the line).  The host forms those products ONCE per launch, with the arithmetic
// of ata_value and the multiplication by alpha in TF (this translation unit is compiled with -ffp-contract=off like the kernels),
// and the kernel adds them in set order: Q[:, col] = Q[:, col] + p_i(class(g)) -- the same numbers added in the same order as
// k_q_update and CDS_scaled_add!.jl:16-22, so Q stays bit-identical (the tests compare the engine's Q bit for bit with mat2CDS / CDS_scaled_add! restated on the CPU).
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_q_update_plan(Grid G, long long r0, long long r1, QPlan<T> p, T* __restrict__ Q) {
  __shared__ T tab[QP_TAB];
  for (int i = threadIdx.x; i < p.ntab; i += BLOCK) tab[i] = p.tab[i];
  __syncthreads();
  const int n1 = (int)G.n[0], n2 = (int)G.n[1], n3 = (int)G.n[2];
  const long long nvec = r1 / V;
  for (long long vi = r0 / V + (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long g = vi * V;
    const Coord c = coords(G, g);
    // class of a coordinate: 0 first, 1 interior, 2 last point of its line (a line of one point is never differenced)
    int cx[V];
#pragma unroll
    for (int k = 0; k < V; ++k) cx[k] = (c.i + k) == 0 ? 0 : ((c.i + k) == n1 - 1 ? 2 : 1);
    const int cy = c.j == 0 ? 0 : (c.j == n2 - 1 ? 2 : 1), cz = c.k == 0 ? 0 : (c.k == n3 - 1 ? 2 : 1);
    for (int b = 0; b < p.nbands; ++b) {
      T* row = Q + (long long)p.col[b] * G.N + g;
      Vec<T, V> qv = ldv<T, V>(row);
      for (int ti = 0; ti < p.nterms[b]; ++ti) {
        const QPlanTerm& t = p.t[b][ti];
        if (t.kind == 0) {
          const T v = tab[t.tab];
#pragma unroll
          for (int k = 0; k < V; ++k) qv.v[k] = qv.v[k] + v;
        } else if (t.kind == 1) {
          int base = 0, mulx = 0, m3 = 1;            // index = base + mulx * class_x: only x varies inside the vector
          for (int q = 0; q < t.nblk; ++q) {
            const int d = t.dir[q];
            if (d == 0) mulx += m3; else base += (d == 1 ? cy : cz) * m3;
            m3 *= 3;
          }
#pragma unroll
          for (int k = 0; k < V; ++k) qv.v[k] = qv.v[k] + tab[t.tab + base + mulx * cx[k]];
        } else {
          const int d = t.dir[0];
          const int edge = t.kind == 2 ? 2 : 0;      // +stride: the row exists unless this is the LAST point; -stride: unless the FIRST
          const T on = tab[t.tab], off = tab[t.tab + 1];
#pragma unroll
          for (int k = 0; k < V; ++k) {
            const int cl = d == 0 ? cx[k] : (d == 1 ? cy : cz);
            qv.v[k] = qv.v[k] + (cl != edge ? on : off);
          }
        }
      }
      stv<T, V>(row, qv);
    }
  }
}

// builds the plan; false: the general kernel has to do it (explicit AtA bands, more bands / sets than the plan holds, two blocks
// of one set with the same stride)
template <typename T>
static bool make_q_plan(const Grid& g, const CdsArgs& q, const QArgs<T>& a, QPlan<T>& p) {
  p.nbands = 0;
  p.ntab = 0;
  for (int si = 0; si < a.nsets; ++si) {
    const QSet<T>& S = a.s[si];
    if (S.ata) return false;
    for (int u = 0; u < S.nblk; ++u)
      for (int w = u + 1; w < S.nblk; ++w)
        if (g.st[S.dir[u]] == g.st[S.dir[w]]) return false;
  }
  for (int b = 0; b < q.d; ++b) {
    const long long o = q.off[b];
    if (q.sym && o < 0) continue;            // never read by the SpMV: rebuilt from the partner band on demand (k_mirror_bands)
    int nt = 0;
    QPlanTerm terms[QP_MAXT];
    for (int si = 0; si < a.nsets; ++si) {
      const QSet<T>& S = a.s[si];
      bool has = false;
      for (int t = 0; t < S.nband; ++t) has |= S.off[t] == o;
      if (!has) continue;
      if (nt == QP_MAXT) return false;
      QPlanTerm& t = terms[nt++];
      t.nblk = S.nblk;
      for (int u = 0; u < 3; ++u) t.dir[u] = S.dir[u];
      t.tab = p.ntab;
      if (o == 0 && S.nblk == 0) {             // identity: AtA = I
        t.kind = 0;
        if (p.ntab + 1 > QP_TAB) return false;
        p.tab[p.ntab++] = S.alpha * T(1);
      } else if (o == 0) {                     // diagonal of a difference set: 3^nblk classes, ata_value's sums in block order
        t.kind = 1;
        int ncls = 1;
        for (int u = 0; u < S.nblk; ++u) ncls *= 3;
        if (p.ntab + ncls > QP_TAB) return false;
        for (int idx = 0; idx < ncls; ++idx) {
          T val = T(0);
          int rem = idx;
          for (int u = 0; u < S.nblk; ++u) {
            const int cl = rem % 3;
            rem /= 3;
            const T ih = S.ih[u], nih = -ih;
            if (cl != 0) val = val + ih * ih;          // cc > 0: row g - st holds +ih in column g
            if (cl != 2) val = val + nih * nih;        // cc < n - 1: row g holds -ih in column g
          }
          p.tab[p.ntab++] = S.alpha * val;
        }
      } else {                                 // band +-stride of exactly one of the set's blocks
        int u = -1;
        for (int w = 0; w < S.nblk; ++w)
          if (g.st[S.dir[w]] == (o > 0 ? o : -o)) u = w;
        if (p.ntab + 2 > QP_TAB) return false;
        const T ih = u >= 0 ? S.ih[u] : T(0), nih = -ih;
        t.kind = o > 0 ? 2 : 3;
        t.dir[0] = u >= 0 ? S.dir[u] : 0;
        // (a listed offset that no block of the set has: ata_value is zero everywhere -- both entries alpha * 0)
        const T val_on = u >= 0 ? (o > 0 ? T(0) + nih * ih : T(0) + ih * nih) : T(0);
        p.tab[p.ntab++] = S.alpha * val_on;
        p.tab[p.ntab++] = S.alpha * T(0);
      }
    }
    if (nt == 0) continue;
    if (p.nbands == QP_MAXB) return false;
    p.col[p.nbands] = b;
    p.nterms[p.nbands] = nt;
    for (int t = 0; t < nt; ++t) p.t[p.nbands][t] = terms[t];
    p.nbands += 1;
  }
  return true;
}

template <typename T>
void K<T>::q_update(hipStream_t s, const Grid& g, long long r0, long long r1, const CdsArgs& q, const QArgs<T>& a, T* Q) {
  if (a.nsets == 0 || r1 <= r0) return;
  if (r0 < 0 || r1 > g.N) throw std::runtime_error("q_update: row range outside the matrix");
  double touched = 0, survey = 0;        // bands of Q read and written once; SURVEY 8(d): B_Q = sum over changed sets of 3 d_i N w
  for (int b = 0; b < q.d; ++b) {
    if (q.sym && q.off[b] < 0) continue;
    bool hit = false;
    for (int si = 0; si < a.nsets; ++si)
      for (int t = 0; t < a.s[si].nband; ++t) {
        if (a.s[si].off[t] != q.off[b]) continue;
        hit = true;
        if (a.s[si].ata) touched += 1;
      }
    touched += hit ? 2 : 0;
  }
  for (int si = 0; si < a.nsets; ++si) survey += 3.0 * a.s[si].nband;
  ObsScope obs(KID_Q_UPDATE, s, survey * (double)(r1 - r0) * sizeof(T), touched * (double)(r1 - r0) * sizeof(T));
  QPlan<T> plan;
  if (env_knobs().q_plan && make_q_plan<T>(g, q, a, plan)) {      // (SIPX_Q_PLAN=0: A/B switch, tests)
    if (plan.nbands == 0) return;
    if (g.n[0] % 4 == 0 && r0 % 4 == 0 && r1 % 4 == 0)
      hipLaunchKernelGGL((k_q_update_plan<T, 4>), dim3(fit_grid((r1 - r0) / 4, NB)), dim3(BLOCK), 0, s, g, r0, r1, plan, Q);
    else
      hipLaunchKernelGGL((k_q_update_plan<T, 1>), dim3(fit_grid(r1 - r0, NB)), dim3(BLOCK), 0, s, g, r0, r1, plan, Q);
    SIPX_HIP(hipGetLastError());
    return;
  }
  if (g.n[0] % 4 == 0 && r0 % 4 == 0 && r1 % 4 == 0)
    hipLaunchKernelGGL((k_q_update<T, 4>), dim3(fit_grid((r1 - r0) / 4, NB)), dim3(BLOCK), 0, s, g, r0, r1, q, a, Q);
  else
    hipLaunchKernelGGL((k_q_update<T, 1>), dim3(fit_grid(r1 - r0, NB)), dim3(BLOCK), 0, s, g, r0, r1, q, a, Q);
  SIPX_HIP(hipGetLastError());
}

// Negative bands of the symmetric Q from their partners: Q[r, r+o] = Q[r+o, r] for o < 0, zero where r+o < 0
// (mat2CDS.jl:24-28).  Only needed when the bands themselves are handed out (sipx_get_Q).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_mirror_bands(long long N, CdsArgs q, T* __restrict__ Q) {
  for (long long r = (long long)blockIdx.x * BLOCK + threadIdx.x; r < N; r += (long long)gridDim.x * BLOCK) {
    for (int b = 0; b < q.d; ++b) {
      const long long o = q.off[b];
      if (o >= 0) continue;
      Q[(long long)b * N + r] = (r + o >= 0) ? Q[(long long)q.partner[b] * N + r + o] : T(0);
    }
  }
}
template <typename T>
void K<T>::mirror_bands(hipStream_t s, long long N, const CdsArgs& q, T* Q) {
  hipLaunchKernelGGL((k_mirror_bands<T>), dim3(NB), dim3(BLOCK), 0, s, N, q, Q);
  SIPX_HIP(hipGetLastError());
}

// Minkowski mode: Q is 2N x d.  Row g = (block row br, local point gl), band offset O -> column (bc, cl); the entry of
// set i is (A_i'A_i)[gl, cl] when block (br, bc) of its AtA is populated ([B 0;0 0], [0 0;0 B] or [B B;B B]).
// Q[:, b] += alpha_i * AtA_i[:, b] for every listed set (assembly: alpha = rho; update: alpha = delta rho).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_q_update_mk(Grid G, CdsArgs q, MkArgs<T> a, T* __restrict__ Q) {
  const long long N = G.N, Nx = 2 * G.N;
  for (long long g = (long long)blockIdx.x * BLOCK + threadIdx.x; g < Nx; g += (long long)gridDim.x * BLOCK) {
    const int br = g >= N;
    const long long gl = g - (long long)br * N;
    const Coord c = coords(G, gl);
    for (int b = 0; b < q.d; ++b) {
      if (q.sym && q.off[b] < 0) continue;
      const long long cc = g + q.off[b];
      if (cc < 0 || cc >= Nx) continue;
      const int bc = cc >= N;
      const long long o = (cc - (long long)bc * N) - gl;
      T qv = Q[(long long)b * Nx + g];
      for (int si = 0; si < a.nsets; ++si) {
        const MkSet<T>& S = a.s[si];
        const bool pop = S.comp == 3 || (S.comp == 1 && br == 0 && bc == 0) || (S.comp == 2 && br == 1 && bc == 1);
        if (!pop) continue;
        qv = qv + S.alpha * ata_value<T>(G, S.nblk, S.dir, S.ih, o, c);
      }
      Q[(long long)b * Nx + g] = qv;
    }
  }
}
template <typename T>
void K<T>::q_update_mk(hipStream_t s, const Grid& g, const CdsArgs& q, const MkArgs<T>& a, T* Q) {
  if (a.nsets == 0) return;
  hipLaunchKernelGGL((k_q_update_mk<T>), dim3(NB), dim3(BLOCK), 0, s, g, q, a, Q);
  SIPX_HIP(hipGetLastError());
}

template <typename T>
void K<T>::gen_ata(hipStream_t s, const Grid& g, int nblk, const int* dir, const T* ih, int nband, const long long* offs,
                   T* R) {
  GenArgs a;
  a.nblk = nblk;
  a.nband = nband;
  for (int q = 0; q < 3; ++q) a.dir[q] = q < nblk ? dir[q] : 0;
  for (int b = 0; b < nband; ++b) a.off[b] = offs[b];
  hipLaunchKernelGGL((k_gen_ata<T>), dim3(NB), dim3(BLOCK), 0, s, g, a, ih[0], nblk > 1 ? ih[1] : T(0),
                     nblk > 2 ? ih[2] : T(0), R);
  SIPX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// CG scalar steps: one 256-thread block sums the block partials in fixed order, thread 0 applies
// the reference's scalar logic and mirrors the state into pinned host memory.
// The host decides whether another CG iteration is enqueued from ONE 8-byte word in pinned memory, written with a
// system-scope release store as soon as the verdict of an iteration is known (workgroup 0 of k_cg_update_p, before its
// streaming loop): (seq << 32) | (iter << 1) | done.  The next product is then queued while this kernel is still
// streaming, and no launch is ever made for an iteration that does not run (rocprofv3's per-kernel averages hold work only).
__device__ __forceinline__ void publish_ticket(unsigned long long* ticket, unsigned seq, int iter, int done) {
  const unsigned long long v = ((unsigned long long)seq << 32) | ((unsigned long long)(unsigned)iter << 1) | (done ? 1ull : 0ull);
  __hip_atomic_store(ticket, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_cg_begin(const double* __restrict__ partials, CgState<T>* st,
                                                    CgState<T>* host, int it_outer, T tol_ref, unsigned seq,
                                                    unsigned long long* ticket) {
  const double ss_r = block_sum_partials(partials);
  const double ss_b = block_sum_partials(partials + NB);
  if (threadIdx.x == 0) {
    const T nr0 = (T)sqrt(ss_b), nres = (T)sqrt(ss_r);
    // argmin_x.jl:33-37 -- the 0.1 factor is a Float64 literal; x_solve_tol_ref arrives as a kernel argument
    const double cand = jl_max(0.1 * (double)nres / (double)nr0, (double)(T(10) * eps_of<T>()));
    const T tol = (it_outer < 3) ? (T)cand : (T)jl_min(cand, (double)tol_ref);
    st->tol = tol;
    st->tol_ref = tol;
    st->nr0 = nr0;
    st->ss = ss_r;
    st->rr = (T)ss_r;
    st->gamma = st->alpha = st->beta = T(0);
    st->res_last = T(0);
    st->iters = 0;
    st->flag = -1;
    st->done = 0;
    st->it_outer = it_outer;
    st->seq = seq;
    if (nr0 == T(0)) {            // cg.jl:51  -> x = zeros, flag -9, iter 0
      st->flag = -9;
      st->done = 1;
    } else if (nres / nr0 <= tol) {  // cg.jl:73-76 -> flag 0, iter 1, relres 0
      st->flag = 0;
      st->done = 1;
      st->iters = 1;
    }
    *host = *st;
    publish_ticket(ticket, seq, 0, st->done);
  }
}
template <typename T>
void K<T>::cg_begin(hipStream_t s, double* partials, CgState<T>* st, CgState<T>* host, int it_outer, T tol_ref, unsigned seq,
                    unsigned long long* ticket) {
  ObsScope obs(KID_CG_BEGIN, s, 0.0);
  hipLaunchKernelGGL((k_cg_begin<T>), dim3(1), dim3(BLOCK), 0, s, partials, st, host, it_outer, tol_ref, seq, ticket);
  SIPX_HIP(hipGetLastError());
}

// The scalar steps of an iteration are folded into the vector kernels that consume them: every workgroup sums the
// block partials of the dot product itself (same fixed order, so all of them and the former one-workgroup kernel
// get the same bits), and workgroup 0 records the state.  Two launches and two dependent round trips fewer per iteration.
//
// alpha = dot(r,z) / dot(p,Ap) ; x += alpha p ; r -= alpha Ap ; partial ||r||^2 (slot 1)   (cg.jl:83-100)
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_cg_update_xr(long long N, const T* x_in, T* x, const T* r_in, T* r,
                                                        const T* p, const T* __restrict__ Ap,
                                                        double* __restrict__ partials, CgState<T>* __restrict__ st,
                                                        CgState<T>* __restrict__ host, int iter,
                                                        unsigned long long* ticket, long long hlo, long long hhi) {
  if (st->done) return;
  const double pAp = block_sum_partials(partials);
  const T gamma = st->rr;                   // dot(r,z), cg.jl:86 (not written by this kernel)
  const T alpha = gamma / (T)pAp;           // cg.jl:88
  const bool bad = (isinf(alpha) && alpha > T(0)) || alpha < T(0);   // alpha==Inf || alpha<0, cg.jl:91-93
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->iters = iter;                       // lastIter, cg.jl:83
    st->gamma = gamma;
    st->alpha = alpha;
    if (bad) {
      st->flag = -2;
      st->done = 1;
      st->res_last = T(0);
      *host = *st;
      publish_ticket(ticket, st->seq, iter, 1);
    }
  }
  if (bad) return;
  const long long nvec = N / V;
  double acc[1] = {0};
  for (long long vi = (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    // x_in: the iterate the solve started from on the first iteration (it stays behind untouched as x_old: the engine's x
    // buffers take turns, PARSDMM.jl:128's copy is never made), x itself afterwards
    Vec<T, V> xv = ldv<T, V>(x_in + vi * V), rv = ldv<T, V>(r_in + vi * V);     // r_in == p on the first iteration (p_1 = r_0)
    const Vec<T, V> pv = ldv<T, V>(p + vi * V), av = ldv_nt<T, V>(Ap + vi * V);     // Ap: written and read once per iteration
#pragma unroll
    for (int k = 0; k < V; ++k) {
      xv.v[k] = xv.v[k] + alpha * pv.v[k];
      rv.v[k] = rv.v[k] - alpha * av.v[k];
      acc[0] += (double)rv.v[k] * (double)rv.v[k];
    }
    stv<T, V>(x + vi * V, xv);
    stv<T, V>(r + vi * V, rv);
  }
  // sharded: the planes of x next to this rank's rows follow along (the neighbour's x += alpha p on the copies of x and p held
  // here: same alpha, same operands, same bits), so that x never has to be exchanged after the solve
  for (long long j = (long long)blockIdx.x * BLOCK + threadIdx.x; j < hlo + hhi; j += (long long)gridDim.x * BLOCK) {
    const long long i = j < hlo ? j - hlo : N + (j - hlo);
    x[i] = x_in[i] + alpha * p[i];
  }
  block_reduce_store<1>(acc, partials, 1);    // its own slot: other workgroups may still be reading slot 0
}
template <typename T>
void K<T>::cg_update_xr(hipStream_t s, long long N, const T* x_in, T* x, const T* r_in, T* r, const T* p, const T* Ap, double* partials,
                        CgState<T>* st, CgState<T>* host, int iter, unsigned long long* ticket, long long hlo, long long hhi) {
  ObsScope obs(KID_CG_XR, s, 6.0 * (double)N * sizeof(T));        // x, r, p, Ap read; x, r written
  if (N % 4 == 0 && aligned16(x, r_in, r, p, Ap) && aligned16(x_in, x))
    hipLaunchKernelGGL((k_cg_update_xr<T, 4>), dim3(fit_grid(N / 4, SIPX_CG_GRID)), dim3(BLOCK), 0, s, N, x_in, x, r_in, r, p, Ap, partials, st, host, iter, ticket, hlo, hhi);
  else
    hipLaunchKernelGGL((k_cg_update_xr<T, 1>), dim3(fit_grid(N, SIPX_CG_GRID)), dim3(BLOCK), 0, s, N, x_in, x, r_in, r, p, Ap, partials, st, host, iter, ticket, hlo, hhi);
  SIPX_HIP(hipGetLastError());
}

// resvec[iter] = ||r|| / nr0, stop test, beta = dot(z,r) / gamma ; p = r + beta p   (cg.jl:100-114)
template <typename T, int V>
__global__ __launch_bounds__(BLOCK) void k_cg_update_p(long long N, T* __restrict__ p, const T* __restrict__ r,
                                                       const double* __restrict__ partials, CgState<T>* __restrict__ st,
                                                       CgState<T>* __restrict__ host, unsigned long long* ticket,
                                                       long long hlo, long long hhi) {
  // `done` may be raised by workgroup 0 of this very launch: a workgroup that starts late and sees it returns, which is
  // what it would have decided from the partials anyway
  if (st->done) return;
  const double ss = block_sum_partials(partials + NB);
  const T rr = (T)ss;
  const T res = (T)sqrt(ss) / st->nr0;      // cg.jl:100
  const bool conv = res <= st->tol;         // cg.jl:104-106
  const T beta = rr / st->gamma;            // cg.jl:110 (gamma: written by the previous kernel)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->ss = ss;
    st->res_last = res;
    if (conv) {
      st->flag = 0;
      st->done = 1;
    } else {
      st->beta = beta;
    }
    st->rr = rr;
    *host = *st;
    publish_ticket(ticket, st->seq, st->iters, conv ? 1 : 0);
  }
  if (conv) return;
  const long long nvec = N / V;
  for (long long vi = (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    Vec<T, V> pv = ldv<T, V>(p + vi * V);
    const Vec<T, V> rv = ldv<T, V>(r + vi * V);
#pragma unroll
    for (int k = 0; k < V; ++k) pv.v[k] = rv.v[k] + beta * pv.v[k];
    stv<T, V>(p + vi * V, pv);
  }
  // sharded: the neighbours' boundary planes of p are formed here from their planes of r (received) and the copy of p_k
  for (long long j = (long long)blockIdx.x * BLOCK + threadIdx.x; j < hlo + hhi; j += (long long)gridDim.x * BLOCK) {
    const long long i = j < hlo ? j - hlo : N + (j - hlo);
    p[i] = r[i] + beta * p[i];
  }
}
template <typename T>
void K<T>::cg_update_p(hipStream_t s, long long N, T* p, const T* r, const double* partials, CgState<T>* st,
                       CgState<T>* host, unsigned long long* ticket, long long hlo, long long hhi) {
  ObsScope obs(KID_CG_P, s, 3.0 * (double)N * sizeof(T));         // r, p read; p written
  if (N % 4 == 0 && aligned16(p, r))
    hipLaunchKernelGGL((k_cg_update_p<T, 4>), dim3(fit_grid(N / 4, NB)), dim3(BLOCK), 0, s, N, p, r, partials, st, host, ticket, hlo, hhi);
  else
    hipLaunchKernelGGL((k_cg_update_p<T, 1>), dim3(fit_grid(N, NB)), dim3(BLOCK), 0, s, N, p, r, partials, st, host, ticket, hlo, hhi);
  SIPX_HIP(hipGetLastError());
}

// CG iterations from the second on, fused (grids where it pays, one rank): the scalar step that k_cg_update_p opens with
// (resvec, stop test, beta; cg.jl:100-110) followed by the NEXT product taken directly on p_{k+1} = r_{k+1} + beta p_k, which
// is formed on the fly wherever a band needs it (same arithmetic as k_cg_update_p, so the same bits) and stored once, by the
// thread that owns the row.  One launch and the 3 N w bytes of the p-update less per iteration (the product reads r and p_k
// instead of p_{k+1}: + 1 N w, and writes p_{k+1}: + 1 N w), and the kernel is queued right behind the x / r update without
// waiting for the host: when the iteration it belongs to does not run it returns at once, as k_cg_update_p does when CG has
// converged.  p_k and p_{k+1} live in two arrays that take turns (a neighbour may still need p_k after this row's p_{k+1} is out).
// The 3 d loads of a row group cost registers (154 VGPRs at d = 5: three waves per SIMD; capping them at 128 / 96 spills and
// loses what the fusion gains), so the kernel itself is no faster than product + p-update; what it saves is the launch and
// the round trip: 2048^2 1990 -> 2057 it/s, 256^3 unchanged (there the unfused form stays the default).
template <typename T, int V, int D>
__global__ __launch_bounds__(BLOCK) void k_cds_fused(long long N, const T* __restrict__ R, CdsArgs a, const T* __restrict__ r,
                                                     const T* __restrict__ p_old, T* __restrict__ p_new, T* __restrict__ Ap,
                                                     double* __restrict__ partials, CgState<T>* __restrict__ st,
                                                     CgState<T>* __restrict__ host, unsigned long long* ticket) {
  if (st->done) return;
  const double ss = block_sum_partials(partials + NB);
  const T rr = (T)ss;
  const T res = (T)sqrt(ss) / st->nr0;      // cg.jl:100
  const bool conv = res <= st->tol;         // cg.jl:104-106
  const T beta = rr / st->gamma;            // cg.jl:110 (gamma: written by the x / r update)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->ss = ss;
    st->res_last = res;
    if (conv) {
      st->flag = 0;
      st->done = 1;
    } else {
      st->beta = beta;
    }
    st->rr = rr;
    *host = *st;
    publish_ticket(ticket, st->seq, st->iters, conv ? 1 : 0);
  }
  if (conv) return;
  const LoadFused<T, V> xl{r, p_old, beta};
  const long long nvec = N / V;
  double acc0 = 0;
  for (long long vi = (long long)blockIdx.x * BLOCK + threadIdx.x; vi < nvec; vi += (long long)gridDim.x * BLOCK) {
    const long long row = vi * V;
    T s[V];
    cds_rows<T, V, D>(N, R, a, xl, row, s);
    const Vec<T, V> pv = xl(row);
    Vec<T, V> o;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      o.v[k] = s[k];
      acc0 += (double)pv.v[k] * (double)s[k];
    }
    stv<T, V>(p_new + row, pv);
    stv_nt<T, V>(Ap + row, o);
  }
  double acc[1] = {acc0};
  block_reduce_store<1>(acc, partials, 0);
}
template <typename T>
void K<T>::spmv_fused(hipStream_t s, long long N, const T* R, const CdsArgs& a, const T* r, const T* p_old, T* p_new, T* Ap,
                      double* partials, CgState<T>* st, CgState<T>* host, unsigned long long* ticket) {
  if (a.d < 1 || a.d > MAXD) throw std::runtime_error("cds: band count out of range");
  int read_bands = a.d;
  if (a.sym) { read_bands = 0; for (int b = 0; b < a.d; ++b) read_bands += a.off[b] >= 0 ? 1 : 0; }
  ObsScope obs(KID_CDS_FUSED, s, (a.d + 4.0) * (double)N * sizeof(T), (read_bands + 4.0) * (double)N * sizeof(T));   // r, p_k read; p_k+1, Ap written
  if (try_march<T, 3>(s, N, 0, N, R, a, r, Ap, p_old, p_new, nullptr, partials, nullptr, st, host, ticket)) {
    SIPX_HIP(hipGetLastError());
    return;
  }
#define SIPX_CDSF(V, D) \
  hipLaunchKernelGGL((k_cds_fused<T, V, D>), dim3(fit_grid(N / V, SIPX_CG_GRID)), dim3(BLOCK), 0, s, N, R, a, r, p_old, p_new, Ap, partials, st, host, ticket)
  constexpr int VW = sizeof(T) == 8 ? SIPX_F64_VEC : 4;
  if (N % 4 == 0) {
    switch (a.d) {
      case 1: SIPX_CDSF(VW, 1); break;
      case 3: SIPX_CDSF(VW, 3); break;
      case 5: SIPX_CDSF(VW, 5); break;
      case 7: SIPX_CDSF(VW, 7); break;
      default: SIPX_CDSF(VW, 0); break;
    }
  } else {
    SIPX_CDSF(1, 0);
  }
#undef SIPX_CDSF
  SIPX_HIP(hipGetLastError());
}

// The residual product of an x-step that follows a change of rho, with the Q update applied on the fly (z-marching matrices, one
// rank, band values generated from the descriptors): reads the four stored bands of the OLD matrix once, adds the changed sets'
// rho differences in k_q_update's order (same arithmetic: the same bits), uses the result in the product and writes it into the
// second copy of Q -- 8 N w for update + product instead of 8 + 4, and one launch less.  false: not applicable (caller: k_q_update, then the product).
template <typename T>
bool K<T>::resid_qupdate(hipStream_t s, const Grid& g, long long N, const T* R_old, T* R_new, const CdsArgs& a, const QArgs<T>& qa, const T* x,
                         const T* b, T* r, T* p, T* xold, double* partials) {
  if (!a.march || !a.sym || a.d != 7 || qa.nsets == 0) return false;
  for (int i = 0; i < qa.nsets; ++i)
    if (qa.s[i].ata) return false;                     // caller-supplied A'A bands: the separate kernel reads them
  QUpd<T> u;
  if (!make_q_plan<T>(g, a, qa, u.plan)) return false;
  for (int q = 0; q < 4; ++q) {
    u.qn[q] = R_new + (long long)a.mb[q] * N;
    u.pb[q] = -1;
    for (int b = 0; b < u.plan.nbands; ++b)
      if (u.plan.col[b] == a.mb[q]) u.pb[q] = b;
  }
  u.G = g;
  const double rw = (double)N * sizeof(T);
  ObsScope obs(KID_CDS_RESID, s, (a.d + 4.0) * rw + 3.0 * 4.0 * rw, (4.0 + 4.0) * rw + 4.0 * rw);      // SURVEY: B_resid0 + B_Q; moved: 4 bands + x, b, r, x_old + 4 bands written
  return try_march<T, 4>(s, N, 0, N, R_old, a, x, r, b, p, xold, partials, nullptr, nullptr, nullptr, nullptr, &u);
}

// ---------------------------------------------------------------------------------------------
// out[slot] = sum of the NB partials of each slot (one block per slot, fixed order)
// word (pinned, optional): the workgroup that finishes last publishes `seq` there with a system-scope release store, behind every
// workgroup's sum -- the host spins on the word instead of waiting on an event record (which costs the stream about 6 us: the
// engine records section marks on a sample of the iterations only, parsdmm_step)
__global__ __launch_bounds__(BLOCK) void k_fin_sum(const double* __restrict__ partials, double* __restrict__ out_dev,
                                                   double* __restrict__ out_host, unsigned* __restrict__ ticket,
                                                   unsigned long long* __restrict__ word, unsigned long long seq) {
  const double s = block_sum_partials(partials + (long long)blockIdx.x * NB);
  if (threadIdx.x == 0) {
    if (out_dev) out_dev[blockIdx.x] = s;
    if (out_host) out_host[blockIdx.x] = s;
    if (word) {
      __threadfence_system();
      const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t == gridDim.x - 1) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
template <typename T>
void K<T>::fin_sum(hipStream_t s, const double* partials, int nslots, double* out_dev, double* out_host, unsigned* ticket,
                   unsigned long long* word, unsigned long long seq) {
  ObsScope obs(KID_FIN_SUM, s, (double)nslots * NB * sizeof(double));
  hipLaunchKernelGGL(k_fin_sum, dim3(nslots), dim3(BLOCK), 0, s, partials, out_dev, out_host, ticket, word, seq);
  SIPX_HIP(hipGetLastError());
}

// n doubles from device memory into (pinned) host memory: what a hipMemcpyAsync would do through the copy engine, without its
// set-up latency on the stream (sharded: the all-reduced per-set sums on their way to the host's rules)
__global__ __launch_bounds__(BLOCK) void k_copy_f64(const double* __restrict__ src, double* __restrict__ dst, int n) {
  for (int i = threadIdx.x; i < n; i += BLOCK) dst[i] = src[i];
}
template <typename T>
void K<T>::copy_f64(hipStream_t s, const double* src, double* dst, int n) {
  hipLaunchKernelGGL(k_copy_f64, dim3(1), dim3(BLOCK), 0, s, src, dst, n);
  SIPX_HIP(hipGetLastError());
}

// explicit instantiation of the members defined in this file
#define SIPX_INST(T)                                                                                                  \
  template void K<T>::spmv(hipStream_t, const Grid&, long long, const T*, const CdsArgs&, const T*, T*);             \
  template void K<T>::spmv_dot(hipStream_t, long long, long long, long long, const T*, const CdsArgs&, const T*, T*, \
                               double*, const CgState<T>*);                                                           \
  template void K<T>::spmv_fused(hipStream_t, long long, const T*, const CdsArgs&, const T*, const T*, T*, T*, double*, \
                                 CgState<T>*, CgState<T>*, unsigned long long*);                                      \
  template void K<T>::resid(hipStream_t, long long, long long, long long, const T*, const CdsArgs&, const T*, const T*, T*, \
                            T*, T*, double*);                                                                         \
  template bool K<T>::resid_qupdate(hipStream_t, const Grid&, long long, const T*, T*, const CdsArgs&, const QArgs<T>&, const T*, \
                                    const T*, T*, T*, T*, double*);                                                   \
  template void K<T>::q_axpy(hipStream_t, long long, T*, const T*, T);                                               \
  template void K<T>::q_update_mk(hipStream_t, const Grid&, const CdsArgs&, const MkArgs<T>&, T*);                    \
  template void K<T>::mirror_bands(hipStream_t, long long, const CdsArgs&, T*);                                       \
  template void K<T>::sq_spmv(hipStream_t, const Grid&, const StencilQ<T>&, const T*, T*);                           \
  template void K<T>::sq_spmv_dot(hipStream_t, const Grid&, const StencilQ<T>&, const T*, T*, double*,               \
                                  const CgState<T>*);                                                                 \
  template void K<T>::sq_resid(hipStream_t, const Grid&, const StencilQ<T>&, const T*, const T*, T*, T*, T*, double*); \
  template void K<T>::q_update(hipStream_t, const Grid&, long long, long long, const CdsArgs&, const QArgs<T>&, T*);  \
  template void K<T>::gen_ata(hipStream_t, const Grid&, int, const int*, const T*, int, const long long*, T*);       \
  template void K<T>::cg_begin(hipStream_t, double*, CgState<T>*, CgState<T>*, int, T, unsigned, unsigned long long*);  \
  template void K<T>::cg_update_xr(hipStream_t, long long, const T*, T*, const T*, T*, const T*, const T*, double*, CgState<T>*, \
                                   CgState<T>*, int, unsigned long long*, long long, long long);                     \
  template void K<T>::cg_update_p(hipStream_t, long long, T*, const T*, const double*, CgState<T>*, CgState<T>*,     \
                                  unsigned long long*, long long, long long);                                        \
  template void K<T>::fin_sum(hipStream_t, const double*, int, double*, double*, unsigned*, unsigned long long*, unsigned long long); \
  template void K<T>::copy_f64(hipStream_t, const double*, double*, int);
SIPX_INST(float)
SIPX_INST(double)

}  // namespace sipx
