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
// g - stride.  A workgroup owns a tile of (V LX) x TY grid points of a plane and marches along the last dimension:
//   -x neighbour: the lane to the left (one shuffle); at a tile edge inside the grid the one point is recomputed;
//   -y neighbour: the thread one row up, through LDS (double buffered by plane parity: one barrier per plane); the row in
//                 front of the tile is recomputed by the tile's first row of threads -- for the blocks that difference along
//                 y only, 1 / TY of one block's work;
//   -z neighbour: the same thread's values of the previous plane, kept in registers; the plane in front of a chunk of planes
//                 is recomputed once per chunk.
// Recomputation is exact because the prox is element-wise once its scalars (theta, scale) are known.  Every element goes
// through the arithmetic of k_yl / k_rhs / k_adj_norm in the same order (-ffp-contract=off), so y, l and rhs are
// bit-identical to the separate kernels (tested); the float64 sums differ in their summation order only.
//
// The kernel is COMPILED PER BLOCK LAYOUT (which block is an identity, a difference along x / y / z, the distance term; where
// the sets end): a first version that took the layout at run time carried every variant of every block (24 000 instructions,
// 256 VGPRs, one wave per SIMD) and was bound by instruction issue, not by memory.  The layouts of the BASELINE configurations
// and of the common one- and two-operator lists are instantiated; any other list keeps the per-set kernels.
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "sipx_device.h"

namespace sipx {

#ifndef SIPX_MULTI_PREFETCH
#define SIPX_MULTI_PREFETCH 1
#endif
#ifndef SIPX_MULTI_VF
#define SIPX_MULTI_VF 4          // points per thread, Float32 (Float64: half as many)
#endif
#ifndef SIPX_MULTI_WAVES
#define SIPX_MULTI_WAVES 1       // waves per SIMD the register allocation aims at
#endif
constexpr int MULTI_NT = 256;

// ---- block layouts: 4 bits per block, block 0 in the lowest nibble: kind (3 bits) | last block of its set (bit 3) ----
enum { LK_I = 1, LK_X = 2, LK_Y = 3, LK_Z = 4, LK_D = 5 };      // identity / forward difference along x, y, z / distance term
constexpr unsigned long long lay_blk(int kind, bool last) { return (unsigned long long)(kind | (last ? 8 : 0)); }
constexpr unsigned long long lay_pack(unsigned long long b0, unsigned long long b1 = 0, unsigned long long b2 = 0, unsigned long long b3 = 0,
                                      unsigned long long b4 = 0, unsigned long long b5 = 0, unsigned long long b6 = 0,
                                      unsigned long long b7 = 0) {
  return b0 | (b1 << 4) | (b2 << 8) | (b3 << 12) | (b4 << 16) | (b5 << 20) | (b6 << 24) | (b7 << 28);
}
constexpr int lay_kind(unsigned long long L, int b) { return (int)((L >> (4 * b)) & 7); }
constexpr bool lay_last(unsigned long long L, int b) { return ((L >> (4 * b)) & 8) != 0; }
constexpr bool lay_first(unsigned long long L, int b) { return b == 0 || lay_last(L, b - 1); }
constexpr int lay_count(unsigned long long L) {
  int n = 0;
  while (n < MULTI_MAXB && lay_kind(L, n) != 0) ++n;
  return n;
}
constexpr int lay_count_kind(unsigned long long L, int kind) {
  int n = 0;
  for (int b = 0; b < MULTI_MAXB; ++b) n += lay_kind(L, b) == kind;
  return n;
}
constexpr int lay_index_among(unsigned long long L, int b, int kind) {      // how many blocks of `kind` precede block b
  int n = 0;
  for (int c = 0; c < b; ++c) n += lay_kind(L, c) == kind;
  return n;
}
constexpr int lay_sets(unsigned long long L) {
  int n = 0;
  for (int b = 0; b < MULTI_MAXB; ++b) n += (lay_kind(L, b) != 0 && lay_last(L, b)) ? 1 : 0;
  return n;
}

// The prox of a block in ONE branch-free form, so that the kernel carries no switch over projector kinds:
//     y = clamp( fill ? scale : soft_threshold(v, theta) * scale, lo, hi )
// where every factor a block does not use is an EXACT identity -- soft_threshold(v, 0) = v (signed zeros included), v * 1 = v,
// clamping to (-inf, +inf) -- so each kind gets the bits of prox_apply: bounds (lo, hi), l1 ball / prox_l1 (theta), l2 ball /
// annulus (scale, fill).  The distance term, (v rho + m) / (rho + 1) with its Float64 division (prox_l2s!.jl:4), is the one
// other form.  Per-element bounds and cardinality are not taken by this kernel (the engine keeps the per-set kernels then).
template <typename T>
struct UniProx {
  T lo, hi, theta, scale;
  int fill;
};
template <typename T>
__device__ __forceinline__ UniProx<T> make_uniprox(const MultiBlk<T>& B) {
  UniProx<T> u;
  u.lo = -(T)INFINITY; u.hi = (T)INFINITY; u.theta = T(0); u.scale = T(1); u.fill = 0;
  if (B.prox == PX_BOUNDS) { u.lo = B.plo; u.hi = B.phi; }
  if (B.prox == PX_L1) u.theta = B.ps->theta;
  if (B.prox == PX_PROX_L1) u.theta = T(1) / B.phi;              // prox_l1!(x, constraint.max): threshold 1/rho
  if (B.prox == PX_L2 || B.prox == PX_ANNULUS) { u.scale = B.ps->scale; u.fill = B.ps->fill; }
  return u;
}

// s = A x at VV consecutive points of a line, then the element-wise update of update_y_l.jl:64-78 (the arithmetic of k_yl).
// KIND is the block's compile-time kind; `valid`: the forward-difference row exists.
template <typename T, int VV, int KIND>
__device__ __forceinline__ void multi_block_update(const MultiBlk<T>& B, const UniProx<T>& pc, const Vec<T, VV>& xc, const Vec<T, VV>& xn,
                                                   const bool (&valid)[VV], const Vec<T, VV>& yv, const Vec<T, VV>& lv, const Vec<T, VV>& mv,
                                                   Vec<T, VV>& yn, Vec<T, VV>& ln, T (&rp)[VV], T (&sv)[VV]) {
  const bool relax = !(B.gamma == T(1));
  const T gam = B.gamma, omg = T(1) - B.gamma, nih = -B.ih;
#pragma unroll
  for (int k = 0; k < VV; ++k) {
    T s;
    if constexpr (KIND == LK_I || KIND == LK_D) {
      s = xc.v[k];
    } else {
      const T d = nih * xc.v[k] + B.ih * xn.v[k];              // fwd_dir: the two products of a CSC row in column order
      s = valid[k] ? d : T(0);
    }
    const T yo = yv.v[k], lo = lv.v[k];
    const T xh = relax ? (gam * s + omg * yo) : s;              // update_y_l.jl:72
    const T v = xh - lo * B.rho1;                               // :67 / :74
    T y1;
    if constexpr (KIND == LK_D) {
      y1 = (T)((double)(v * B.rho + mv.v[k]) / ((double)B.rho + 1.0));     // prox_l2s!.jl:4
    } else {
      const T t = soft_thr(v, pc.theta);                        // project_l1_Duchi!.jl:49, prox_l1!.jl:9
      const T q = pc.fill ? pc.scale : t * pc.scale;            // project_l2!.jl / project_annulus!.jl:9-17
      const T c = q < pc.hi ? q : pc.hi;                        // max(LB, min(x, UB))  project_bounds!.jl:9
      y1 = pc.lo > c ? pc.lo : c;
      if constexpr (KIND != LK_I) { if (!valid[k]) y1 = T(0); }
    }
    rp[k] = y1 - s;                                             // :69 / :76
    ln.v[k] = relax ? (lo + B.rho * (y1 - xh)) : (lo + B.rho * rp[k]);   // :70 / :77
    yn.v[k] = y1;
    sv[k] = s;
  }
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (the block index selects code, not just data)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// FULL: the variant for iterations with Barzilai-Borwein sums (F_BB: the four snapshot arrays are read, l_hat_0 and s_0
// rewritten, six more sums per set -- adapt_rho_gamma.jl:41-53, PARSDMM.jl:192-206), the first iteration (F_FIRST: snapshots
// written, PARSDMM.jl:164-180) and / or feasibility estimates of the element-wise sets (F_FEAS, update_y_l.jl:90-99).  Its
// eight extra sums per set live in per-thread LDS slots (as registers they would halve the occupancy of the plain variant).
template <typename T, int V, unsigned long long L, bool FULL, bool X0>
__global__ __launch_bounds__(MULTI_NT, SIPX_MULTI_WAVES) void k_yl_multi(Grid G, MultiArgs<T> a, int lgLX, int tiles_x, int tiles_y, int zchunk,
                                                       long long items, long long jlo, long long jhi, long long jsum, int xsplit) {
  constexpr int NBLK = lay_count(L), NY = lay_count_kind(L, LK_Y), NZ = lay_count_kind(L, LK_Z);
  constexpr int NSETS = lay_sets(L), NI = lay_count_kind(L, LK_I);
  constexpr int NACC = FULL ? 6 * NSETS : 1;                    // the six BB sums of every set
  // (summed over the four lanes of a quad first -- two DPP hops -- so that a slot per quad suffices: 15 KB instead of 60 KB,
  // which decides whether two workgroups fit a compute unit)
  __shared__ double lacc[NACC][MULTI_NT / 4];
  __shared__ T ybuf[2][NY > 0 ? NY : 1][2][V][MULTI_NT];       // [plane parity][y block][w | dy][element][thread]
  const bool f_first = FULL && (a.flags & F_FIRST) != 0, f_bb = FULL && (a.flags & F_BB) != 0 && !f_first;
  const bool f_feas = FULL && (a.flags & F_FEAS) != 0;
  if constexpr (FULL) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) lacc[q][threadIdx.x >> 2] = 0.0;   // a quad's slot is touched by its lane 0 only: no barrier needed
  }
  const int tid = threadIdx.x, LX = 1 << lgLX, tx = tid & (LX - 1), ty = tid >> lgLX, TY = MULTI_NT >> lgLX;
  const long long n1 = G.n[0], n2 = G.n[1], n3 = G.n[2], st1 = G.st[1], st2 = G.st[2];
  const bool fuse_rhs = a.rhs != nullptr;

  UniProx<T> pc[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) pc[b] = make_uniprox<T>(a.b[b]);
  double acc_rp[NBLK], acc_du[NBLK];      // r_pri sums per block (folded per set at the end); r_dual at a set's last block
#pragma unroll
  for (int b = 0; b < NBLK; ++b) acc_rp[b] = acc_du[b] = 0;
  double acc_obj = 0, acc_evo = 0, acc_xx = 0;
  double acc_fe[NI > 0 ? NI : 1], acc_ss[NI > 0 ? NI : 1];      // feasibility estimate of the element-wise identity sets (FULL)
#pragma unroll
  for (int q = 0; q < (NI > 0 ? NI : 1); ++q) acc_fe[q] = acc_ss[q] = 0;

  const long long tiles = (long long)tiles_x * tiles_y;
  for (long long item = blockIdx.x; item < items; item += gridDim.x) {
    const long long zc = item / tiles, tile = item - zc * tiles;
    const int tile_y = (int)(tile / tiles_x), tile_x = (int)(tile - (long long)tile_y * tiles_x);
    // xsplit (2-D grids, marched along their second dimension): the TY "rows" of the tile are TY segments of ONE grid line
    const long long i0 = xsplit ? (((long long)tile_x * TY + ty) * LX + tx) * V : ((long long)tile_x * LX + tx) * V;
    const long long j = xsplit ? jlo + tile_y : jlo + (long long)tile_y * TY + ty;
    const bool active = i0 < n1 && j < jhi;
    const long long k0 = a.zlo + zc * zchunk, k1 = (k0 + zchunk < a.zhi) ? k0 + zchunk : a.zhi;
    // Addresses = uniform base of the plane (scalar registers) + the thread's 32-bit offset inside the plane: one vector
    // register serves every array (a 64-bit vector address per array cost 40 registers and as many 64-bit adds per plane)
    const unsigned go = active ? (unsigned)(i0 + st1 * j) : 0u;       // inactive threads shadow point 0 of the plane (nothing stored)
    bool vx[V], mxm[V];                                        // forward-difference row exists / its left neighbour row exists (x)
#pragma unroll
    for (int k = 0; k < V; ++k) { vx[k] = (i0 + k) < n1 - 1; mxm[k] = (i0 + k) > 0; }
    const bool vy = j < n2 - 1, mym = j > 0;
    if constexpr (NY > 0) __syncthreads();                     // the previous item's LDS traffic is over

    // the previous plane of the blocks that difference along the march dimension: registers, recomputed at a chunk's start
    T zw[NZ > 0 ? NZ : 1][V], zd[NZ > 0 ? NZ : 1][V];
    static_for<0, NBLK>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      if constexpr (lay_kind(L, b) == LK_Z) {
        constexpr int zi = lay_index_among(L, b, LK_Z);
        const MultiBlk<T>& B = a.b[b];
#pragma unroll
        for (int k = 0; k < V; ++k) zw[zi][k] = zd[zi][k] = T(0);
        if (active && k0 > 0) {
          const long long pz = st2 * (k0 - 1);
          const Vec<T, V> xc = ldv<T, V>(a.x + pz + go), xn = ldv_u<T, V>(a.x + pz + st2 + go);
          const Vec<T, V> yv = ldv<T, V>(B.y + pz + go), lv = ldv<T, V>(B.l + pz + go);
          bool valid[V];
#pragma unroll
          for (int k = 0; k < V; ++k) valid[k] = true;        // plane k0 - 1 <= n3 - 2
          Vec<T, V> yn, ln;
          T rp[V], sv[V];
          multi_block_update<T, V, LK_Z>(B, pc[b], xc, xn, valid, yv, lv, zerov<T, V>(), yn, ln, rp, sv);
#pragma unroll
          for (int k = 0; k < V; ++k) { zw[zi][k] = B.rho * yn.v[k] + ln.v[k]; zd[zi][k] = yn.v[k] - yv.v[k]; }
        }
      }
    });

    Vec<T, V> xnext = active ? ldv<T, V>(a.x + st2 * k0 + go) : zerov<T, V>();
    // x0 mode (a.x0 != nullptr): s_0 = A x_0 is RECOMPUTED from a snapshot of x (one array for all sets) instead of being kept
    // per set -- the same arithmetic on the same x, so the same bits -- which saves reading and rewriting s_0 of every block
    constexpr bool x0mode = FULL && X0;
    Vec<T, V> x0next = (x0mode && f_bb && active) ? ldv<T, V>(a.x0 + st2 * k0 + go) : zerov<T, V>();
    for (long long kz = k0; kz < k1; ++kz) {
      const int par = (int)(kz & 1);
      const long long pz = st2 * kz;                           // uniform: the plane's first point
      const Vec<T, V> xc = xnext;
      const bool vz = kz < n3 - 1, mzm = kz > 0;
      // planes (2-D: rows) in front of zsum (jsum) are the last ones of the rank below, recomputed: stored, not summed
      const bool own = kz >= a.zsum && j >= jsum;
      Vec<T, V> xpx = zerov<T, V>(), xpy = zerov<T, V>();
      // ---- all loads of the plane first: the stores of one block must not hold back the loads of the next --------------
      Vec<T, V> yv[NBLK], lv[NBLK], mv = zerov<T, V>(), xo = zerov<T, V>();
#pragma unroll
      for (int b = 0; b < NBLK; ++b) { yv[b] = zerov<T, V>(); lv[b] = zerov<T, V>(); }
      if (active) {
        xnext = ldv_u<T, V>(a.x + pz + st2 + go);              // x carries an end halo of a plane: unconditional
        if constexpr (lay_count_kind(L, LK_X) > 0) xpx = ldv_u<T, V>(a.x + pz + 1 + go);
        if constexpr (NY > 0) xpy = ldv_u<T, V>(a.x + pz + st1 + go);
        if constexpr (SIPX_MULTI_PREFETCH && !FULL) {
#pragma unroll
          for (int b = 0; b < NBLK; ++b) {
            yv[b] = ldv_nt<T, V>(a.b[b].y + pz + go);         // last use of the old iterate: streaming loads
            lv[b] = ldv_nt<T, V>(a.b[b].l + pz + go);
          }
        }
        if constexpr (lay_count_kind(L, LK_D) > 0) { mv = ldv<T, V>(a.m + pz + go); xo = ldv<T, V>(a.xold + pz + go); }
      }
      Vec<T, V> x0c = x0next, x0px = zerov<T, V>(), x0py = zerov<T, V>();
      if constexpr (FULL) {
        if (x0mode && f_bb && active) {
          x0next = ldv_u<T, V>(a.x0 + pz + st2 + go);
          if constexpr (lay_count_kind(L, LK_X) > 0) x0px = ldv_u<T, V>(a.x0 + pz + 1 + go);
          if constexpr (NY > 0) x0py = ldv_u<T, V>(a.x0 + pz + st1 + go);
        }
        // the new snapshot goes into the OTHER snapshot array: neighbouring threads, tiles and chunks still read the old one
        if (x0mode && a.x0w && (f_bb || f_first) && active) stv_nt<T, V>(a.x0w + pz + go, xc);     // (x0w == nullptr: the caller keeps x itself as the snapshot)
      }
      // ---- phase A: the update of every block; w = rho y + l and y - y_old stay in registers ---------------------------
      T wv[NBLK][V], dv[NBLK][V];
      static_for<0, NBLK>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        constexpr int KIND = lay_kind(L, b);
        const MultiBlk<T>& B = a.b[b];
#pragma unroll
        for (int k = 0; k < V; ++k) wv[b][k] = dv[b][k] = T(0);
        if (active) {
          if constexpr (!SIPX_MULTI_PREFETCH || FULL) {       // (the variant with the snapshot arrays loads block by block: registers)
            yv[b] = ldv_nt<T, V>(B.y + pz + go);
            lv[b] = ldv_nt<T, V>(B.l + pz + go);
          }
          bool valid[V];
#pragma unroll
          for (int k = 0; k < V; ++k) valid[k] = KIND == LK_X ? vx[k] : (KIND == LK_Y ? vy : (KIND == LK_Z ? vz : true));
          const Vec<T, V>& xn = KIND == LK_X ? xpx : (KIND == LK_Y ? xpy : xnext);
          Vec<T, V> yn, ln;
          T rp[V], sv[V];
          multi_block_update<T, V, KIND>(B, pc[b], xc, xn, valid, yv[b], lv[b], mv, yn, ln, rp, sv);
          if constexpr (FULL) {
            constexpr int si = lay_sets(L & ((1ull << (4 * b)) - 1ull));      // index of the block's set
            Vec<T, V> lh, svv;
#pragma unroll
            for (int k = 0; k < V; ++k) {
              lh.v[k] = lv[b].v[k] + B.rho * (yv[b].v[k] - sv[k]);           // l_hat = l_old + rho(-s + y_old)  PARSDMM.jl:173
              svv.v[k] = sv[k];
            }
            if (f_bb) {                                                        // adapt_rho_gamma.jl:41-53
              // (the snapshot arrays are touched once every rho_update_frequency iterations: streaming loads and stores)
              const Vec<T, V> a0 = ldv_nt<T, V>(B.lh0 + pz + go), b0 = ldv_nt<T, V>(B.y0 + pz + go), d0 = ldv_nt<T, V>(B.l0 + pz + go);
              Vec<T, V> c0;
              if (x0mode) {                                  // s_0 = A x_0 at this point: fwd_dir on the snapshot
                const Vec<T, V>& x0n = KIND == LK_X ? x0px : (KIND == LK_Y ? x0py : x0next);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                  if constexpr (KIND == LK_I || KIND == LK_D) {
                    c0.v[k] = x0c.v[k];
                  } else {
                    const T d = (-B.ih) * x0c.v[k] + B.ih * x0n.v[k];
                    c0.v[k] = valid[k] ? d : T(0);
                  }
                }
              } else {
                c0 = ldv_nt<T, V>(B.s0 + pz + go);
              }
              double q6[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
              for (int k = 0; k < V; ++k) {
                const T dlh = lh.v[k] - a0.v[k], dH = sv[k] - c0.v[k], dl = ln.v[k] - d0.v[k], dG = -(yn.v[k] - b0.v[k]);
                q6[0] += (double)dH * (double)dlh;
                q6[1] += (double)dH * (double)dH;
                q6[2] += (double)dlh * (double)dlh;
                q6[3] += (double)dl * (double)dl;
                q6[4] += (double)dG * (double)dG;
                q6[5] += (double)dG * (double)dl;
              }
#pragma unroll
              for (int q = 0; q < 6; ++q) {
                double v = own ? q6[q] : 0.0;
                v += dpp_hop<0xb1, 0xf>(v);        // quad_perm:[1,0,3,2]
                v += dpp_hop<0x4e, 0xf>(v);        // quad_perm:[2,3,0,1]: every lane of the quad holds the quad's sum
                if ((tid & 3) == 0) lacc[6 * si + q][tid >> 2] += v;
              }
            }
            if (f_bb || f_first) {                                             // PARSDMM.jl:174-177, 200-203
              stv_nt<T, V>(B.lh0 + pz + go, lh);     // y_0 <- y and l_0 <- l need no copy: the update below is written into
              if (!x0mode) stv_nt<T, V>(B.s0 + pz + go, svv);     // the snapshot pair itself (engine)
            }
            if constexpr (KIND == LK_I) {
              if (f_feas && B.feas_el && own) {                                // update_y_l.jl:90-99 (element-wise sets)
                constexpr int ii = lay_index_among(L, b, LK_I);
                double fe = 0, ss = 0;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                  const T t = soft_thr(sv[k], pc[b].theta);
                  const T q = pc[b].fill ? pc[b].scale : t * pc[b].scale;
                  const T c = q < pc[b].hi ? q : pc[b].hi;
                  const T psv = pc[b].lo > c ? pc[b].lo : c;
                  const T d = psv - sv[k];
                  fe += (double)d * (double)d;
                  ss += (double)sv[k] * (double)sv[k];
                }
                acc_fe[ii] += fe;
                acc_ss[ii] += ss;
              }
            }
          }
          stv_nt<T, V>(B.yo + pz + go, yn);
          stv_nt<T, V>(B.lo + pz + go, ln);
#pragma unroll
          for (int k = 0; k < V; ++k) {
            wv[b][k] = B.rho * yn.v[k] + ln.v[k];              // rhs_compose.jl:28-30 on the new iterate
            dv[b][k] = yn.v[k] - yv[b].v[k];                   // x_hat = y - y_old  update_y_l.jl:82
            if (own) acc_rp[b] += (double)rp[k] * (double)rp[k];
            if constexpr (KIND == LK_I || KIND == LK_D) { if (own) acc_du[b] += (double)dv[b][k] * (double)dv[b][k]; }
          }
          if constexpr (KIND == LK_D) {                         // PARSDMM.jl:140,145
            if (own) {
#pragma unroll
              for (int k = 0; k < V; ++k) {
                const T dd = xc.v[k] - mv.v[k], ev = xo.v[k] - xc.v[k];
                acc_obj += (double)dd * (double)dd;
                acc_evo += (double)ev * (double)ev;
                acc_xx += (double)xc.v[k] * (double)xc.v[k];
              }
            }
          }
        }
        if constexpr (KIND == LK_Y) {
          constexpr int yi = lay_index_among(L, b, LK_Y);
#pragma unroll
          for (int k = 0; k < V; ++k) { ybuf[par][yi][0][k][tid] = wv[b][k]; ybuf[par][yi][1][k][tid] = dv[b][k]; }
        }
      });
      if constexpr (NY > 0) __syncthreads();
      // ---- phase B: adjoint stencils on the new values -> r_dual sums and the right-hand side ---------------------------
      T out[V], tr[V], td[V];
#pragma unroll
      for (int k = 0; k < V; ++k) out[k] = tr[k] = td[k] = T(0);
      static_for<0, NBLK>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        constexpr int KIND = lay_kind(L, b);
        const MultiBlk<T>& B = a.b[b];
        if constexpr (lay_first(L, b)) {
#pragma unroll
          for (int k = 0; k < V; ++k) tr[k] = td[k] = T(0);
        }
        if constexpr (KIND == LK_I || KIND == LK_D) {          // identity: t = rho y + l (k_rhs), no neighbour
#pragma unroll
          for (int k = 0; k < V; ++k) tr[k] = wv[b][k];
        } else {
          T pw[V], pd[V];                                       // the values at g - stride
          bool mm[V], mc[V];                                    // that row exists / the row at g exists
          if constexpr (KIND == LK_X) {
            // the lane to the left holds the point in front of this vector; at the left edge of a tile inside the grid the
            // one point is recomputed
            T lw = __shfl_up(wv[b][V - 1], 1, 64), ld = __shfl_up(dv[b][V - 1], 1, 64);
            if (tx == 0 && active && i0 > 0) {       // (xsplit: also the first lane of every segment inside the tile)
              Vec<T, 1> x1, x2, y1v, l1v, yn1, ln1;
              x1.v[0] = (a.x + pz - 1)[go]; x2.v[0] = xc.v[0];
              y1v.v[0] = (B.y + pz - 1)[go]; l1v.v[0] = (B.l + pz - 1)[go];
              const bool v1[1] = {true};
              T rp1[1], sv1[1];
              multi_block_update<T, 1, LK_X>(B, pc[b], x1, x2, v1, y1v, l1v, zerov<T, 1>(), yn1, ln1, rp1, sv1);
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
          } else if constexpr (KIND == LK_Y) {
            constexpr int yi = lay_index_among(L, b, LK_Y);
            if (ty > 0) {
#pragma unroll
              for (int k = 0; k < V; ++k) { pw[k] = ybuf[par][yi][0][k][tid - LX]; pd[k] = ybuf[par][yi][1][k][tid - LX]; }
            } else {
#pragma unroll
              for (int k = 0; k < V; ++k) pw[k] = pd[k] = T(0);
              if (active && mym) {                              // the row in front of the tile: recomputed
                const Vec<T, V> x1 = ldv<T, V>(a.x + pz - st1 + go), yh = ldv<T, V>(B.y + pz - st1 + go), lh = ldv<T, V>(B.l + pz - st1 + go);
                Vec<T, V> yn, ln;
                bool valid[V];
#pragma unroll
                for (int k = 0; k < V; ++k) valid[k] = true;
                T rp1[V], sv1[V];
                multi_block_update<T, V, LK_Y>(B, pc[b], x1, xc, valid, yh, lh, zerov<T, V>(), yn, ln, rp1, sv1);
#pragma unroll
                for (int k = 0; k < V; ++k) { pw[k] = B.rho * yn.v[k] + ln.v[k]; pd[k] = yn.v[k] - yh.v[k]; }
              }
            }
#pragma unroll
            for (int k = 0; k < V; ++k) { mm[k] = mym; mc[k] = vy; }
          } else {
            constexpr int zi = lay_index_among(L, b, LK_Z);
#pragma unroll
            for (int k = 0; k < V; ++k) {
              pw[k] = zw[zi][k];
              pd[k] = zd[zi][k];
              mm[k] = mzm;
              mc[k] = vz;
              zw[zi][k] = wv[b][k];
              zd[zi][k] = dv[b][k];
            }
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
        if constexpr (lay_last(L, b)) {
#pragma unroll
          for (int k = 0; k < V; ++k) {
            if (B.in_rhs) out[k] = out[k] + tr[k];              // sets added in order (rhs_compose.jl:24-31)
            if constexpr (KIND != LK_I && KIND != LK_D) { if (active && own) acc_du[b] += (double)td[k] * (double)td[k]; }
          }
        }
      });
      if (fuse_rhs && active && own) {
        Vec<T, V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] = out[k];
        stv<T, V>(a.rhs + pz + go, o);
      }
    }
  }
  // ---- sums: the blocks of a set fold into one r_pri sum; r_dual sits at the set's last block; one block-wide reduction
  // per slot.  Flat slot index into the engine's partial array: set * SET_SLOTS + {SL_RPRI, SL_DY | SL_ADJ, SL_OBJ ...}
  constexpr bool HAS_D = lay_count_kind(L, LK_D) > 0;
  constexpr int K0 = 2 * NSETS + (HAS_D ? 3 : 0);
  constexpr int K = K0 + (FULL ? NACC + 2 * NI : 0);
  double acc[K];
  int slots[K];
  {
    double rp_set = 0;
    static_for<0, NBLK>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      rp_set = lay_first(L, b) ? acc_rp[b] : rp_set + acc_rp[b];
      if constexpr (lay_last(L, b)) {
        constexpr int KIND = lay_kind(L, b);
        constexpr int si = lay_sets(L & ((1ull << (4 * b)) - 1ull));         // sets that end before block b
        acc[2 * si] = rp_set;
        slots[2 * si] = a.b[b].set * SET_SLOTS + SL_RPRI;
        acc[2 * si + 1] = acc_du[b];
        slots[2 * si + 1] = a.b[b].set * SET_SLOTS + ((KIND == LK_I || KIND == LK_D) ? SL_DY : SL_ADJ);
        if constexpr (KIND == LK_D) {
          acc[2 * NSETS] = acc_obj; acc[2 * NSETS + 1] = acc_evo; acc[2 * NSETS + 2] = acc_xx;
          slots[2 * NSETS] = a.b[b].set * SET_SLOTS + SL_OBJ;
          slots[2 * NSETS + 1] = a.b[b].set * SET_SLOTS + SL_EVO;
          slots[2 * NSETS + 2] = a.b[b].set * SET_SLOTS + SL_XX;
        }
        if constexpr (FULL) {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            acc[K0 + 6 * si + q] = (tid & 3) == 0 ? lacc[6 * si + q][tid >> 2] : 0.0;
            slots[K0 + 6 * si + q] = a.b[b].set * SET_SLOTS + SL_HL + q;       // SL_HL, HH, LH, DL, GG, GL are consecutive
          }
        }
      }
      if constexpr (FULL && lay_kind(L, b) == LK_I) {
        constexpr int ii = lay_index_among(L, b, LK_I);
        acc[K0 + 6 * NSETS + 2 * ii] = acc_fe[ii];
        acc[K0 + 6 * NSETS + 2 * ii + 1] = acc_ss[ii];
        slots[K0 + 6 * NSETS + 2 * ii] = a.b[b].set * SET_SLOTS + SL_FE;
        slots[K0 + 6 * NSETS + 2 * ii + 1] = a.b[b].set * SET_SLOTS + SL_SS;
      }
    });
  }
  __shared__ double sm[K][MULTI_NT / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) sm[k][w] = v;
  }
  __syncthreads();
  static_assert(K <= MULTI_NT, "one thread per slot in the epilogue");
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

template <typename T, int V, unsigned long long L>
static void launch_multi(hipStream_t s, const Grid& g, const MultiArgs<T>& a, double bytes) {
  static_assert(SL_HH == SL_HL + 1 && SL_LH == SL_HL + 2 && SL_DL == SL_HL + 3 && SL_GG == SL_HL + 4 && SL_GL == SL_HL + 5, "BB slots");
  // tile geometry: LX lanes of V points along x (a power of two, at most a wave), TY = 256 / LX rows
  const long long nvx = g.n[0] / V;
  int lg = 0;
  while ((1 << lg) < nvx && lg < 6) ++lg;
  const int LX = 1 << lg, TY = MULTI_NT / LX;
  const bool three = g.n[1] > 1;       // (a 2-D grid arrives as (n1, 1, n2), see yl_multi)
  // A 2-D grid (n1, n2) is marched along its SECOND dimension (the engine presents it as (n1, 1, n2): a difference along
  // dimension 1 is then a Z block, no block needs the LDS row exchange), and the TY thread rows of a tile are TY segments of
  // one grid line (xsplit).  3-D: rows [jlo, jhi) of every plane, planes [zlo, zhi) marched.
  const int xsplit = three ? 0 : 1;
  long long jlo = 0, jhi = g.n[1], jsum = 0, zlo = a.zlo, zhi = a.zhi;
  const bool empty = jhi <= jlo || zhi <= zlo;        // a rank without planes still clears its partial slots (one idle workgroup)
  if (empty) { jhi = jlo + 1; zhi = zlo + 1; }
  const int tiles_x = xsplit ? (int)((nvx + (long long)LX * TY - 1) / ((long long)LX * TY)) : (int)((nvx + LX - 1) / LX);
  const int tiles_y = xsplit ? 1 : (int)((jhi - jlo + TY - 1) / TY);
  const long long tiles = (long long)tiles_x * tiles_y;
  // chunks of planes: enough work items to fill the chip several times over, chunks long enough that the plane recomputed in
  // front of each stays a small share (<= 1 / 8 of one block's work)
  const long long planes = zhi - zlo;
  const long long zc_env = env_knobs().multi_zchunk;          // SIPX_MULTI_ZCHUNK, read when the context was finalised
  long long want = (4ll * NB_7 + tiles - 1) / tiles;          // chunks per tile column for ~4 items per workgroup slot
  if (want < 1) want = 1;
  long long zchunk = (planes + want - 1) / want;
  if (zchunk < 8) zchunk = planes < 8 ? planes : 8;
  if (zc_env > 0) zchunk = zc_env < planes ? zc_env : planes;
  const long long nchunks = (planes + zchunk - 1) / zchunk;
  const long long items = empty ? 0 : tiles * nchunks;
  const int grid = (int)(items < 1 ? 1 : (items < NB_7 ? items : NB_7));
  MultiArgs<T> b = a;
  b.zlo = zlo; b.zhi = zhi;
  if (empty) { b.zlo = b.zhi = 0; }
  ObsScope obs(KID_YL_MULTI, s, bytes);
  if (a.flags && a.x0)
    hipLaunchKernelGGL((k_yl_multi<T, V, L, true, true>), dim3(grid), dim3(MULTI_NT), 0, s, g, b, lg, tiles_x, tiles_y, (int)zchunk, items, jlo, jhi, jsum, xsplit);
  else if (a.flags)
    hipLaunchKernelGGL((k_yl_multi<T, V, L, true, false>), dim3(grid), dim3(MULTI_NT), 0, s, g, b, lg, tiles_x, tiles_y, (int)zchunk, items, jlo, jhi, jsum, xsplit);
  else
    hipLaunchKernelGGL((k_yl_multi<T, V, L, false, false>), dim3(grid), dim3(MULTI_NT), 0, s, g, b, lg, tiles_x, tiles_y, (int)zchunk, items, jlo, jhi, jsum, xsplit);
  SIPX_HIP(hipGetLastError());
}

// the instantiated layouts (I = identity set, X / Y / Z = a set of one difference block, [..] = blocks of one set, D = distance)
#define SIPX_LAYOUTS(F)                                                                                                          \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_X, 1), lay_blk(LK_Y, 1), lay_blk(LK_Z, 1), lay_blk(LK_D, 1)))   /* C3: I X Y Z D */        \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_Z, 0), lay_blk(LK_Y, 0), lay_blk(LK_X, 1), lay_blk(LK_D, 1)))   /* C5: I [Z Y X] D */      \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_Z, 0), lay_blk(LK_X, 1), lay_blk(LK_D, 1)))                     /* C2: I [Z X] D (2-D TV, marched along dim 2) */ \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_X, 1), lay_blk(LK_Z, 1), lay_blk(LK_D, 1)))                     /* I X Z D */             \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_Z, 1), lay_blk(LK_X, 1), lay_blk(LK_D, 1)))                     /* I Z X D (2-D D_z, D_x) */ \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_Z, 1), lay_blk(LK_D, 1)))                                       /* I Z D */               \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_Y, 1), lay_blk(LK_D, 1)))                                       /* I Y D */               \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_X, 1), lay_blk(LK_D, 1)))                                       /* I X D */               \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_Z, 0), lay_blk(LK_Y, 0), lay_blk(LK_X, 1), lay_blk(LK_I, 1), lay_blk(LK_D, 1)))   /* I [Z Y X] I D */ \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_D, 1)))                                                         /* I D */ \
  F(lay_pack(lay_blk(LK_I, 1), lay_blk(LK_X, 1), lay_blk(LK_Y, 1), lay_blk(LK_Z, 1), lay_blk(LK_I, 1), lay_blk(LK_D, 1)))   /* C4's element-wise and l1 terms: I X Y Z I(annulus) D */

template <typename T>
bool K<T>::yl_multi(hipStream_t s, const Grid& g, const MultiArgs<T>& a, bool probe_only) {
  constexpr int V = sizeof(T) == 8 ? SIPX_MULTI_VF / 2 : SIPX_MULTI_VF;
  if (a.nblk < 1 || a.nblk > MULTI_MAXB || g.n[0] % V != 0) return false;
  unsigned long long code = 0;
  for (int b = 0; b < a.nblk; ++b) {
    const int px = a.b[b].prox;
    if (!(px == PX_BOUNDS || px == PX_L1 || px == PX_PROX_L1 || px == PX_L2 || px == PX_ANNULUS || px == PX_DIST)) return false;
    if (a.b[b].dir == 2 && g.n[2] <= 1) return false;
    const bool two_d = g.n[2] <= 1;
    const int kind = a.b[b].dist ? LK_D : (a.b[b].dir < 0 ? LK_I : (a.b[b].dir == 0 ? LK_X : (a.b[b].dir == 1 ? (two_d ? LK_Z : LK_Y) : LK_Z)));
    code |= lay_blk(kind, a.b[b].last != 0) << (4 * b);
  }
  // algorithmic bytes: x, (m, x_old for the distance term) read; y, l of every block read and written; rhs written
  const bool three = g.n[2] > 1;
  const double pts = (double)(a.zhi - a.zlo) * (three ? (double)g.st[2] : (double)g.st[1]);
  Grid gk = g;                      // the grid as the kernel walks it
  if (!three) {
    gk.n[1] = 1; gk.n[2] = g.n[1];
    gk.st[1] = g.n[0]; gk.st[2] = g.n[0];
  }
  double vecs = 1.0 + (a.rhs ? 1.0 : 0.0);
  const bool first = (a.flags & F_FIRST) != 0, bb = (a.flags & F_BB) != 0 && !first;
  const bool x0m = a.x0 != nullptr;       // s_0 recomputed from the snapshot of x: per block 4 (2) instead of 6 (3), + the snapshot itself
  for (int b = 0; b < a.nblk; ++b) vecs += 4.0 + (a.b[b].dist ? 2.0 : 0.0) + (bb ? (x0m ? 4.0 : 6.0) : (first ? (x0m ? 1.0 : 2.0) : 0.0));
  if (x0m) vecs += (bb ? 1.0 : 0.0) + ((a.x0w && (bb || first)) ? 1.0 : 0.0);
  const double bytes = vecs * pts * sizeof(T);
#define SIPX_TRY_LAYOUT(LL)                                    \
  if (code == (LL)) {                                          \
    if (!probe_only) launch_multi<T, V, (LL)>(s, gk, a, bytes); \
    return true;                                               \
  }
  SIPX_LAYOUTS(SIPX_TRY_LAYOUT)
#undef SIPX_TRY_LAYOUT
  return false;
}

template bool K<float>::yl_multi(hipStream_t, const Grid&, const MultiArgs<float>&, bool);
template bool K<double>::yl_multi(hipStream_t, const Grid&, const MultiArgs<double>&, bool);

}  // namespace sipx
