/* TEST INFRASTRUCTURE ONLY -- C/OpenMP port of the reference's CPU PARSDMM path.
 *
 * Purpose: (1) the CPU baseline timed beside the GPU engine (bench.py "cpu_baseline", kind
 * "port"); (2) a second, independent restatement cross-checked against oracle/parsdmm_oracle.py
 * in tests/test_port.py.  It is NOT part of the product and is never linked into libsipx.so.
 *
 * It keeps the ALGORITHMIC STRUCTURE of slimgroup/SetIntersectionProjection.jl (so the timing is
 * a fair stand-in for the reference's multithreaded CPU path, which cannot be run here: no Julia):
 *   - TD_OP as explicit sparse matrices with Int64 indices: A*x and A'*v products
 *     (src/update_y_l.jl:43,84, src/rhs_compose.jl:28)
 *   - CDS SpMV one threaded sweep PER DIAGONAL after a fill!  (src/CDS_MVp_MT.jl:17-23,
 *     src/CDS_MVp_MT_subfunc.jl:15-18, src/argmin_x.jl:72-78)
 *   - un-fused CG: separate dot / axpy / norm passes, 2 + cg_it SpMVs per iteration
 *     (src/cg.jl:82-115, src/argmin_x.jl:34,39)
 *   - sort-based l1-ball projection: radix sort, pairwise cumsum, serial scan
 *     (src/projectors/project_l1_Duchi!.jl:33-46)
 *   - the same y/l update, Barzilai-Borwein adaptation, stopping rules and incremental Q update.
 * Where Julia runs a loop serially (broadcasts, sparse products) this port is MORE generous and
 * threads it with OpenMP -- the baseline errs on the fast side.
 * Reductions follow the oracle's convention: float64 accumulation, one rounding to T.
 *
 * Build: see oracle/Makefile (REAL = float | double).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL T;
typedef int64_t I;

enum { OP_ID = 0, OP_DX = 1, OP_DY = 2, OP_DZ = 3, OP_TV = 4 };
enum { PJ_BOUNDS = 0, PJ_L1 = 2, PJ_DIST = 100 };

static double now(void) { return omp_get_wtime(); }
static T eps_T(void) { return sizeof(T) == 4 ? (T)1.1920928955078125e-07 : (T)2.220446049250313e-16; }

/* ---- sparse matrix in both orientations (CSR for A*x, CSC for A'*v), Int64 indices ---- */
typedef struct {
  I rows, cols, nnz;
  I *rowptr, *colind; T* rval;   /* CSR, columns ascending inside a row */
  I *colptr, *rowind; T* cval;   /* CSC, rows ascending inside a column */
} Sp;

static void sp_free(Sp* A) { free(A->rowptr); free(A->colind); free(A->rval); free(A->colptr); free(A->rowind); free(A->cval); }

static void sp_finish_csc(Sp* A) {
  A->colptr = calloc(A->cols + 1, sizeof(I));
  A->rowind = malloc(A->nnz * sizeof(I));
  A->cval = malloc(A->nnz * sizeof(T));
  for (I k = 0; k < A->nnz; ++k) A->colptr[A->colind[k] + 1]++;
  for (I j = 0; j < A->cols; ++j) A->colptr[j + 1] += A->colptr[j];
  I* pos = malloc(A->cols * sizeof(I));
  memcpy(pos, A->colptr, A->cols * sizeof(I));
  for (I r = 0; r < A->rows; ++r)
    for (I k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) {
      I p = pos[A->colind[k]]++;
      A->rowind[p] = r;
      A->cval[p] = A->rval[k];
    }
  free(pos);
}

/* get_discrete_Grad (src/get_discrete_Grad.jl:16-76): forward differences with entries -1/h, +1/h */
static void build_op(Sp* A, int op, int ndim, const I* n, const T* ih) {
  I N = n[0] * n[1] * n[2];
  I st[3] = {1, n[0], n[0] * n[1]};
  int dirs[3], nblk = 0;
  int zdir = ndim == 2 ? 1 : 2;
  if (op == OP_DX) { dirs[nblk++] = 0; }
  else if (op == OP_DY) { dirs[nblk++] = 1; }
  else if (op == OP_DZ) { dirs[nblk++] = zdir; }
  else if (op == OP_TV) { if (ndim == 2) { dirs[0] = 1; dirs[1] = 0; nblk = 2; } else { dirs[0] = 2; dirs[1] = 1; dirs[2] = 0; nblk = 3; } }
  A->cols = N;
  if (op == OP_ID) {
    A->rows = N; A->nnz = N;
    A->rowptr = malloc((N + 1) * sizeof(I)); A->colind = malloc(N * sizeof(I)); A->rval = malloc(N * sizeof(T));
    for (I g = 0; g < N; ++g) { A->rowptr[g] = g; A->colind[g] = g; A->rval[g] = (T)1; }
    A->rowptr[N] = N;
    sp_finish_csc(A);
    return;
  }
  I rows = 0;
  for (int q = 0; q < nblk; ++q) rows += N / n[dirs[q]] * (n[dirs[q]] - 1);
  A->rows = rows; A->nnz = 2 * rows;
  A->rowptr = malloc((rows + 1) * sizeof(I)); A->colind = malloc(2 * rows * sizeof(I)); A->rval = malloc(2 * rows * sizeof(T));
  I r = 0;
  for (int q = 0; q < nblk; ++q) {
    int a = dirs[q];
    for (I k = 0; k < n[2]; ++k) for (I j = 0; j < n[1]; ++j) for (I i = 0; i < n[0]; ++i) {
      I c = a == 0 ? i : (a == 1 ? j : k);
      if (c >= n[a] - 1) continue;
      I g = i + n[0] * (j + n[1] * k);
      A->rowptr[r] = 2 * r;
      A->colind[2 * r] = g; A->rval[2 * r] = -ih[a];
      A->colind[2 * r + 1] = g + st[a]; A->rval[2 * r + 1] = ih[a];
      ++r;
    }
  }
  A->rowptr[rows] = 2 * rows;
  sp_finish_csc(A);
}

/* s = A*x: per row the products in ascending column order (== Julia's column sweep order) */
static void sp_mul(const Sp* A, const T* x, T* y) {
#pragma omp parallel for schedule(static)
  for (I r = 0; r < A->rows; ++r) {
    T acc = 0;
    for (I k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) acc = acc + A->rval[k] * x[A->colind[k]];
    y[r] = acc;
  }
}
/* t = A'*v: per column the products in ascending row order */
static void sp_mul_adj(const Sp* A, const T* v, T* y) {
#pragma omp parallel for schedule(static)
  for (I j = 0; j < A->cols; ++j) {
    T acc = 0;
    for (I k = A->colptr[j]; k < A->colptr[j + 1]; ++k) acc = acc + A->cval[k] * v[A->rowind[k]];
    y[j] = acc;
  }
}

/* ---- BLAS-1 style passes (separate sweeps, like the reference's BLAS calls) ---- */
static double sumsq(const T* x, I n) { double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (I i = 0; i < n; ++i) s += (double)x[i] * (double)x[i];
  return s; }
static T nrm2(const T* x, I n) { return (T)sqrt(sumsq(x, n)); }
static T dotp(const T* x, const T* y, I n) { double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (I i = 0; i < n; ++i) s += (double)x[i] * (double)y[i];
  return (T)s; }
static T asum(const T* x, I n) { double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (I i = 0; i < n; ++i) s += fabs((double)x[i]);
  return (T)s; }
static T nrm2_diff(const T* a, const T* b, I n) { double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (I i = 0; i < n; ++i) { T d = a[i] - b[i]; s += (double)d * (double)d; }
  return (T)sqrt(s); }

/* ---- CDS ---- */
typedef struct { I N; int d; I off[32]; T* R; } Cds;

/* AtA = A'A accumulated in ascending row order of A, stored as CDS (mat2CDS of the product;
 * src/PARSDMM_precompute_distribute.jl:44-59, src/mat2CDS.jl:7-32) */
static void ata_cds(const Sp* A, Cds* C) {
  I N = A->cols;
  C->N = N; C->d = 0;
  for (I j = 0; j < N; ++j)
    for (I kk = A->colptr[j]; kk < A->colptr[j + 1]; ++kk) {
      I k = A->rowind[kk];
      for (I ii = A->rowptr[k]; ii < A->rowptr[k + 1]; ++ii) {
        I off = j - A->colind[ii];       /* entry (i,j): row i, offset j-i */
        int f = 0;
        for (int b = 0; b < C->d; ++b) if (C->off[b] == off) f = 1;
        if (!f) { if (C->d >= 32) { fprintf(stderr, "too many bands\n"); exit(1); } C->off[C->d++] = off; }
      }
    }
  for (int a = 0; a < C->d; ++a) for (int b = a + 1; b < C->d; ++b) if (C->off[b] < C->off[a]) { I t = C->off[a]; C->off[a] = C->off[b]; C->off[b] = t; }
  C->R = calloc((size_t)N * C->d, sizeof(T));
  for (I j = 0; j < N; ++j)
    for (I kk = A->colptr[j]; kk < A->colptr[j + 1]; ++kk) {
      I k = A->rowind[kk];
      T akj = A->cval[kk];
      for (I ii = A->rowptr[k]; ii < A->rowptr[k + 1]; ++ii) {
        I i = A->colind[ii];
        I off = j - i;
        int b = 0; while (C->off[b] != off) ++b;
        C->R[(size_t)b * N + i] = C->R[(size_t)b * N + i] + A->rval[ii] * akj;
      }
    }
}

/* y = A x: fill! then one threaded sweep per diagonal (CDS_MVp_MT) */
static void cds_mvp(const Cds* Q, const T* x, T* y) {
  I N = Q->N;
#pragma omp parallel for schedule(static)
  for (I r = 0; r < N; ++r) y[r] = 0;
  for (int i = 0; i < Q->d; ++i) {
    I d = Q->off[i];
    I r0 = d < 0 ? -d : 0, r1 = d > 0 ? N - d : N;
    const T* R = Q->R + (size_t)i * N;
#pragma omp parallel for schedule(static)
    for (I r = r0; r < r1; ++r) y[r] = y[r] + R[r] * x[r + d];
  }
}

/* ---- l1-ball projection, sort based (project_l1_Duchi!.jl) ---- */
static void radix_sort_desc(T* u, T* tmp, I n) {     /* non-negative floats: bit pattern is monotone */
  if (sizeof(T) == 4) {
    uint32_t *a = (uint32_t*)u, *b = (uint32_t*)tmp;
    for (int pass = 0; pass < 4; ++pass) {
      I cnt[257] = {0};
      int sh = pass * 8;
      for (I i = 0; i < n; ++i) cnt[((a[i] >> sh) & 255) + 1]++;
      for (int k = 0; k < 256; ++k) cnt[k + 1] += cnt[k];
      for (I i = 0; i < n; ++i) b[cnt[(a[i] >> sh) & 255]++] = a[i];
      uint32_t* t = a; a = b; b = t;
    }
  } else {
    uint64_t *a = (uint64_t*)u, *b = (uint64_t*)tmp;
    for (int pass = 0; pass < 8; ++pass) {
      I cnt[257] = {0};
      int sh = pass * 8;
      for (I i = 0; i < n; ++i) cnt[((a[i] >> sh) & 255) + 1]++;
      for (int k = 0; k < 256; ++k) cnt[k + 1] += cnt[k];
      for (I i = 0; i < n; ++i) b[cnt[(a[i] >> sh) & 255]++] = a[i];
      uint64_t* t = a; a = b; b = t;
    }
  }
  for (I i = 0; i < n / 2; ++i) { T t = u[i]; u[i] = u[n - 1 - i]; u[n - 1 - i] = t; }   /* ascending -> descending */
}
static T acc_pairwise(T* c, const T* v, T s, I i1, I n) {   /* Julia Base._accumulate_pairwise! */
  if (n < 128) {
    T s_ = v[i1];
    c[i1] = s + s_;
    for (I i = i1 + 1; i < i1 + n; ++i) { s_ = s_ + v[i]; c[i] = s + s_; }
    return s_;
  }
  I n2 = n >> 1;
  T s_ = acc_pairwise(c, v, s, i1, n2);
  s_ = s_ + acc_pairwise(c, v, s + s_, i1 + n2, n - n2);
  return s_;
}
static void project_l1(T* v, I n, T b, T* u, T* sv) {
  if (asum(v, n) <= b) return;
#pragma omp parallel for schedule(static)
  for (I i = 0; i < n; ++i) u[i] = (T)fabs((double)v[i]);
  radix_sort_desc(u, sv, n);
  sv[0] = u[0];
  if (n > 1) acc_pairwise(sv, u, u[0], 1, n - 1);
  I rho = 0;
  while (rho + 1 < n && u[rho] > ((sv[rho] - b) / (T)(rho + 1))) rho++;   /* project_l1_Duchi!.jl:42-44 */
  if (rho < 1) rho = 1;
  T theta = (sv[rho - 1] - b) / (T)rho;
  if (theta < 0) theta = 0;
#pragma omp parallel for schedule(static)
  for (I i = 0; i < n; ++i) {
    T a = (T)fabs((double)v[i]) - theta;
    if (a < 0) a = 0;
    v[i] = v[i] > 0 ? a : (v[i] < 0 ? -a : v[i]);
  }
}

typedef struct { int op, proj; T pmin, pmax; Sp A; Cds AtA; I M; } Set;

static void apply_prox(const Set* S, T* v, I n, T rho, const T* m, T* w1, T* w2) {
  if (S->proj == PJ_BOUNDS) {
#pragma omp parallel for schedule(static)
    for (I i = 0; i < n; ++i) { T t = v[i] < S->pmax ? v[i] : S->pmax; v[i] = S->pmin > t ? S->pmin : t; }
  } else if (S->proj == PJ_L1) {
    project_l1(v, n, S->pmax, w1, w2);
  } else {   /* prox_l2s!: Float64 division */
#pragma omp parallel for schedule(static)
    for (I i = 0; i < n; ++i) v[i] = (T)((double)(v[i] * rho + m[i]) / ((double)rho + 1.0));
  }
}

static double jl_max(const double* a, I n) { double m = -INFINITY; for (I i = 0; i < n; ++i) { if (isnan(a[i])) return NAN; if (a[i] > m) m = a[i]; } return m; }

static void bb_rule(T hl, T nH, T nlh, T ndl, T nG, T gl, int adj_rho, int adj_gamma, T* rho, T* gamma) {
  const T safeguard = sizeof(T) == 8 ? (T)1e-10 : (T)1e-6, epsc = (T)0.3;
  int ar = 0, br = 0, ac = 0, bc = 0; T acor = 0, bcor = 0, ah = 0, bh = 0;
  if ((nH * nlh) > safeguard && (nH * nH) > safeguard && hl > safeguard) { ar = 1; acor = hl / (nH * nlh); }
  if ((nG * ndl) > safeguard && (nG * nG) > safeguard && gl > safeguard) { br = 1; bcor = gl / (nG * ndl); }
  if (ar && acor > epsc) { ac = 1; T mg = hl / (nH * nH), sd = (nlh * nlh) / hl; ah = ((T)2 * mg) > sd ? mg : sd - mg / (T)2; }
  if (br && bcor > epsc) { bc = 1; T mg = gl / (nG * nG), sd = (ndl * ndl) / gl; bh = ((T)2 * mg) > sd ? mg : sd - mg / (T)2; }
  if (adj_rho) { if (ac && bc) *rho = (T)sqrt((double)(ah * bh)); else if (ac) *rho = ah; else if (bc) *rho = bh; }
  if (adj_gamma) {
    if (ac && bc) *gamma = (T)1 + (((T)2 * (T)sqrt((double)(ah * bh))) / (ah + bh));
    else if (ac) *gamma = (T)1.9; else if (bc) *gamma = (T)1.1; else *gamma = (T)1.5;
  }
}

/* The solver: src/PARSDMM.jl:25-258 (serial CDS path, zero initial guess). */
int parsdmm_port(int ndim, const int64_t* n_in, const double* h, int pp, const int* set_op, const int* set_proj,
                 const double* pmin, const double* pmax, const T* m, int maxit, double evol_rel_tol_, double feas_tol_,
                 double obj_tol_, double rho_ini, double gamma_ini, int freq, int adjust_rho, int adjust_gamma,
                 int adjust_feas_rho, int nthreads, T* x_out, double* log_obj, double* log_rpri_total,
                 int64_t* log_cg, double* log_rho, double* log_gamma, double* log_feas, double* log_evol,
                 int* n_iter, int* n_feas_rows, double* loop_seconds) {
  if (nthreads > 0) omp_set_num_threads(nthreads);
  I n[3] = {n_in[0], n_in[1], ndim > 2 ? n_in[2] : 1};
  T ih[3] = {(T)1 / (T)h[0], (T)1 / (T)h[1], ndim > 2 ? (T)1 / (T)h[2] : 0};
  const I N = n[0] * n[1] * n[2];
  const int p = pp + 1;
  const T evol_rel_tol = (T)evol_rel_tol_, feas_tol = (T)feas_tol_, obj_tol = (T)obj_tol_;
  Set* S = calloc(p, sizeof(Set));
  I maxM = N;
  for (int i = 0; i < p; ++i) {
    S[i].op = i < pp ? set_op[i] : OP_ID;
    S[i].proj = i < pp ? set_proj[i] : PJ_DIST;
    S[i].pmin = i < pp ? (T)pmin[i] : 0; S[i].pmax = i < pp ? (T)pmax[i] : 0;
    build_op(&S[i].A, S[i].op, ndim, n, ih);
    ata_cds(&S[i].A, &S[i].AtA);
    S[i].M = S[i].A.rows;
    if (S[i].M > maxM) maxM = S[i].M;
  }
  T *rho = malloc(p * sizeof(T)), *gamma = malloc(p * sizeof(T)), *rho_log = malloc(p * sizeof(T));
  for (int i = 0; i < p; ++i) { rho[i] = (T)rho_ini; gamma[i] = (T)gamma_ini; }
  /* Q, first-seen offsets (PARSDMM_initialize.jl:216-230) */
  Cds Q; Q.N = N; Q.d = 0;
  for (int i = 0; i < p; ++i) {
    for (int b = 0; b <= S[i].AtA.d; ++b) {
      I o = b < S[i].AtA.d ? S[i].AtA.off[b] : 0;
      int f = 0; for (int c = 0; c < Q.d; ++c) if (Q.off[c] == o) f = 1;
      if (!f) Q.off[Q.d++] = o;
    }
  }
  Q.R = calloc((size_t)N * Q.d, sizeof(T));
  for (int i = 0; i < p; ++i)
    for (int b = 0; b < S[i].AtA.d; ++b) {
      int c = 0; while (Q.off[c] != S[i].AtA.off[b]) ++c;
      T* q = Q.R + (size_t)c * N; const T* a = S[i].AtA.R + (size_t)b * N; T r = rho[i];
#pragma omp parallel for schedule(static)
      for (I g = 0; g < N; ++g) q[g] = q[g] + r * a[g];
    }
#define VEC(nm) T** nm = malloc(p * sizeof(T*)); for (int i = 0; i < p; ++i) nm[i] = calloc(S[i].M, sizeof(T));
  VEC(y) VEC(l) VEC(y_old) VEC(l_old) VEC(sv) VEC(x_hat) VEC(r_pri) VEC(y_0) VEC(l_0) VEC(s_0) VEC(l_hat_0) VEC(l_hat)
  T *x = calloc(N, sizeof(T)), *x_old = calloc(N, sizeof(T)), *rhs = calloc(N, sizeof(T)), *tmpN = calloc(N, sizeof(T));
  T *r = calloc(N, sizeof(T)), *pv = calloc(N, sizeof(T)), *Ap = calloc(N, sizeof(T));
  T *w1 = calloc(maxM, sizeof(T)), *w2 = calloc(maxM, sizeof(T)), *tmpM = calloc(maxM, sizeof(T));
  double *rpri_row = calloc(p, sizeof(double)), *feas_prev = calloc(pp > 0 ? pp : 1, sizeof(double));
  /* initial feasibility (PARSDMM_initialize.jl:97-99) */
  for (int i = 0; i < pp; ++i) {
    sp_mul(&S[i].A, m, tmpM);
    memcpy(x_hat[i], tmpM, S[i].M * sizeof(T));
    apply_prox(&S[i], x_hat[i], S[i].M, 0, NULL, w1, w2);
    log_feas[i] = (double)(nrm2_diff(x_hat[i], tmpM, S[i].M) / (nrm2(tmpM, S[i].M) + (T)100 * eps_T()));
  }
  *n_iter = maxit; *n_feas_rows = 2; *loop_seconds = 0;
  if (pp > 0 && jl_max(log_feas, pp) < (double)feas_tol) {
    memcpy(x_out, m, N * sizeof(T)); *n_iter = 1; *n_feas_rows = 1; goto done;
  }
  int counter = 2, ind_ref = maxit;
  T tol_ref = 1;
  double t0 = now();
  for (int it = 1; it <= maxit; ++it) {
    /* rhs_compose.jl:24-31 */
#pragma omp parallel for schedule(static)
    for (I g = 0; g < N; ++g) rhs[g] = 0;
    for (int i = 0; i < p; ++i) {
      T rr = rho[i]; T *yy = y[i], *ll = l[i];
#pragma omp parallel for schedule(static)
      for (I k = 0; k < S[i].M; ++k) tmpM[k] = rr * yy[k] + ll[k];
      sp_mul_adj(&S[i].A, tmpM, tmpN);
#pragma omp parallel for schedule(static)
      for (I g = 0; g < N; ++g) rhs[g] = rhs[g] + tmpN[g];
    }
    memcpy(x_old, x, N * sizeof(T));
    /* argmin_x.jl:33-39 + cg.jl */
    int64_t cg_it = 0; {
      cds_mvp(&Q, x, Ap);
      T nb = nrm2(rhs, N);
      T nres = nrm2_diff(Ap, rhs, N);
      double cand = 0.1 * (double)nres / (double)nb, fl = (double)((T)10 * eps_T());
      cand = (isnan(cand)) ? NAN : (cand > fl ? cand : fl);
      T tol = it < 3 ? (T)cand : (T)((isnan(cand) || isnan((double)tol_ref)) ? NAN : (cand < (double)tol_ref ? cand : (double)tol_ref));
      tol_ref = tol;
      if (nb == 0) { memset(x, 0, N * sizeof(T)); cg_it = 0; }
      else {
        cds_mvp(&Q, x, Ap);                                   /* r = b - A(x): the reference's second SpMV */
#pragma omp parallel for schedule(static)
        for (I g = 0; g < N; ++g) { r[g] = rhs[g] - Ap[g]; pv[g] = r[g]; }
        double ss = sumsq(r, N);
        T rr = (T)ss;
        if ((T)sqrt(ss) / nb <= tol) cg_it = 1;
        else for (int k = 1; k <= 1000; ++k) {
          cg_it = k;
          cds_mvp(&Q, pv, Ap);
          T gam = rr;                                         /* dot(r,z) */
          T alpha = gam / dotp(pv, Ap, N);
          if ((isinf((double)alpha) && alpha > 0) || alpha < 0) break;
#pragma omp parallel for schedule(static)
          for (I g = 0; g < N; ++g) x[g] = x[g] + alpha * pv[g];
#pragma omp parallel for schedule(static)
          for (I g = 0; g < N; ++g) r[g] = r[g] - alpha * Ap[g];
          ss = sumsq(r, N); rr = (T)ss;
          if ((T)sqrt(ss) / nb <= tol) break;
          T beta = rr / gam;
#pragma omp parallel for schedule(static)
          for (I g = 0; g < N; ++g) pv[g] = r[g] + beta * pv[g];
        }
      }
    }
    log_cg[it - 1] = cg_it;
    /* update_y_l.jl:36-101 */
    T sp_tot = 0;
    for (int i = 0; i < p; ++i) {
      const I M = S[i].M; T rr = rho[i], r1 = (T)1 / rho[i], g_ = gamma[i];
      T *yy = y[i], *ll = l[i], *ss_ = sv[i], *xh = x_hat[i], *rp = r_pri[i];
      memcpy(y_old[i], yy, M * sizeof(T)); memcpy(l_old[i], ll, M * sizeof(T));
      sp_mul(&S[i].A, x, ss_);
      if (g_ == (T)1) {
#pragma omp parallel for schedule(static)
        for (I k = 0; k < M; ++k) yy[k] = ss_[k] - ll[k] * r1;
        apply_prox(&S[i], yy, M, rr, m, w1, w2);
#pragma omp parallel for schedule(static)
        for (I k = 0; k < M; ++k) { rp[k] = -ss_[k] + yy[k]; ll[k] = ll[k] + rr * rp[k]; }
      } else {
        T omg = (T)1 - g_;
#pragma omp parallel for schedule(static)
        for (I k = 0; k < M; ++k) { xh[k] = g_ * ss_[k] + omg * yy[k]; yy[k] = xh[k] - ll[k] * r1; }
        apply_prox(&S[i], yy, M, rr, m, w1, w2);
#pragma omp parallel for schedule(static)
        for (I k = 0; k < M; ++k) { rp[k] = -ss_[k] + yy[k]; ll[k] = ll[k] + rr * (-xh[k] + yy[k]); }
      }
      T nr = nrm2(rp, M);
      rpri_row[i] = nr;
      sp_tot = i == 0 ? nr : sp_tot + nr;
#pragma omp parallel for schedule(static)
      for (I k = 0; k < M; ++k) xh[k] = yy[k] - y_old[i][k];
      sp_mul_adj(&S[i].A, xh, tmpN);                          /* r_dual (log only) */
      (void)nrm2(tmpN, N);
      if (it % 10 == 0 && i < pp) {
        memcpy(xh, ss_, M * sizeof(T));
        apply_prox(&S[i], xh, M, 0, NULL, w1, w2);
        log_feas[(size_t)(counter - 1) * pp + i] = (double)(nrm2_diff(xh, ss_, M) / (nrm2(ss_, M) + (T)100 * eps_T()));
      }
    }
    if (it % 10 == 0) counter++;
    log_rpri_total[it - 1] = sp_tot;
    { T nd = nrm2_diff(x, m, N); log_obj[it - 1] = (double)((T)0.5 * (nd * nd)); }
    log_evol[it - 1] = (double)(nrm2_diff(x_old, x, N) / nrm2(x, N));
    for (int i = 0; i < p; ++i) { log_rho[(size_t)(it - 1) * p + i] = rho[i]; log_gamma[(size_t)(it - 1) * p + i] = gamma[i]; rho_log[i] = rho[i]; }
    /* stop_PARSDMM.jl:23-52 */
    int stop = 0;
    if (it > 6 && pp > 0 && jl_max(log_feas + (size_t)(counter - 2) * pp, pp) < (double)feas_tol) {
      double mx = -INFINITY; int nanf = 0;
      for (int k = it - 6; k < it; ++k) { T a = (T)log_obj[k], b = (T)log_obj[k - 1]; T v = (T)fabs((double)((a - b) / b)); if (isnan((double)v)) nanf = 1; if (v > mx) mx = v; }
      if (!nanf && mx < (double)obj_tol) stop = 1;
    }
    if (it > 5 && jl_max(log_evol + (it - 6), 6) < (double)evol_rel_tol) stop = 1;
    if (it > 20 && adjust_rho) {
      int lo = it - 50 > 1 ? it - 50 : 1;
      if (log_rpri_total[it - 1] > jl_max(log_rpri_total + (lo - 1), it - lo)) { adjust_rho = adjust_feas_rho = adjust_gamma = 0; ind_ref = it; }
    }
    if (!adjust_rho && it > ind_ref + 25) {
      int lo = it - 50 > 1 ? it - 50 : 1; if (ind_ref > lo) lo = ind_ref;
      if (log_rpri_total[it - 1] > jl_max(log_rpri_total + (lo - 1), it - lo)) stop = 1;
    }
    if (stop) { *n_iter = it; *n_feas_rows = counter; break; }
    /* PARSDMM.jl:164-206 */
    if (it == 1)
      for (int i = 0; i < p; ++i) {
        const I M = S[i].M; T rr = rho[i];
#pragma omp parallel for schedule(static)
        for (I k = 0; k < M; ++k) { l_hat[i][k] = l_old[i][k] + rr * (-sv[i][k] + y_old[i][k]); l_hat_0[i][k] = l_hat[i][k]; y_0[i][k] = y[i][k]; s_0[i][k] = sv[i][k]; l_0[i][k] = l[i][k]; }
      }
    if ((adjust_rho || adjust_gamma) && it % freq == 0) {
      for (int i = 0; i < p; ++i) {                           /* adapt_rho_gamma.jl:40-127: 5 vector passes + 6 reductions */
        const I M = S[i].M; T rr = rho[i];
        T *d1 = w1, *d2 = w2;
#pragma omp parallel for schedule(static)
        for (I k = 0; k < M; ++k) { l_hat[i][k] = l_old[i][k] + rr * (-sv[i][k] + y_old[i][k]); d1[k] = l_hat[i][k] - l_hat_0[i][k]; d2[k] = sv[i][k] - s_0[i][k]; }
        T hl = dotp(d2, d1, M), nH = nrm2(d2, M), nlh = nrm2(d1, M);
#pragma omp parallel for schedule(static)
        for (I k = 0; k < M; ++k) { d1[k] = l[i][k] - l_0[i][k]; d2[k] = -(y[i][k] - y_0[i][k]); }
        T ndl = nrm2(d1, M), nG = nrm2(d2, M), gl = dotp(d2, d1, M);
        bb_rule(hl, nH, nlh, ndl, nG, gl, adjust_rho, adjust_gamma, &rho[i], &gamma[i]);
      }
      if (it > 1)
        for (int i = 0; i < p; ++i) { const I M = S[i].M; memcpy(l_hat_0[i], l_hat[i], M * sizeof(T)); memcpy(y_0[i], y[i], M * sizeof(T)); memcpy(s_0[i], sv[i], M * sizeof(T)); memcpy(l_0[i], l[i], M * sizeof(T)); }
    }
    if (adjust_feas_rho && it % 10 == 0 && it > 10 && pp > 0) {
      const double* row = log_feas + (size_t)(counter - 2) * pp; int arg = 0, fn = 0;
      for (int k = 0; k < pp && !fn; ++k) { if (isnan(row[k])) { arg = k; fn = 1; } else if (row[k] > row[arg]) arg = k; }
      rho[arg] = (T)2 * rho[arg];
    }
    for (int i = 0; i < p; ++i) { T v = rho[i] < (T)1e4 ? rho[i] : (T)1e4; rho[i] = v > (T)1e-2 ? v : (T)1e-2; }
    for (int i = 0; i < p; ++i) {                             /* Q_update!.jl:45-48 */
      if (rho[i] == rho_log[i]) continue;
      T alpha = rho[i] - rho_log[i];
      for (int b = 0; b < S[i].AtA.d; ++b) {
        int c = 0; while (Q.off[c] != S[i].AtA.off[b]) ++c;
        T* q = Q.R + (size_t)c * N; const T* a = S[i].AtA.R + (size_t)b * N;
#pragma omp parallel for schedule(static)
        for (I g = 0; g < N; ++g) q[g] = q[g] + alpha * a[g];
      }
    }
    if (it == maxit) { *n_iter = it; *n_feas_rows = counter; }
  }
  *loop_seconds = now() - t0;
  memcpy(x_out, x, N * sizeof(T));
done:
  for (int i = 0; i < p; ++i) {
    sp_free(&S[i].A); free(S[i].AtA.R);
    free(y[i]); free(l[i]); free(y_old[i]); free(l_old[i]); free(sv[i]); free(x_hat[i]); free(r_pri[i]);
    free(y_0[i]); free(l_0[i]); free(s_0[i]); free(l_hat_0[i]); free(l_hat[i]);
  }
  free(y); free(l); free(y_old); free(l_old); free(sv); free(x_hat); free(r_pri); free(y_0); free(l_0); free(s_0); free(l_hat_0); free(l_hat);
  free(S); free(Q.R); free(rho); free(gamma); free(rho_log); free(x); free(x_old); free(rhs); free(tmpN); free(r); free(pv); free(Ap);
  free(w1); free(w2); free(tmpM); free(rpri_row); free(feas_prev);
  return 0;
}
