/* The PHASE-LEVEL entry points of include/sipx.h driven from plain C in the order the reference's main loop calls what
 * they replace (src/PARSDMM.jl:97-254):
 *     rhs_compose -> argmin_x -> update_y_l -> log_scalars -> [stop_PARSDMM] -> adapt_rho_gamma -> [rho rules] -> q_update
 * with the scalar rules restated here on the host exactly as a Julia shim that keeps PARSDMM.jl's own loop would keep them
 * (julia/SipxPARSDMM.jl, second half).  The result must equal sipx_parsdmm() -- the same loop run natively -- BIT FOR BIT:
 * x, every y_i and l_i, and every log column.  Run for Float64 and Float32 (argv[1] = "f32").  Exit code 0 = pass. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sipx.h"

#define CHECK(call)                                                            \
  do {                                                                         \
    if ((call) != 0) {                                                         \
      fprintf(stderr, "%s failed: %s\n", #call, sipx_last_error());            \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static int F32 = 0;
/* a value of the working precision TF, carried in a double (sipx.h: per-set scalars cross the ABI as double) */
static double tf(double v) { return F32 ? (double)(float)v : v; }

static void alloc_log(sipx_log* lg, int maxit, int p) {
  memset(lg, 0, sizeof(*lg));
  lg->set_feasibility = calloc((size_t)maxit * p, sizeof(double));
  lg->r_dual = calloc((size_t)maxit * p, sizeof(double));
  lg->r_pri = calloc((size_t)maxit * p, sizeof(double));
  lg->r_dual_total = calloc(maxit, sizeof(double));
  lg->r_pri_total = calloc(maxit, sizeof(double));
  lg->obj = calloc(maxit, sizeof(double));
  lg->evol_x = calloc(maxit, sizeof(double));
  lg->rho = calloc((size_t)maxit * p, sizeof(double));
  lg->gamma = calloc((size_t)maxit * p, sizeof(double));
  lg->cg_it = calloc(maxit, sizeof(int64_t));
  lg->cg_relres = calloc(maxit, sizeof(double));
}

/* Julia maximum(): NaN-propagating */
static double jmax(const double* v, int n) {
  double m = -INFINITY;
  for (int i = 0; i < n; ++i) {
    if (isnan(v[i])) return NAN;
    if (v[i] > m) m = v[i];
  }
  return m;
}

enum { NX = 24, NZ = 16, N = NX * NZ, P = 3, PP = 2 };

static int build(sipx_ctx** out, const void* m, double l1_radius) {
  const int64_t n[2] = {NX, NZ};
  const double h[2] = {25.0, 6.0};
  const double rho_ini[1] = {10.0};
  sipx_ctx* ctx = NULL;
  CHECK(sipx_create(&ctx, F32 ? SIPX_F32 : SIPX_F64, 2, n, h, 0));
  sipx_set_desc d;
  memset(&d, 0, sizeof(d));
  d.op = SIPX_OP_IDENTITY; d.proj = SIPX_PROJ_BOUNDS; d.pmin = 1600.0; d.pmax = 3900.0;
  if (sipx_add_set(ctx, &d, NULL, NULL, 0) != 0) { fprintf(stderr, "add_set: %s\n", sipx_last_error()); return 1; }   /* index 0 */
  memset(&d, 0, sizeof(d));
  d.op = SIPX_OP_TV; d.proj = SIPX_PROJ_L1; d.pmax = l1_radius;
  if (sipx_add_set(ctx, &d, NULL, NULL, 0) != 1) { fprintf(stderr, "add_set: %s\n", sipx_last_error()); return 1; }   /* index 1 */
  double feas0[PP];
  CHECK(sipx_finalize(ctx, m, rho_ini, 1, 1.0, 0, 1, NULL, NULL, NULL, feas0));
  *out = ctx;
  return 0;
}

int main(int argc, char** argv) {
  F32 = argc > 1 && strcmp(argv[1], "f32") == 0;
  const size_t w = F32 ? 4 : 8;
  const int maxit = 70;
  const int64_t rows_tv = (NX - 1) * NZ + NX * (NZ - 1);
  const int64_t rows[P] = {N, rows_tv, N};
  /* model */
  double md[N];
  float mf[N];
  unsigned s = 12345u;
  for (int i = 0; i < N; ++i) {
    s = s * 1664525u + 1013904223u;
    md[i] = 1500.0 + 2500.0 * (double)(i / NX) / (NZ - 1) + 300.0 * ((double)(s >> 8) / 16777216.0 - 0.5);
    mf[i] = (float)md[i];
    if (F32) md[i] = (double)mf[i];
  }
  const void* m = F32 ? (const void*)mf : (const void*)md;
  double radius;
  {
    const int64_t n[2] = {NX, NZ};
    const double h[2] = {25.0, 6.0};
    sipx_ctx* tmp = NULL;
    CHECK(sipx_create(&tmp, F32 ? SIPX_F32 : SIPX_F64, 2, n, h, 0));
    void* tv = malloc(w * 2 * N);
    CHECK(sipx_apply_op(tmp, SIPX_OP_TV, m, tv));
    double a = 0;
    for (int64_t i = 0; i < rows_tv; ++i) a += fabs(F32 ? (double)((float*)tv)[i] : ((double*)tv)[i]);
    radius = 0.5 * a;
    free(tv);
    sipx_destroy(tmp);
  }
  sipx_options opt = {maxit, 1e-3, 5e-2, 1e-3, 2, 1, 1, 1};
  if (F32) { opt.evol_rel_tol = (float)1e-3; opt.feas_tol = (float)5e-2; opt.obj_tol = (float)1e-3; }   /* convert_options! */

  /* ---- (B) whole solve natively ---- */
  sipx_ctx* cb = NULL;
  if (build(&cb, m, radius)) return 1;
  sipx_log lb;
  alloc_log(&lb, maxit, P);
  CHECK(sipx_parsdmm(cb, &opt, &lb));
  void* xb = malloc(w * N);
  void *yb[P], *lbv[P];
  for (int i = 0; i < P; ++i) { yb[i] = malloc(w * rows[i]); lbv[i] = malloc(w * rows[i]); }
  CHECK(sipx_download(cb, xb, lbv, yb));
  sipx_destroy(cb);

  /* ---- (A) the same loop over the phase entry points ---- */
  sipx_ctx* ca = NULL;
  if (build(&ca, m, radius)) return 1;
  sipx_log la;
  alloc_log(&la, maxit, P);
  {
    double feas0[PP];   /* row 1 of set_feasibility = the initial feasibility (PARSDMM_initialize.jl:236) -- recomputed by a twin context */
    sipx_ctx* t2 = NULL;
    const int64_t n[2] = {NX, NZ};
    const double h[2] = {25.0, 6.0}, rho_ini[1] = {10.0};
    CHECK(sipx_create(&t2, F32 ? SIPX_F32 : SIPX_F64, 2, n, h, 0));
    sipx_set_desc d;
    memset(&d, 0, sizeof(d));
    d.op = SIPX_OP_IDENTITY; d.proj = SIPX_PROJ_BOUNDS; d.pmin = 1600.0; d.pmax = 3900.0;
    if (sipx_add_set(t2, &d, NULL, NULL, 0) < 0) return 1;
    memset(&d, 0, sizeof(d));
    d.op = SIPX_OP_TV; d.proj = SIPX_PROJ_L1; d.pmax = radius;
    if (sipx_add_set(t2, &d, NULL, NULL, 0) < 0) return 1;
    CHECK(sipx_finalize(t2, m, rho_ini, 1, 1.0, 0, 1, NULL, NULL, NULL, feas0));
    sipx_destroy(t2);
    for (int k = 0; k < PP; ++k) la.set_feasibility[k] = feas0[k];
  }
  double rho[P], gamma[P], rho_new[P], rpri[P], rdual[P], feas[PP];
  for (int k = 0; k < P; ++k) { rho[k] = tf(10.0); gamma[k] = tf(1.0); }
  int adjust_rho = 1, adjust_gamma = 1, adjust_feas_rho = 1;
  const int freq = opt.rho_update_frequency;
  int counter = 2, ind_ref = maxit, n_iter = maxit;
  double tol_ref = 1.0;
  for (int i = 1; i <= maxit; ++i) {
    CHECK(sipx_rhs_compose(ca, rho));                                                   /* PARSDMM.jl:101 */
    int64_t cg_it; double relres; int flag;
    CHECK(sipx_argmin_x(ca, i, &tol_ref, &cg_it, &relres, &flag));                      /* :106-107 */
    la.cg_it[i - 1] = cg_it; la.cg_relres[i - 1] = relres;
    int flags = 0;
    if (i % 10 == 0) flags |= SIPX_YL_FEAS;
    if (i == 1) flags |= SIPX_YL_FIRST;
    if ((adjust_rho || adjust_gamma) && i % freq == 0) flags |= SIPX_YL_BB;
    CHECK(sipx_update_y_l(ca, i, flags, rho, gamma, rpri, rdual, feas));                /* :133 */
    double sd = tf(rdual[0]), sp = tf(rpri[0]);
    for (int k = 0; k < P; ++k) {
      la.r_pri[(size_t)(i - 1) * P + k] = rpri[k];
      la.r_dual[(size_t)(i - 1) * P + k] = rdual[k];
      if (k > 0) { sd = tf(sd + tf(rdual[k])); sp = tf(sp + tf(rpri[k])); }
      la.rho[(size_t)(i - 1) * P + k] = rho[k];
      la.gamma[(size_t)(i - 1) * P + k] = gamma[k];
    }
    la.r_dual_total[i - 1] = sd; la.r_pri_total[i - 1] = sp;                             /* :134,138 */
    if (i % 10 == 0) {
      for (int k = 0; k < PP; ++k) la.set_feasibility[(size_t)(counter - 1) * PP + k] = feas[k];
      counter += 1;
    }
    CHECK(sipx_log_scalars(ca, &la.obj[i - 1], &la.evol_x[i - 1]));                      /* :140,145 */
    /* ---- stop_PARSDMM.jl:23-52 ---- */
    int stop = 0;
    if (i > 6) {
      const double* row = la.set_feasibility + (size_t)(counter - 2) * PP;
      if (jmax(row, PP) < opt.feas_tol) {
        double mx = -INFINITY; int nan = 0;
        for (int k = i - 6; k < i; ++k) {
          const double a = tf(la.obj[k]), b = tf(la.obj[k - 1]);
          const double v = fabs(tf(tf(a - b) / b));
          if (isnan(v)) nan = 1;
          if (v > mx) mx = v;
        }
        if (!nan && mx < opt.obj_tol) stop = 1;
      }
    }
    if (i > 5 && jmax(la.evol_x + (i - 6), 6) < opt.evol_rel_tol) stop = 1;
    if (i > 20 && adjust_rho) {
      const int lo = i - 50 > 1 ? i - 50 : 1;
      if (la.r_pri_total[i - 1] > jmax(la.r_pri_total + (lo - 1), i - lo)) {
        adjust_rho = adjust_feas_rho = adjust_gamma = 0;
        ind_ref = i;
      }
    }
    if (!adjust_rho && i > ind_ref + 25) {
      int lo = i - 50 > 1 ? i - 50 : 1;
      if (ind_ref > lo) lo = ind_ref;
      if (la.r_pri_total[i - 1] > jmax(la.r_pri_total + (lo - 1), i - lo)) stop = 1;
    }
    if (stop) { n_iter = i; break; }
    /* ---- adjust rho and gamma (PARSDMM.jl:163-227) ---- */
    memcpy(rho_new, rho, sizeof(rho));
    if ((adjust_rho || adjust_gamma) && i % freq == 0) CHECK(sipx_adapt_rho_gamma(ca, adjust_rho, adjust_gamma, rho_new, gamma));
    if (adjust_feas_rho && i % 10 == 0 && i > 10) {                                      /* :213-223 */
      const double* row = la.set_feasibility + (size_t)(counter - 2) * PP;
      int arg = 0, found_nan = 0;
      for (int k = 0; k < PP && !found_nan; ++k) {
        if (isnan(row[k])) { arg = k; found_nan = 1; }
        else if (row[k] > row[arg]) arg = k;
      }
      rho_new[arg] = tf(tf(2.0) * tf(rho_new[arg]));
    }
    for (int k = 0; k < P; ++k) {                                                        /* :226 */
      double r = tf(rho_new[k]);
      r = r < tf(1e4) ? r : tf(1e4);
      rho_new[k] = r > tf(1e-2) ? r : tf(1e-2);
    }
    CHECK(sipx_q_update(ca, rho_new, rho));                                              /* :230-243 */
    memcpy(rho, rho_new, sizeof(rho));
  }
  void* xa = malloc(w * N);
  void *ya[P], *lav[P];
  for (int i = 0; i < P; ++i) { ya[i] = malloc(w * rows[i]); lav[i] = malloc(w * rows[i]); }
  CHECK(sipx_download(ca, xa, lav, ya));
  sipx_destroy(ca);

  /* ---- bit for bit ---- */
  int bad = 0;
  if (n_iter != lb.n_iter || counter != lb.n_feas_rows) { fprintf(stderr, "iterations %d vs %d, feasibility rows %d vs %d\n", n_iter, lb.n_iter, counter, lb.n_feas_rows); return 1; }
  bad |= memcmp(xa, xb, w * N) != 0;
  for (int i = 0; i < P; ++i) bad |= memcmp(ya[i], yb[i], w * rows[i]) != 0 || memcmp(lav[i], lbv[i], w * rows[i]) != 0;
  if (bad) { fprintf(stderr, "x / y / l differ between the phase-level loop and sipx_parsdmm\n"); return 1; }
#define SAME(f, cnt) if (memcmp(la.f, lb.f, sizeof(*la.f) * (size_t)(cnt)) != 0) { fprintf(stderr, "log column " #f " differs\n"); return 1; }
  SAME(obj, n_iter) SAME(evol_x, n_iter) SAME(r_pri, n_iter * P) SAME(r_dual, n_iter * P) SAME(r_pri_total, n_iter) SAME(r_dual_total, n_iter)
  SAME(rho, n_iter * P) SAME(gamma, n_iter * P) SAME(cg_it, n_iter) SAME(cg_relres, n_iter) SAME(set_feasibility, counter * PP)
  printf("phase-level loop == sipx_parsdmm bit for bit (%s): %d iterations, %d feasibility rows, obj %.9e\n", F32 ? "Float32" : "Float64",
         n_iter, counter, la.obj[n_iter - 1]);
  printf("OK\n");
  return 0;
}
