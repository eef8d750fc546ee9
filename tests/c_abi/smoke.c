/* A host program in plain C driving libsipx.so through include/sipx.h only (no Python, no torch, no HIP headers):
 * what a cgo / ccall / JNI binding does.  Projects a 24 x 16 model onto {bounds} and checks the closed form
 * (single set with the identity operator: PARSDMM(m) = clip(m), test/test_PARSDMM.jl:192-242 pattern), then runs
 * {bounds, l1 on TV} and checks the log bookkeeping.  Exit code 0 = pass. */
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

int main(void) {
  const int64_t n[2] = {24, 16};
  const double h[2] = {25.0, 6.0};
  const int N = 24 * 16;
  double* m = malloc(sizeof(double) * N);
  double* x = malloc(sizeof(double) * N);
  unsigned s = 12345u;
  for (int i = 0; i < N; ++i) {
    s = s * 1664525u + 1013904223u;
    m[i] = 1500.0 + 2500.0 * (double)(i / 24) / 15.0 + 300.0 * ((double)(s >> 8) / 16777216.0 - 0.5);
  }
  const double rho_ini[1] = {10.0};
  sipx_options opt = {200, 1e-8, 1e-8, 1e-8, 2, 1, 1, 1};

  /* ---- one set, identity operator: the projection is the clip ---- */
  sipx_ctx* ctx = NULL;
  CHECK(sipx_create(&ctx, SIPX_F64, 2, n, h, 0));
  sipx_set_desc d;
  memset(&d, 0, sizeof(d));
  d.op = SIPX_OP_IDENTITY; d.proj = SIPX_PROJ_BOUNDS; d.pmin = 1600.0; d.pmax = 3900.0;
  if (sipx_add_set(ctx, &d, NULL, NULL, 0) < 0) { fprintf(stderr, "add_set: %s\n", sipx_last_error()); return 1; }
  double feas0[1];
  CHECK(sipx_finalize(ctx, m, rho_ini, 1, 1.0, 0, 1, NULL, NULL, NULL, feas0));
  int p = 0, pp = 0;
  CHECK(sipx_num_terms(ctx, &p, &pp));
  if (p != 2 || pp != 1) { fprintf(stderr, "terms %d %d\n", p, pp); return 1; }
  sipx_log lg;
  alloc_log(&lg, opt.maxit, p);
  CHECK(sipx_parsdmm(ctx, &opt, &lg));
  CHECK(sipx_download(ctx, x, NULL, NULL));
  double err = 0, nrm = 0;
  for (int i = 0; i < N; ++i) {
    const double c = m[i] < 1600.0 ? 1600.0 : (m[i] > 3900.0 ? 3900.0 : m[i]);
    err += (x[i] - c) * (x[i] - c);
    nrm += c * c;
  }
  printf("single set: %d iterations, ||x - clip(m)|| / ||clip(m)|| = %.3e\n", lg.n_iter, sqrt(err / nrm));
  if (!(sqrt(err / nrm) < 1e-6)) return 1;
  sipx_destroy(ctx);

  /* ---- {bounds, l1 on TV}: log bookkeeping ---- */
  CHECK(sipx_create(&ctx, SIPX_F64, 2, n, h, 0));
  if (sipx_add_set(ctx, &d, NULL, NULL, 0) < 0) return 1;
  int64_t rows = 0;
  sipx_set_desc t;
  memset(&t, 0, sizeof(t));
  t.op = SIPX_OP_TV; t.proj = SIPX_PROJ_L1; t.pmax = 1.0;          /* radius fixed below from ||TV m||_1 */
  {
    double* tv = malloc(sizeof(double) * 2 * N);
    sipx_ctx* tmp = NULL;
    CHECK(sipx_create(&tmp, SIPX_F64, 2, n, h, 0));
    CHECK(sipx_apply_op(tmp, SIPX_OP_TV, m, tv));
    rows = (24 - 1) * 16 + 24 * (16 - 1);
    double a = 0;
    for (int64_t i = 0; i < rows; ++i) a += fabs(tv[i]);
    t.pmax = 0.5 * a;
    sipx_destroy(tmp);
    free(tv);
  }
  if (sipx_add_set(ctx, &t, NULL, NULL, 0) < 0) { fprintf(stderr, "add_set: %s\n", sipx_last_error()); return 1; }
  int64_t r1 = 0;
  CHECK(sipx_set_rows(ctx, 1, &r1));
  if (r1 != rows) { fprintf(stderr, "rows %lld vs %lld\n", (long long)r1, (long long)rows); return 1; }
  double feas2[2];
  CHECK(sipx_finalize(ctx, m, rho_ini, 1, 1.0, 0, 1, NULL, NULL, NULL, feas2));
  opt.maxit = 60; opt.evol_rel_tol = 1e-3; opt.feas_tol = 5e-2; opt.obj_tol = 1e-3;
  sipx_log l2;
  alloc_log(&l2, opt.maxit, 3);
  CHECK(sipx_parsdmm(ctx, &opt, &l2));
  printf("two sets: %d iterations, obj %.6e, first cg_it %lld, evol_x[0] is %s\n", l2.n_iter, l2.obj[l2.n_iter - 1],
         (long long)l2.cg_it[0], isnan(l2.evol_x[0]) ? "NaN" : "finite");
  if (l2.n_iter < 2 || l2.cg_it[0] != 0 || !isnan(l2.evol_x[0]) || !(l2.obj[l2.n_iter - 1] > 0)) return 1;
  /* an error path: unknown projector kind must come back as a message, not a crash */
  sipx_set_desc bad;
  memset(&bad, 0, sizeof(bad));
  bad.proj = 99;
  sipx_ctx* c3 = NULL;
  CHECK(sipx_create(&c3, SIPX_F32, 2, n, h, 0));
  if (sipx_add_set(c3, &bad, NULL, NULL, 0) >= 0) { fprintf(stderr, "bad descriptor accepted\n"); return 1; }
  printf("error path: %s\n", sipx_last_error());
  sipx_destroy(c3);
  sipx_destroy(ctx);
  printf("OK\n");
  return 0;
}
