// extern "C" surface of libsipx.so (see include/sipx.h).  Exceptions never cross the ABI:
// every entry returns 0 / non-zero and leaves the text in sipx_last_error().
#include <string>

#include "comm.h"
#include "engine.h"

struct sipx_ctx {
  sipx::EngineBase* e;
};

static thread_local std::string g_err;

#define SIPX_TRY(body)                   \
  try {                                  \
    body;                                \
    return 0;                            \
  } catch (const std::exception& ex) {   \
    g_err = ex.what();                   \
    return 1;                            \
  } catch (...) {                        \
    g_err = "unknown error";             \
    return 1;                            \
  }

extern "C" {

const char* sipx_last_error(void) { return g_err.c_str(); }

int sipx_create(sipx_ctx** out, int dtype, int ndim, const int64_t* n, const double* h, int device) {
  SIPX_TRY({
    if (!out) throw std::runtime_error("null output handle");
    *out = nullptr;
    sipx::EngineBase* e = sipx::make_engine(dtype, ndim, n, h, device);
    *out = new sipx_ctx{e};
  })
}
void sipx_destroy(sipx_ctx* ctx) {
  if (!ctx) return;
  try {
    delete ctx->e;
  } catch (...) {
  }
  delete ctx;
}
int sipx_add_set(sipx_ctx* c, const sipx_set_desc* d, const void* R, const int64_t* off, int d_i) {
  try {
    return c->e->add_set(d, R, off, d_i);
  } catch (const std::exception& ex) {
    g_err = ex.what();
    return -1;
  } catch (...) {
    g_err = "unknown error";
    return -1;
  }
}
int sipx_set_rows(sipx_ctx* c, int set, int64_t* rows) { SIPX_TRY(*rows = c->e->set_rows(set)) }
int sipx_num_terms(sipx_ctx* c, int* p, int* pp) { SIPX_TRY(c->e->num_terms(p, pp)) }
int sipx_finalize(sipx_ctx* c, const void* m, const double* rho_ini, int n_rho, double gamma_ini, int feasibility_only,
                  int zero_ini_guess, const void* x0, const void* const* l0, const void* const* y0,
                  double* feasibility_initial) {
  SIPX_TRY(c->e->finalize(m, rho_ini, n_rho, gamma_ini, feasibility_only, zero_ini_guess, x0, l0, y0,
                          feasibility_initial))
}
int sipx_reset(sipx_ctx* c, const void* m, const double* rho_ini, int n_rho, double gamma_ini, int zero_ini_guess, const void* x0,
               const void* const* l0, const void* const* y0, double* feasibility_initial) {
  SIPX_TRY(c->e->reset(m, rho_ini, n_rho, gamma_ini, zero_ini_guess, x0, l0, y0, feasibility_initial))
}
int sipx_rhs_compose(sipx_ctx* c, const double* rho) { SIPX_TRY(c->e->rhs_compose(rho)) }
int sipx_argmin_x(sipx_ctx* c, int it, double* tol_ref_io, int64_t* cg_it, double* cg_relres, int* cg_flag) {
  SIPX_TRY(c->e->argmin_x(it, tol_ref_io, cg_it, cg_relres, cg_flag))
}
int sipx_update_y_l(sipx_ctx* c, int it, int flags, const double* rho, const double* gamma, double* r_pri,
                    double* r_dual, double* feas) {
  SIPX_TRY(c->e->update_y_l(it, flags, rho, gamma, r_pri, r_dual, feas))
}
int sipx_log_scalars(sipx_ctx* c, double* obj, double* evol_x) { SIPX_TRY(c->e->log_scalars(obj, evol_x)) }
int sipx_adapt_rho_gamma(sipx_ctx* c, int adjust_rho, int adjust_gamma, double* rho_io, double* gamma_io) {
  SIPX_TRY(c->e->adapt_rho_gamma(adjust_rho, adjust_gamma, rho_io, gamma_io))
}
int sipx_q_update(sipx_ctx* c, const double* rho_new, const double* rho_old) { SIPX_TRY(c->e->q_update(rho_new, rho_old)) }
int sipx_download(sipx_ctx* c, void* x, void* const* l, void* const* y) { SIPX_TRY(c->e->download(x, l, y)) }
int sipx_warm_start_from(sipx_ctx* fine, sipx_ctx* coarse) {
  SIPX_TRY({
    if (!fine || !coarse) throw std::runtime_error("null context");
    fine->e->warm_start_from(coarse->e);
  })
}
int sipx_parsdmm(sipx_ctx* c, const sipx_options* opt, sipx_log* log) { SIPX_TRY(c->e->parsdmm(opt, log)) }
int sipx_parsdmm_begin(sipx_ctx* c, const sipx_options* opt, sipx_log* log) { SIPX_TRY(c->e->parsdmm_begin(opt, log)) }
int sipx_parsdmm_steps(sipx_ctx* c, int nsteps, int* done) {
  SIPX_TRY({
    bool d = false;
    for (int k = 0; k < nsteps && !d; ++k) d = c->e->parsdmm_step();
    if (done) *done = d ? 1 : 0;
  })
}

int sipx_cds_spmv(int dtype, int64_t N, int d, const void* R, const int64_t* off, const void* x, void* y, int device) {
  SIPX_TRY(sipx::cds_spmv_host(dtype, N, d, R, off, x, y, device))
}
int sipx_resample_nn(int dtype, int ndim, const int64_t* nc, const int64_t* nf, const void* in, void* out, int device) {
  SIPX_TRY(sipx::resample_nn_host(dtype, ndim, nc, nf, in, out, device))
}
int sipx_apply_op(sipx_ctx* c, int op, const void* x, void* s) { SIPX_TRY(c->e->apply_op(op, x, s, false)) }
int sipx_apply_op_adj(sipx_ctx* c, int op, const void* v, void* t) { SIPX_TRY(c->e->apply_op(op, v, t, true)) }
int sipx_project(sipx_ctx* c, const sipx_set_desc* d, void* v, int64_t len) { SIPX_TRY(c->e->project(d, v, len)) }
int sipx_get_Q(sipx_ctx* c, void* Q, int64_t* offsets, int* d) { SIPX_TRY(c->e->get_Q(Q, offsets, d)) }
int sipx_time_spmv(sipx_ctx* c, int reps, double* avg_ms) { SIPX_TRY(*avg_ms = c->e->time_spmv(reps)) }
int sipx_kernel_stats(sipx_ctx* c, int enable, int64_t* launches, double* total_ms) {
  SIPX_TRY(c->e->kernel_stats(enable, launches, total_ms))
}
const char* sipx_kernel_stats_json(sipx_ctx* c, int enable) {
  try {
    return c->e->kernel_stats_json(enable);
  } catch (const std::exception& ex) {
    g_err = ex.what();
    return nullptr;
  } catch (...) {
    g_err = "unknown error";
    return nullptr;
  }
}
int sipx_debug_proj(sipx_ctx* c, int set, int which, double* out16) { SIPX_TRY(c->e->debug_proj(set, which, out16)) }
void* sipx_stream(sipx_ctx* c) { return c->e->stream(); }
void* sipx_dev_rhs(sipx_ctx* c) { return c->e->dev_rhs(); }
void* sipx_dev_x(sipx_ctx* c) { return c->e->dev_x(); }
int sipx_get_rhs(sipx_ctx* c, void* rhs) { SIPX_TRY(c->e->get_rhs(rhs)) }
int sipx_prox_l2s(int dtype, int64_t n, void* x, double rho, const void* m, int device) {
  SIPX_TRY(sipx::prox_l2s_host(dtype, n, x, rho, m, device))
}
int sipx_set_owned(sipx_ctx* c, const int32_t* owned) { SIPX_TRY(c->e->set_owned(owned)) }
int sipx_rccl_unique_id(void* id128) { SIPX_TRY(sipx::rccl_unique_id(id128)) }
int sipx_set_comm_rccl(sipx_ctx* c, const void* id128, int world, int rank) {
  SIPX_TRY({
    c->e->bind_device();      // ncclCommInitRank binds the communicator to the current device
    c->e->set_comm(sipx::make_rccl_comm(id128, world, rank));
  })
}
int sipx_set_comm(sipx_ctx* c, const sipx_comm* comm) {
  SIPX_TRY({
    if (!comm) throw std::runtime_error("null communicator");
    c->e->set_comm(sipx::make_callback_comm(comm));
  })
}
int sipx_comm_info(sipx_ctx* c, int* nranks, int* rank, char* version, int version_len, int* decomposition) {
  SIPX_TRY(c->e->comm_info(nranks, rank, version, version_len, decomposition))
}
int sipx_device_bytes(sipx_ctx* c, int64_t* context_bytes, int64_t* device_used, int64_t* device_total) {
  SIPX_TRY(c->e->device_bytes(context_bytes, device_used, device_total))
}
int sipx_slab(sipx_ctx* c, int64_t* row0, int64_t* row1, int64_t* chunk) { SIPX_TRY(c->e->slab(row0, row1, chunk)) }
int sipx_set_q_mode(sipx_ctx* c, int mode) { SIPX_TRY(c->e->set_q_mode(mode)) }
int sipx_set_decomp(sipx_ctx* c, int mode) { SIPX_TRY(c->e->set_decomp(mode)) }
int sipx_apply_Q(sipx_ctx* c, const void* x, void* y) { SIPX_TRY(c->e->apply_Q(x, y)) }

}  // extern "C"
