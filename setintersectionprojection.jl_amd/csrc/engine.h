// Host-side engine: device-resident PARSDMM state + the phases of the reference's main loop.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sipx.h"
#include "sipx_common.h"

namespace sipx {

struct Comm;

struct EngineBase {
  virtual ~EngineBase() {}
  virtual int add_set(const sipx_set_desc* d, const void* ata_R, const int64_t* ata_off, int d_i) = 0;
  virtual int64_t set_rows(int set) = 0;
  virtual void num_terms(int* p, int* pp) = 0;
  virtual void finalize(const void* m, const double* rho_ini, int n_rho, double gamma_ini, int feasibility_only,
                        int zero_ini_guess, const void* x0, const void* const* l0, const void* const* y0,
                        double* feasibility_initial) = 0;
  virtual void reset(const void* m, const double* rho_ini, int n_rho, double gamma_ini, int zero_ini_guess, const void* x0,
                     const void* const* l0, const void* const* y0, double* feasibility_initial) = 0;
  virtual void rhs_compose(const double* rho) = 0;
  virtual void argmin_x(int it, double* tol_ref_io, int64_t* cg_it, double* cg_relres, int* cg_flag) = 0;
  virtual void update_y_l(int it, int flags, const double* rho, const double* gamma, double* r_pri, double* r_dual,
                          double* feas) = 0;
  virtual void log_scalars(double* obj, double* evol_x) = 0;
  virtual void adapt_rho_gamma(int adjust_rho, int adjust_gamma, double* rho_io, double* gamma_io) = 0;
  virtual void q_update(const double* rho_new, const double* rho_old) = 0;
  virtual void download(void* x, void* const* l, void* const* y) = 0;
  virtual void warm_start_from(EngineBase* coarse) = 0;
  virtual void parsdmm(const sipx_options* opt, sipx_log* log) = 0;
  virtual void parsdmm_begin(const sipx_options* opt, sipx_log* log) = 0;
  virtual bool parsdmm_step() = 0;
  virtual void apply_op(int op, const void* x, void* s, bool adjoint) = 0;
  virtual void project(const sipx_set_desc* d, void* v, int64_t len) = 0;
  virtual void get_Q(void* Q, int64_t* offsets, int* d) = 0;
  virtual void apply_Q(const void* x, void* y) = 0;
  virtual double time_spmv(int reps) = 0;
  virtual void kernel_stats(int enable, int64_t* launches, double* total_ms) = 0;
  virtual const char* kernel_stats_json(int enable) = 0;
  virtual void debug_proj(int set, int which, double* out16) = 0;
  virtual void* stream() = 0;
  virtual void* dev_rhs() = 0;
  virtual void* dev_x() = 0;
  virtual void get_rhs(void* out) = 0;
  virtual void set_owned(const int32_t* owned) = 0;
  virtual void set_q_mode(int mode) = 0;
  virtual void set_decomp(int mode) = 0;
  virtual void set_comm(Comm* c) = 0;       // takes ownership
  virtual void comm_info(int* nranks, int* rank, char* version, int version_len, int* decomposition) = 0;
  virtual void bind_device() = 0;           // makes the context's GPU the calling thread's current device
  virtual void device_bytes(int64_t* context_bytes, int64_t* device_used, int64_t* device_total) = 0;
  virtual void slab(int64_t* row0, int64_t* row1, int64_t* chunk) = 0;
};

EngineBase* make_engine(int dtype, int ndim, const int64_t* n, const double* h, int device);
void resample_nn_host(int dtype, int ndim, const int64_t* nc, const int64_t* nf, const void* in, void* out, int device);
void prox_l2s_host(int dtype, int64_t n, void* x, double rho, const void* m, int device);
void cds_spmv_host(int dtype, int64_t N, int d, const void* R, const int64_t* off, const void* x, void* y, int device);

}  // namespace sipx
