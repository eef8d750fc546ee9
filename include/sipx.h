/*
 * sipx.h -- C ABI of the MI355X-native PARSDMM projection engine (libsipx.so).
 *
 * This is the drop-in boundary for ONE hot path of slimgroup/SetIntersectionProjection.jl:
 * the PARSDMM iteration body behind
 *     PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options[, x, l, y]) -> (x, log, l, y)
 * (reference src/PARSDMM.jl:25-35,257).  Plain pointers and sizes only; every function that
 * returns int returns 0 on success and non-zero on error (text via sipx_last_error()) -- with ONE
 * exception, sipx_add_set, which returns the index (>= 0) of the set it added and -1 on error.
 * Host arrays are
 * copied in / copied out; the library never keeps a caller pointer after a call returns
 * (reference ownership model: Julia owns every array).
 *
 * Two granularities, B built on A:
 *   A. phase level  -- one entry per step of the reference's main loop, so PARSDMM.jl's own
 *      loop, @timeit sections and stop_PARSDMM can stay in Julia and ccall each phase;
 *   B. whole solve  -- sipx_parsdmm() runs the restated loop natively.
 *
 * A handle is single-threaded (one solve at a time); different handles may live on
 * different host threads / GPUs.  All vectors are TF = float or double as chosen at
 * sipx_create().  Per-set scalars cross the ABI as double (exact for float values).
 */
#ifndef SIPX_H
#define SIPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sipx_ctx sipx_ctx;

enum { SIPX_F32 = 0, SIPX_F64 = 1 };

/* Transform-domain operator A_i, matrix-free.  Replaces the SparseMatrixCSC TD_OP[i] built by
 * get_TD_operator / get_discrete_Grad (src/get_TD_operator.jl:12-95, src/get_discrete_Grad.jl:16-76);
 * the shim maps set_Prop.tag[i][2] ("identity","D_x","D_y","D_z","TV") + comp_grid.n/.d to it.
 * 2-D grids: D_x = dim 1, D_z = dim 2 (pass n3 = 1); TV = [D_z; D_x] (2-D), [D_z; D_y; D_x] (3-D). */
enum { SIPX_OP_IDENTITY = 0, SIPX_OP_DX = 1, SIPX_OP_DY = 2, SIPX_OP_DZ = 3, SIPX_OP_TV = 4,
       SIPX_OP_CSC = 5 /* a caller-supplied sparse matrix (constraint.custom_TD_OP[1], setup_constraints.jl:70-72): the
                          SparseMatrixCSC arrays in sipx_set_desc.csc_*, 0-based; AtA must be passed explicitly in CDS */ };

/* Projector descriptor replacing the opaque closure P_sub[i] (src/get_projector.jl:3-103). */
enum {
  SIPX_PROJ_BOUNDS      = 0, /* project_bounds!(x, LB, UB) scalar bounds  (projectors/project_bounds!.jl:3-12)  */
  SIPX_PROJ_BOUNDS_VEC  = 1, /* mode WHOLE: per-element bounds lb/ub TF[M_i]           (project_bounds!.jl:14-25);
                                mode FIBER: per-fiber bounds lb/ub TF[TD_n[dir]]       (project_bounds!.jl:38-88) */
  SIPX_PROJ_L1          = 2, /* project_l1_Duchi!(x, pmax)                (projectors/project_l1_Duchi!.jl:21-52) */
  SIPX_PROJ_L2          = 3, /* project_l2!(x, pmax)                      (projectors/project_l2!.jl:3-16) */
  SIPX_PROJ_ANNULUS     = 4, /* project_annulus!(x, pmin, pmax)           (projectors/project_annulus!.jl:3-21) */
  SIPX_PROJ_CARDINALITY = 5, /* project_cardinality!(x, k = pmax): whole vector, per fiber or per slice
                                (projectors/project_cardinality!.jl:3-146) */
  SIPX_PROJ_PROX_L1     = 6, /* prox_l1!(x, pmax)                         (src/prox_l1!.jl:8-10) */
  SIPX_PROJ_L1_DFT      = 7, /* x -> Re(F' project_l1_Duchi!(F x, pmax)), F = unitary DFT; op must be identity
                                (src/get_projector.jl:29-35 with TD_OP "DFT", src/get_TD_operator.jl:45-47,80-82) */
  SIPX_PROJ_RANK        = 8, /* project_rank!(x, r = pmax): matrix (2-D grid, mode WHOLE) or every slice orthogonal to dir
                                (3-D, mode SLICE)   (projectors/project_rank!.jl:3-48) */
  SIPX_PROJ_NUCLEAR     = 9, /* project_nuclear!(x, sigma = pmax), same modes as RANK (projectors/project_nuclear!.jl:3-62) */
  SIPX_PROJ_HISTOGRAM   = 10,/* project_histogram_relaxed!(x, lb, ub): lb/ub ascending TF[M_i]
                                (projectors/project_histogram_relaxed.jl:9-27) */
  SIPX_PROJ_SUBSPACE    = 11,/* project_subspace!(x, A, orth): whole vector, fibers of a matrix (2-D) or slices of a
                                tensor (3-D)  (projectors/project_subspace!.jl:10-125); op must be identity */
  SIPX_PROJ_BOUNDS_DFT  = 12 /* x -> Re(F' (ub .* (F x))), F = unitary DFT: bounds in the Fourier domain with a binary lower
                                bound of zeros and a mask ub TF[N] in the transform's element order
                                (project_bounds!.jl:27-36 under get_projector.jl:8-9 with TD_OP "DFT"); op must be identity */
};
/* constraint.app_mode (set_definitions): ("matrix"|"tensor", _) = WHOLE, ("fiber", d), ("slice", d).
 * dir is the 0-based array dimension: "x" = 0, "y" = 1, "z" = 2 on a 3-D grid, "z" = 1 on a 2-D grid. */
enum { SIPX_MODE_WHOLE = 0, SIPX_MODE_FIBER = 1, SIPX_MODE_SLICE = 2 };
enum { SIPX_TRANSFORM_NONE = 0, SIPX_TRANSFORM_DCT = 1 };

typedef struct {
  int32_t op;        /* SIPX_OP_*   */
  int32_t proj;      /* SIPX_PROJ_* */
  double pmin, pmax; /* constraint[i].min / .max (set_definitions, src/SetIntersectionProjection.jl:142-149) */
  const void* lb;    /* BOUNDS_VEC / HISTOGRAM: host TF vectors (see the kinds above), else NULL */
  const void* ub;
  int32_t ncvx;      /* set_Prop.ncvx[i] (src/setup_constraints.jl:89-97) */
  int32_t reserved;  /* 0 */
  int32_t mode;      /* SIPX_MODE_* */
  int32_t dir;       /* direction of the fibers / normal of the slices */
  const void* basis; /* SUBSPACE: host TF[basis_rows x basis_cols], column-major (constraint.custom_TD_OP[1]) */
  int64_t basis_rows;
  int32_t basis_cols;
  int32_t basis_orth;/* constraint.custom_TD_OP[2]: A'A = I */
  int32_t component; /* Minkowski sets (src/PARSDMM_precompute_distribute_Minkowski.jl:6-173): 0 = ordinary set; 1 = the set
                        constrains the first component u (TD_OP = [A 0]); 2 = the second component v ([0 A]); 3 = their sum
                        u + v ([A A]).  Either every set of a context names a component or none does.  With components the
                        unknown is x = [u; v] (2N entries in x0 / sipx_download / sipx_apply_Q), the distance term is
                        1/2 ||u + v - m||^2, and Q is the 2N x 2N CDS matrix of the reference. */
  const int64_t* csc_colptr; /* SIPX_OP_CSC: colptr[N+1], rowval[nnz] (0-based, ascending inside a column), nzval TF[nnz] */
  const int64_t* csc_rowval;
  const void* csc_nzval;
  int64_t csc_rows;          /* rows of the matrix = length of y_i, l_i */
  int32_t transform; /* SIPX_TRANSFORM_*: an orthogonal transform folded into the projector, x -> A' P(A x) with TD_OP = I
                        (src/get_projector.jl: the branches `constraint.TD_OP in special_operator_list`,
                        src/setup_constraints.jl:54,76-80).  DCT = orthonormal DCT-II along every grid dimension; proj must be
                        BOUNDS, BOUNDS_VEC, L1 or CARDINALITY (mode WHOLE), op the identity.  The DFT has its own kinds above. */
  int32_t pad_;
} sipx_set_desc;

/* PARSDMM_options (src/SetIntersectionProjection.jl:110-128); Blas_active / parallel / FL /
 * x_min_solver / Minkowski have no native meaning (FL = dtype of the context). */
typedef struct {
  int32_t maxit;
  double evol_rel_tol, feas_tol, obj_tol;
  int32_t rho_update_frequency;
  int32_t adjust_rho, adjust_gamma, adjust_feasibility_rho;
} sipx_options;

/* log_type_PARSDMM (src/SetIntersectionProjection.jl:95-108) as flat row-major arrays the caller
 * allocates for `maxit` rows; the executed row counts come back in n_iter / n_feas_rows
 * (output_check_PARSDMM truncation, src/PARSDMM.jl:261-278). */
typedef struct {
  double* set_feasibility; /* [maxit][pp] */
  double* r_dual;          /* [maxit][p]  */
  double* r_pri;           /* [maxit][p]  */
  double* r_dual_total;    /* [maxit] */
  double* r_pri_total;     /* [maxit] */
  double* obj;             /* [maxit] */
  double* evol_x;          /* [maxit] */
  double* rho;             /* [maxit][p] */
  double* gamma;           /* [maxit][p] */
  int64_t* cg_it;          /* [maxit] */
  double* cg_relres;       /* [maxit] */
  double timing_ms[7];     /* the 7 @timeit sections of src/PARSDMM.jl:40,100,105,113,152,163,229 */
  int32_t n_iter;          /* rows filled in the per-iteration logs */
  int32_t n_feas_rows;     /* rows kept in set_feasibility (= `counter`) */
  int32_t stopped_feasible;/* 1 if the input was accepted as feasible (src/PARSDMM.jl:63-82) */
} sipx_log;

const char* sipx_last_error(void);

/* ---- construction (replaces the allocations of PARSDMM_initialize, src/PARSDMM_initialize.jl:117-184) ---- */
int sipx_create(sipx_ctx** out, int dtype, int ndim, const int64_t* n, const double* h, int device);
void sipx_destroy(sipx_ctx* ctx);

/* Adds constraint set i (call in TD_OP order).  ata_R / ata_off / d_i = AtA[i] in CDS (N x d_i,
 * column-major) with set_Prop.AtA_offsets[i] (src/PARSDMM_precompute_distribute.jl:52-59); pass
 * ata_R = NULL to have the bands generated on the device from the operator descriptor.
 * RETURN VALUE (unlike every other entry): the 0-based index of the new set (the i of y_i / l_i, of the rho / gamma /
 * r_pri arrays and of sipx_set_rows) on success, -1 on error -- test `rc < 0`, not `rc != 0`. */
int sipx_add_set(sipx_ctx* ctx, const sipx_set_desc* desc, const void* ata_R, const int64_t* ata_off, int d_i);

/* Rows of A_i (length of y_i / l_i) for set i; i = number of constraint sets addresses the
 * distance term appended by PARSDMM_precompute_distribute (src/PARSDMM_precompute_distribute.jl:17-26). */
int sipx_set_rows(sipx_ctx* ctx, int set, int64_t* rows);
int sipx_num_terms(sipx_ctx* ctx, int* p, int* pp);

/* Uploads m, initial rho/gamma, optional warm start (x0, l0[i], y0[i]; NULL = zeros; ignored when
 * zero_ini_guess != 0, src/PARSDMM_initialize.jl:304-313); appends the distance term unless
 * feasibility_only; assembles Q / Q_offsets (src/PARSDMM_initialize.jl:216-230); fills
 * feasibility_initial[pp] (src/PARSDMM_initialize.jl:97-99). */
int sipx_finalize(sipx_ctx* ctx, const void* m, const double* rho_ini, int n_rho, double gamma_ini,
                  int feasibility_only, int zero_ini_guess, const void* x0, const void* const* l0,
                  const void* const* y0, double* feasibility_initial);

/* The same sets on the same grid, once more: a new model m (and, optionally, another warm start and rho_ini) on a context that
 * has been finalised -- and, usually, solved -- before.  This is how the reference's callers use this entry point: PARSDMM wrapped
 * as a projector and called again and again inside an outer loop (examples/constrained_freq_FWI_simple.jl:468,
 * examples/Constraint_examples_2D.jl:222-223, examples/Dykstra_parallel_vs_PARSDMM.jl:134); every such call of the reference runs
 * PARSDMM_initialize again (src/PARSDMM.jl:58-61).  Here nothing is allocated and no plan, handle, stream or event is created:
 * every array is zero-filled, m uploaded, rho / gamma set as sipx_finalize sets them, Q assembled again (the solve updated it
 * incrementally, src/Q_update!.jl:45-48), every warm start inside the context forgotten, feasibility_initial[pp] taken again
 * (src/PARSDMM_initialize.jl:97-99).  The context is then in the state sipx_finalize leaves it in: a solve on it returns the bits a
 * newly built context returns.  Arguments as for sipx_finalize (feasibility_only stays what it was).  Not available for a rank of a
 * sharded solve. */
int sipx_reset(sipx_ctx* ctx, const void* m, const double* rho_ini, int n_rho, double gamma_ini, int zero_ini_guess,
               const void* x0, const void* const* l0, const void* const* y0, double* feasibility_initial);

/* ---- A. phase level ---- */
/* rhs = sum_i A_i'(rho_i y_i + l_i)                                   (src/rhs_compose.jl:24-36) */
int sipx_rhs_compose(sipx_ctx* ctx, const double* rho);
/* copy!(x_old,x); tolerance rule; warm-started CG on Q x = rhs        (src/PARSDMM.jl:106-107, src/argmin_x.jl:23-39, src/cg.jl:44-128) */
int sipx_argmin_x(sipx_ctx* ctx, int it, double* tol_ref_io, int64_t* cg_it, double* cg_relres, int* cg_flag);
/* y/l update for every local set                                       (src/update_y_l.jl:36-101) */
enum { SIPX_YL_FEAS = 1,  /* also log set feasibility (mod(i,10)==0)            update_y_l.jl:90-99 */
       SIPX_YL_BB = 2,    /* also accumulate the six BB sums and refresh snapshots  adapt_rho_gamma.jl:41-53, PARSDMM.jl:192-206 */
       SIPX_YL_FIRST = 4  /* first iteration: set snapshots                    PARSDMM.jl:164-180 */ };
int sipx_update_y_l(sipx_ctx* ctx, int it, int flags, const double* rho, const double* gamma,
                    double* r_pri, double* r_dual, double* feas /* [pp], written iff SIPX_YL_FEAS */);
/* obj = 1/2||x-m||^2, evol_x = ||x_old-x||/||x||                      (src/PARSDMM.jl:140,145) */
int sipx_log_scalars(sipx_ctx* ctx, double* obj, double* evol_x);
/* Barzilai-Borwein rule from the sums gathered by the last sipx_update_y_l(SIPX_YL_BB)  (src/adapt_rho_gamma.jl:55-126) */
int sipx_adapt_rho_gamma(sipx_ctx* ctx, int adjust_rho, int adjust_gamma, double* rho_io, double* gamma_io);
/* Q += (rho_new - rho_old) AtA_i for changed sets; rebinds the distance prox  (src/Q_update!.jl:45-48, src/PARSDMM.jl:230-243) */
int sipx_q_update(sipx_ctx* ctx, const double* rho_new, const double* rho_old);
/* copy out x, l[i], y[i] (any pointer may be NULL)                     (src/PARSDMM.jl:257) */
int sipx_download(sipx_ctx* ctx, void* x, void* const* l, void* const* y);

/* Multilevel (src/PARSDMM_multi_level.jl:61-83, src/interpolate_y_l.jl:16-94): warm start of a finalized context on a finer
 * grid from a solved context on a coarser one, DEVICE TO DEVICE -- x, every l_i and y_i are resampled (nearest neighbour,
 * the set-by-set block arithmetic of interpolate_y_l) without visiting the host.  Both contexts hold the same sets, in the
 * same precision, on the same device.  Overwrites whatever start sipx_finalize gave the fine context.  Ranks of a sharded
 * solve: both contexts slab-decomposed (sipx_set_decomp); the call is then a collective -- the coarse slabs are all-gathered on
 * the device and every rank resamples the whole iterate. */
int sipx_warm_start_from(sipx_ctx* fine, sipx_ctx* coarse);

/* ---- B. whole solve (src/PARSDMM.jl:97-257 restated natively) ---- */
int sipx_parsdmm(sipx_ctx* ctx, const sipx_options* opt, sipx_log* log);
/* the same solve advanced in pieces: begin, then up to nsteps iterations per call (done = 1 once a stop rule or maxit
 * ended it; the log arrays given to _begin must stay alive) */
int sipx_parsdmm_begin(sipx_ctx* ctx, const sipx_options* opt, sipx_log* log);
int sipx_parsdmm_steps(sipx_ctx* ctx, int nsteps, int* done);

/* ---- kernel-level entry points used by the parity tests and bench (same kernels as above) ---- */
/* y = R x for an arbitrary CDS matrix (CDS_MVp_MT + fill!, src/CDS_MVp_MT.jl:9-25, src/argmin_x.jl:72-78) */
int sipx_cds_spmv(int dtype, int64_t N, int d, const void* R, const int64_t* off, const void* x, void* y, int device);
/* s = A x (op descriptor) and t = A' v on the context grid */
int sipx_apply_op(sipx_ctx* ctx, int op, const void* x, void* s);
int sipx_apply_op_adj(sipx_ctx* ctx, int op, const void* v, void* t);
/* in-place projector on a host vector of length len (src/projectors/project_X.jl) */
int sipx_project(sipx_ctx* ctx, const sipx_set_desc* desc, void* v, int64_t len);
/* nearest-neighbour resampling of an array of shape nc to shape nf, the grid transfer of PARSDMM_multi_level:
 * out[k] = in[round(1 + (k-1)(nc-1)/(nf-1))] per axis (Interpolations.BSpline(Constant()) evaluated on
 * range(1, stop=nc, length=nf); src/PARSDMM_multi_level.jl:41-45,61-65, src/interpolate_y_l.jl:32-88) */
int sipx_resample_nn(int dtype, int ndim, const int64_t* nc, const int64_t* nf, const void* in, void* out, int device);
/* Q as assembled / updated (N x d column-major) and its offsets */
int sipx_get_Q(sipx_ctx* ctx, void* Q, int64_t* offsets, int* d);
/* device-side timing of the dominant kernel: runs cds_spmv on Q `reps` times, returns avg ms (HIP events on the engine stream) */
int sipx_time_spmv(sipx_ctx* ctx, int reps, double* avg_ms);
/* HIP-event timing of the engine's kernels, bracketed on the stream each launch goes to.  launches / total_ms report what
 * was gathered for the product of the CG iteration (cds_spmv fused with the dot product, src/CDS_MVp_MT.jl:9-25 + cg.jl:85-88)
 * since the last call; `enable` then (re)starts the collection: 0 = off, 1 = that kernel only (two event records per CG
 * iteration: cheap enough for a timed region), 2 = EVERY kernel (about 5 us per launch: for a window of its own). */
int sipx_kernel_stats(sipx_ctx* ctx, int enable, int64_t* launches, double* total_ms);
/* The same collection as a JSON text, one entry per kernel that ran: {"mode", "event_pair_overhead_ms", "kernels": [{"name",
 * "launches", "total_ms", "bytes_survey" (SURVEY 8d's algorithmic bytes of the reference function the kernel replaces, summed
 * over the launches), "bytes_moved" (what the kernel has to move at least), "gated" (some launches return at once on a
 * device-side condition), "inclusive" (the interval contains other listed kernels)}]}.  The pointer stays valid until the next
 * call on this context; NULL on error (sipx_last_error).  `enable` as above. */
const char* sipx_kernel_stats_json(sipx_ctx* ctx, int enable);
/* diagnostics: state of the projector-scalar search of set `set` (which = 0: prox, 1: feasibility) as 16 doubles:
 * need, theta, theta_prev, hw, spec_lo, spec_hi, lo, hi, asum, vmax, gathered, overflow, spec_ok, michelot_its, refine, 0 */
int sipx_debug_proj(sipx_ctx* ctx, int set, int which, double* out16);
/* engine stream handle (hipStream_t) so a host harness can order its own work / collectives against it */
void* sipx_stream(sipx_ctx* ctx);
/* device pointers of rhs / x (TF[N]) for in-place collectives on the sharded path.
 * sipx_dev_x: x lives in a ring of three buffers -- every x-step that changes x (sipx_argmin_x, sipx_parsdmm_steps) moves it to
 * another one and leaves the old iterate behind as x_old.  The pointer is therefore valid ONLY UNTIL THE NEXT x-step: ask again
 * after every step, never cache it (a cached pointer names x_old, or the Barzilai-Borwein snapshot, one step later).  Before the
 * first x-step x_old names x itself (the zero / warm start): evol_x of an update taken then is ||x - x|| = 0. */
void* sipx_dev_rhs(sipx_ctx* ctx);
void* sipx_dev_x(sipx_ctx* ctx);
/* rhs as composed by the last sipx_rhs_compose (host TF[N]; TF[2N] in Minkowski mode)           (src/rhs_compose.jl:24-36) */
int sipx_get_rhs(sipx_ctx* ctx, void* rhs);
/* x = (x*rho + m) / (rho + 1.0) on host vectors of length n, through the device function the y/l update applies for the
 * distance term (src/prox_l2s!.jl:3-6: numerator in TF, division in Float64) */
int sipx_prox_l2s(int dtype, int64_t n, void* x, double rho, const void* m, int device);
/* restricts y/l work and rhs contributions to sets with owner[i] != 0 (set sharding); Q stays global */
int sipx_set_owned(sipx_ctx* ctx, const int32_t* owned);

/* ---- sharded solve: one process per GPU ----
 * Replaces the reference's parallel mode (one Julia worker per set, src/PARSDMM.jl:114-131,183-197; the (+) reduction of
 * the partial right-hand sides, src/rhs_compose.jl:17-20; x shipped to every worker, src/update_y_l_parallel.jl:6-90).
 * With a communicator attached (before sipx_finalize) a context is one RANK of the solve:
 *   - the y/l work and the rhs contributions of the sets are split over the ranks (sipx_set_owned; default: set i on rank
 *     i mod world),
 *   - sipx_rhs_compose ends with a reduce-scatter of the partial rhs by z-slab (slabs of ceil(n_last / world) planes),
 *   - sipx_argmin_x runs CG on the rank's slab of rows of Q (one halo plane from each neighbour per product, the dot
 *     products through an all-reduce of the float64 block partials) and ends with an all-gather of x,
 *   - sipx_update_y_l ends with one all-reduce of the packed per-set sums, so every rank returns the r_pri / r_dual /
 *     feasibility of every set and takes the same rho / gamma / stop decisions; sipx_parsdmm runs the whole loop that way.
 *   - a rank / nuclear-norm set on the slices orthogonal to the last grid dimension (project_rank!.jl:23-47 applied slice
 *     by slice) is projected by ALL ranks: the owner scatters v by slab, every rank factorises the slices of its slab, a
 *     gather returns the projected v (the one set of BASELINE config 4 that outweighs all others together).
 * Every rank must make the same sequence of calls.  Q is maintained for the slab rows only (sipx_get_Q refuses), x is
 * complete on every rank, y_i / l_i live on the owner.  Not available with Minkowski components, the stencil form of Q or
 * operators whose A'A reaches further than one plane of the grid.
 *
 * sipx_set_comm_rccl: collectives by RCCL on the engine's own streams.  id128 = the 128-byte ncclUniqueId that rank 0
 * obtained from sipx_rccl_unique_id and the host side handed to every rank (torch.distributed store, MPI, a file ...). */
int sipx_rccl_unique_id(void* id128);
/* sipx_set_decomp (before sipx_finalize, with a communicator attached): how the ranks divide the work.
 *   SIPX_DECOMP_SETS (default): by constraint set, as described above -- the reference's own split.
 *   SIPX_DECOMP_SLAB: by z-slab of the grid, for the WHOLE iteration: every rank holds every set and works on its planes of
 *     the (globally indexed) arrays.  No N-vector crosses the fabric: the right-hand side needs no reduction, x no all-gather
 *     (one halo plane to each neighbour instead), the threshold searches all-reduce their 19 probe sums and all-gather the few
 *     magnitudes inside the final bracket, and every per-set sum goes through the one all-reduce of sipx_update_y_l.  Sets
 *     whose projector only needs sums over the grid -- bounds, l1 / l2 ball, annulus, prox_l1 on the identity or on D_x / D_y /
 *     D_z / TV -- work that way; since round 5 the others are taken too (y_i, l_i on the slabs as well): slice-wise rank /
 *     nuclear norm on z-slices by every rank on its own slices, cardinality by a search over the same collectives, the l1 ball
 *     behind the DFT on a 3-D grid by a slab-decomposed transform (one all-to-all each way), any other projector by an owner
 *     rank on the gathered vector (two fan exchanges for that set).  Refused: caller-supplied sparse operators, Minkowski
 *     contexts, the stencil form of Q.  sipx_set_owned is ignored; sipx_download is then a collective (every rank calls it: it
 *     gathers the slabs of x, y_i, l_i). */
#define SIPX_DECOMP_SETS 0
#define SIPX_DECOMP_SLAB 1
/* the same decomposition with FULL-size arrays on every rank: SIPX_DECOMP_SLAB backs every N-sized array of a rank with memory for
 * its planes and the halo planes around them only (device bytes per rank fall with the number of ranks; the levels of a multilevel
 * solve included since round 5); this mode is the A/B switch and the fallback when mapped memory cannot be exchanged */
#define SIPX_DECOMP_SLAB_FULL 2
int sipx_set_decomp(sipx_ctx* ctx, int mode);
int sipx_set_comm_rccl(sipx_ctx* ctx, const void* id128, int world, int rank);
/* sipx_set_comm: the same operations supplied by the caller.  Each callback enqueues its operation on `stream`
 * (a hipStream_t; or completes it before returning) and returns 0 on success; dtype is SIPX_F32 / SIPX_F64; all buffers are
 * device memory.  In place on `buf` of world * chunk elements:
 *   allreduce_sum:      buf[0 .. count) <- sum over ranks (identical bits on every rank)
 *   reduce_scatter_sum: buf[rank*chunk .. (rank+1)*chunk) <- sum over ranks of that range
 *   allgather:          buf[r*chunk .. (r+1)*chunk) <- that range of rank r, for every r
 *   halo_exchange:      send `count` elements to, and receive as many from, rank prev and rank next (-1 = no neighbour)
 *   scatter / gather:   in place on world * chunk elements: rank r receives buf[r*chunk .. (r+1)*chunk) of rank `root`, or
 *                       rank `root` receives that range of every rank r.  On a rank OTHER than `root` only its own range
 *                       buf[rank*chunk .. (rank+1)*chunk) is memory the callback may touch (round 5: with sparse arrays the rest of
 *                       the exchange layout is not backed there) */
typedef struct {
  void* user;
  int32_t world, rank;
  int (*allreduce_sum)(void* user, void* buf, int64_t count, int32_t dtype, void* stream);
  int (*reduce_scatter_sum)(void* user, void* buf, int64_t chunk, int32_t dtype, void* stream);
  int (*allgather)(void* user, void* buf, int64_t chunk, int32_t dtype, void* stream);
  int (*halo_exchange)(void* user, const void* send_prev, void* recv_prev, int32_t prev, const void* send_next, void* recv_next,
                       int32_t next, int64_t count, int32_t dtype, void* stream);
  int (*scatter)(void* user, void* buf, int64_t chunk, int32_t dtype, int32_t root, void* stream);
  int (*gather)(void* user, void* buf, int64_t chunk, int32_t dtype, int32_t root, void* stream);
} sipx_comm;
int sipx_set_comm(sipx_ctx* ctx, const sipx_comm* comm);
/* What the attached communicator itself reports: the ranks it spans and this rank's number in it (RCCL: ncclCommCount /
 * ncclCommUserRank -- asked of the library, not echoed from sipx_set_comm_rccl's arguments), its version as text ("rccl
 * 2.x.y" from ncclGetVersion, "callbacks (sipx_set_comm)", "none" without a communicator; version_len bytes incl. the
 * terminator), and the decomposition in force (SIPX_DECOMP_*).  So that "did RCCL see N ranks" can be answered from a log. */
int sipx_comm_info(sipx_ctx* ctx, int* nranks, int* rank, char* version, int version_len, int* decomposition);
/* Device memory: bytes this context has allocated (its own arrays and the buffers of its library-backed projectors; library
 * workspaces are not seen), and what the runtime reports for the whole device (used, total).  A slab-decomposed rank holds its
 * planes only (plus halo planes), so context_bytes falls with the number of ranks -- the figure the bench line carries as
 * comm.device_bytes_per_rank.  (No reference counterpart: Julia's GC owns the reference's arrays.) */
int sipx_device_bytes(sipx_ctx* ctx, int64_t* context_bytes, int64_t* device_used, int64_t* device_total);
/* this rank's slab of the x-step: rows [row0, row1) of Q / entries of x, and the elements per rank (chunk) of the padded
 * exchange buffers; without a communicator row0 = 0, row1 = chunk = N */
int sipx_slab(sipx_ctx* ctx, int64_t* row0, int64_t* row1, int64_t* chunk);

/* How the x-step applies Q = sum_i rho_i A_i'A_i.  SIPX_Q_CDS (default): explicit bands in CDS storage, the
 * reference's arithmetic (PARSDMM_initialize.jl:216-230, CDS_MVp_MT.jl:9-25, Q_update!.jl:45-48), (d+2) N w bytes per
 * product.  SIPX_Q_STENCIL ("beyond CDS"): coefficients generated from rho_i, h and the boundary
 * masks, 2 N w bytes per product and no Q_update! traffic; every set must use a descriptor-generated AtA
 * (ata_R = NULL).  Results agree with the CDS mode to rounding, not bit for bit.  Call before sipx_finalize. */
enum { SIPX_Q_CDS = 0, SIPX_Q_STENCIL = 1 };
int sipx_set_q_mode(sipx_ctx* ctx, int mode);
/* y = Q x (host TF[N] in and out) through the kernel the x-step uses, in either mode */
int sipx_apply_Q(sipx_ctx* ctx, const void* x, void* y);

#ifdef __cplusplus
}
#endif
#endif /* SIPX_H */
