// Shared declarations between the HIP kernels (kernels_*.hip) and the host engine (engine.cpp).
// gfx950 only.  All kernels are memory-bound streaming passes: 16 B per lane vector accesses,
// grid-stride over a fixed grid, float64 block partials -> deterministic second-stage sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>

namespace sipx {

constexpr int BLOCK = 256;        // 4 waves of 64
constexpr int NB = 2048;          // fixed grid of every streaming/reduction kernel (8 blocks per CU: measured +8% over 1024)
// Kernels that need 65..72 VGPRs run 7 workgroups per CU: a grid of 2048 would leave a 256-workgroup tail at 1/7
// occupancy, so those families launch 7*256 workgroups (block_reduce_store clears the partial entries beyond a launch's grid).
constexpr int NB_7 = 1792;
// per_cu workgroups on every compute unit of the current device, within the partial arrays (<= NB_7).
// Grid of a grid-stride kernel over nvec thread-iterations: no more workgroups than there is work for (every workgroup
// pays the reduction epilogue; at 64^3 an eighth of 2048 workgroups has anything to do), at most `cap`.
inline int fit_grid(long long nvec, int cap) {
  const long long need = (nvec + BLOCK - 1) / BLOCK;
  return (int)(need < 1 ? 1 : (need < cap ? need : cap));
}
// the 16-byte vector paths need 16-byte aligned bases (pointers shifted to a slab of rows may not be)
template <typename... P>
inline bool aligned16(const P*... p) {
  return (((reinterpret_cast<uintptr_t>(p) & 15u) == 0) && ...);
}
inline int launch_blocks(int per_cu) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  const int g = per_cu * cus;
  return g < 256 ? 256 : (g > NB_7 ? NB_7 : g);
}
constexpr int MAXD = 32;          // CDS bands held in kernel arguments
constexpr int MAX_SETS = 16;      // sets fused in one rhs_compose launch
constexpr int YL_SLOTS = 13;      // reductions produced by one y/l-update launch
constexpr int L1_K = 8;          // probe thresholds of the l1-ball threshold search
#define SIPX_SOLVE_SLOTS 64      // most workgroups a cooperative k_l1_solve may be launched with
#define SIPX_MAX_WORLD 64        // most ranks of a slab-decomposed solve (per-rank max / min entries in ProjScalars)
constexpr int GATHER_HDR = 8;    // TF elements in front of a rank's segment of gathered magnitudes (read as doubles: count, S_above, C_above)
constexpr int PREP_SLOTS = 3 + 2 * L1_K;   // ||v||_1, ||v||_2^2, nnz, S_k, C_k
// header of a rank's FAST segment (speculative exchange of a slab-decomposed search), in TF elements: the PREP_SLOTS sums,
// overflow flag, largest / smallest non-zero magnitude, count -- as doubles
template <typename T>
constexpr int fast_hdr() { return ((PREP_SLOTS + 4) * 8 / (int)sizeof(T) + 3) / 4 * 4; }
// Sampled prediction of the l1 threshold (k_sample, sample_decide): histogram of the sampled magnitudes over
// SAMPLE_BINS bins of 2^-SAMPLE_MBITS of an octave each (the bin key is the leading bits of the floating-point pattern),
// centred on the predicted theta
constexpr int SAMPLE_BINS = 1024;
constexpr int SAMPLE_MBITS = 7;

// ---- per-kernel statistics (sipx_kernel_stats / sipx_kernel_stats_json) -------------------------------------------------
// Every launcher below opens an ObsScope; while the engine collects statistics it installs an observer that brackets the
// launch with two HIP events on the launch's own stream and books the kernel's ALGORITHMIC bytes twice: `survey` = SURVEY
// 8(d)'s count for the reference function the kernel replaces, `moved` = what this kernel has to move at least (they differ
// where the kernel reads less than the reference's pass structure implies, e.g. the symmetric band read of k_cds).
enum KernelId {
  KID_CDS_SPMV = 0, KID_CDS_DOT, KID_CDS_RESID, KID_CDS_FUSED, KID_SQ_SPMV, KID_SQ_DOT, KID_SQ_RESID, KID_CG_BEGIN, KID_CG_XR, KID_CG_P,
  KID_Q_UPDATE, KID_FIN_SUM, KID_RHS, KID_YL, KID_YL_MULTI, KID_ADJ_NORM, KID_LOG3, KID_PASS_FIRST, KID_PASS_LEAN, KID_PASS_PROBE,
  KID_PASS_COMPACT, KID_PASS_DIST, KID_PASS_STORE, KID_PASS_MULTI, KID_SLOT_SUMS, KID_DECIDE, KID_SAMPLE, KID_L1_SOLVE, KID_GATHER,
  KID_PS_RESCALE, KID_CARD, KID_EXT, KID_BB_RULE, KID_OTHER, KID_COUNT
};
inline const char* kernel_name(int k) {
  static const char* const names[KID_COUNT] = {
      "k_cds<MODE=0>", "k_cds<MODE=1>", "k_cds<MODE=2>", "k_cds_fused", "k_sq<MODE=0>", "k_sq<MODE=1>", "k_sq<MODE=2>", "k_cg_begin",
      "k_cg_update_xr", "k_cg_update_p", "k_q_update", "k_fin_sum", "k_rhs", "k_yl", "k_yl_multi", "k_adj_norm", "k_log3",
      "k_pass<M_FIRST>", "k_pass<M_LEAN>", "k_pass<M_PROBE>", "k_pass<M_COMPACT>", "k_pass<M_DIST>", "k_pass<M_STORE>", "k_pass_multi",
      "k_slot_sums", "k_decide", "k_sample", "k_l1_solve", "k_gather_pack/unpack", "k_ps_rescale", "k_card_*", "ext_proj (library-backed)",
      "k_bb_rule", "other"};
  return (k >= 0 && k < KID_COUNT) ? names[k] : "?";
}
struct LaunchObserver {
  void* user;
  void (*begin)(void* user, int kid, hipStream_t s, double bytes_survey, double bytes_moved);
  void (*end)(void* user, int kid, hipStream_t s);
};
// the observer of the calling host thread (a context is driven by one thread at a time); nullptr: nothing is recorded
const LaunchObserver*& launch_observer();

// A/B switches of the launchers, read from the environment ONCE per context (sipx_finalize -> refresh_env_knobs) instead of by a
// getenv at every launch: a test that sets a switch builds a new context afterwards.
struct EnvKnobs {
  int cds_march = 1;              // SIPX_CDS_MARCH: 0 never, 2 also on grids too small to fill the chip (tests)
  long long cds_march_zchunk = 0; // SIPX_CDS_MARCH_ZCHUNK (with =2)
  long long multi_zchunk = 0;     // SIPX_MULTI_ZCHUNK
  int rhs_march = 1;              // SIPX_RHS_MARCH
  long long rhs_march_zchunk = 0; // SIPX_RHS_MARCH_ZCHUNK
  int trace_kernels = 0;          // SIPX_TRACE_KERNELS=1 (debugging): name every launch on stderr and drain the stream behind it
  int trace_searches = 0;         // SIPX_TRACE_SEARCHES=1: every threshold search of the batched chain that needed its fallback sweeps, on stderr
  int mark_stride = 0;            // SIPX_MARK_STRIDE: section timing marks on iterations 1-4 and every such iteration after them (1: every iteration; 0: by grid size, parsdmm_step)
  int q_plan = 1;                 // SIPX_Q_PLAN=0: the Q update regenerates every band value per element (k_q_update) instead of adding planned products
};
const EnvKnobs& env_knobs();
void refresh_env_knobs();
// device bytes allocated on behalf of the context the calling thread is building (nullptr: not counted)
long long*& alloc_tally();
struct ObsScope {
  const LaunchObserver* o;
  int kid;
  hipStream_t s;
  ObsScope(int kid_, hipStream_t s_, double bytes_survey, double bytes_moved = -1.0) : o(launch_observer()), kid(kid_), s(s_) {
    if (env_knobs().trace_kernels) fprintf(stderr, "[sipx trace] %s\n", kernel_name(kid));
    if (o) o->begin(o->user, kid, s, bytes_survey, bytes_moved < 0 ? bytes_survey : bytes_moved);
  }
  ~ObsScope() {
    if (o) o->end(o->user, kid, s);
    if (env_knobs().trace_kernels) {       // SIPX_TRACE_KERNELS=1: every launch named and drained -- a faulting kernel is the last one named
      (void)hipStreamSynchronize(s);
      fprintf(stderr, "[sipx trace]   done\n");
    }
  }
  ObsScope(const ObsScope&) = delete;
  ObsScope& operator=(const ObsScope&) = delete;
};

// reduction slots of k_yl (per set)
enum { SL_RPRI = 0, SL_DY = 1, SL_HL = 2, SL_HH = 3, SL_LH = 4, SL_DL = 5, SL_GG = 6, SL_GL = 7,
       SL_FE = 8, SL_SS = 9, SL_OBJ = 10, SL_EVO = 11, SL_XX = 12,
       SL_ADJ = 13 /* ||A'(y - y_old)||^2 of a difference operator */, SL_FE2 = 14, SL_SS2 = 15 /* two-pass feasibility */ };
constexpr int SET_SLOTS = 16;     // reduction slots per set in the engine's partial array: slot s of set i at ((i * SET_SLOTS + s) * NB)

enum { F_FEAS = 1, F_BB = 2, F_FIRST = 4, F_NOSPEC = 8 /* skip the speculative gather of the l1 search (not set by the engine at present) */,
       F_STORE_DY = 16 /* identity-shaped pass over a materialised s = A x (custom sparse operator): keep y - y_old */ };

// internal prox kinds (public SIPX_PROJ_* plus the distance term)
enum { PX_BOUNDS = 0, PX_BOUNDS_VEC = 1, PX_L1 = 2, PX_L2 = 3, PX_ANNULUS = 4, PX_CARD = 5, PX_PROX_L1 = 6,
       PX_EXT = 8,                   // acts on a materialised vector (ext_proj.h): y arrives precomputed (vsrc == 2)
       PX_DIST = 100 };

struct Grid {
  long long n[3];     // n1 (fastest), n2, n3 (1 for 2-D)
  long long N;        // n1*n2*n3
  long long st[3];    // strides 1, n1, n1*n2
  // Division of a linear index (< 2^31) by n1 and by n2 as a multiply-high and a shift (Granlund-Montgomery): for
  // 2^(l-1) < d <= 2^l, m = floor(2^(31+l) / d) + 1 gives floor(u / d) = mulhi(u, m) >> (l - 1) for every u < 2^31.
  // m == 0: not set up (grids built ad hoc for one-off kernels): the kernels divide.  The stencil kernels take the
  // coordinates of every vector they touch; a hardware-free 32-bit division costs ~25 VALU instructions each.
  unsigned m1 = 0, s1 = 0, m2 = 0, s2 = 0;
  // Slab decomposition of a sharded solve (every rank works on the planes [e0, e1) of the GLOBALLY indexed arrays): the set
  // kernels sweep the grid points e0 <= g < e1 only (e1 < 0: all N).  k_yl may be given one plane more at the front than the
  // rank owns -- it recomputes the neighbour's last plane of y, l, bit for bit, instead of receiving it -- and adds only the
  // points g >= s0 to its sums.
  long long e0 = 0, e1 = -1, s0 = 0;
  void set_fast_div() {
    auto magic = [](long long d, unsigned& m, unsigned& sh) {
      m = 0; sh = 0;
      if (d < 2 || d >= (1ll << 31)) return;          // d == 1 and oversize: plain division
      int l = 0;
      while ((1ll << l) < d) ++l;
      const unsigned long long q = ((unsigned long long)1 << (31 + l)) / (unsigned long long)d + 1ull;
      m = (unsigned)q;
      sh = (unsigned)(l - 1);
    };
    magic(n[0], m1, s1);
    magic(n[1], m2, s2);
  }
};

inline long long range_len(const Grid& g) { return (g.e1 < 0 ? g.N : g.e1) - g.e0; }     // grid points a launch sweeps

struct CdsArgs {
  int d;
  long long off[MAXD];
  // Q = sum rho_i A_i'A_i is symmetric bit for bit (Q[r, r-o] and Q[r-o, r] are the same products summed in the same
  // order), so a band with a negative offset never has to come from HBM: its entry at row r is the entry of the partner
  // band (+|o|) at row r-|o|, which the sweep loaded a moment ago.  sym = 1: read negative bands through `partner`.
  int sym = 0;
  int partner[MAXD] = {};
  // z-marching product (k_cds_march): set by the engine when Q is the 7-band matrix of a 3-D grid, {0, +-1, +-n1, +-n1n2},
  // read through the symmetric partners.  march = 0: not applicable; 1: bands in the order 0, -1, +1, -n1, +n1, -n1n2, +n1n2
  // (identity set first, then D_x, D_y, D_z); 2: 0, -n1n2, -n1, -1, +1, +n1, +n1n2 (identity set, then TV).  The summation
  // order of a row is the band order, so each order has its own instantiation.  mb[q]: band index of offset 0, +1, +n1, +n1n2.
  int march = 0;
  int mb[4] = {0, 0, 0, 0};
  long long gn[3] = {0, 0, 0};
};

// Q = sum_i rho_i A_i'A_i as stencil coefficients (sipx_set_q_mode(SIPX_Q_STENCIL)): w0 = sum of rho over identity
// sets, w[dir] = sum of rho_i / h_dir^2 over sets differencing along dir, mask = directions present.
template <typename T>
struct StencilQ {
  T w0, w[3];
  int mask;
};

// Scalars of the non-elementwise projectors, produced on the device and consumed by k_yl.
// The l1 part persists across PARSDMM iterations: the thresholds probed during the first pass
// are centred on the previous iteration's theta (warm start).
template <typename T>
struct ProjScalars {
  double asum;        // ||v||_1
  double sumsq;       // ||v||_2^2
  T vmax;             // max |v|
  T vmin;             // smallest non-zero |v| (only meaningful when every entry is active, see k_l1_solve)
  int need;           // 1: outside the set, threshold/scale below is active
  T theta;            // l1: soft threshold
  T scale;            // l2 / annulus: multiplier
  int fill;           // annulus zero-vector case: fill with `scale`
  // l1 search state
  double t[L1_K];     // probe thresholds (ascending; +inf = unused)
  double lo, hi;      // bracket: lo <= theta* < hi;  elements in (lo, hi] are compacted
  int refine;         // > 0: bracket still holds too many elements -> one more probe pass (the number of the coming round)
  int rounds_used;    // refinement rounds this search went through (the host sizes the next search's rounds by it, slab-decomposed)
  double theta_prev;  // last non-zero theta (centre of the next probe)
  unsigned long long n_compact;
  // speculative compaction fused into the first pass: magnitudes in (spec_lo, spec_hi] are gathered
  // while v is produced; valid when the final bracket lies inside that range
  double spec_lo, spec_hi;
  double s_above, c_above;   // (sum, count) of |v| > spec_hi from the probe
  double hw;                 // relative half-width of the next speculative range
  int spec_ok, spec_overflow;
  double dbg[4];             // diagnostics of the last search: gathered count, overflow, spec_ok, Michelot iterations
  // cardinality (keep the k largest magnitudes; ties at the threshold by lowest index)
  T tau;              // k-th largest magnitude
  long long quota;    // entries equal to tau are kept iff their padded index <= quota (idx cut)
  long long* cidx;    // device buffer for the indices of the gathered magnitudes
  double c_lo, c_hi;  // counts of |v| > lo and |v| > hi of the current bracket
  // l1 search: the two PROBES that bracket theta* (f(br_tl) >= 0 > f(br_th)) with their exact sums and counts, carried from
  // one decision to the next so that refinement rounds only ever shrink the bracket (br_Cl < 0: count not known)
  double br_tl, br_Sl, br_Cl, br_th, br_Sh, br_Ch;
  T tau_prev;
  // sums of the PREP_SLOTS partial slots of the last probe pass, its largest / smallest non-zero magnitude (k_slot_sums)
  // On a slab-decomposed grid ONE all-reduce (sum) over red, ovf and mm makes them global: mm holds (largest, smallest
  // non-zero) magnitude per rank in the rank's own two entries and zeros elsewhere -- keep the three adjacent
  double red[PREP_SLOTS];
  double ovf;                                // > 0: the speculative gather of some rank overflowed its LDS buffers
  double mm[2 * SIPX_MAX_WORLD];
  int gather_overflow;                       // this search: the ranks gathered more magnitudes than the exchange segments hold (theta = NaN, the host is told)
  int lean;           // the coming first pass evaluates the two edge probes of the speculative range only (k_pass M_FIRST)
  // cooperative sweeps of k_l1_solve: per-workgroup shares of (sum hi, sum lo, count), double buffered by iteration parity
  double coop_hi[2][SIPX_SOLVE_SLOTS], coop_lo[2][SIPX_SOLVE_SLOTS], coop_c[2][SIPX_SOLVE_SLOTS];
  unsigned coop_arrive, coop_finish, coop_abort;
  // sampled prediction: per bin, count and fixed-point sum of the sampled magnitudes packed in one word (integer atomics:
  // the totals do not depend on the order of arrival); zero between searches
  unsigned long long hist[SAMPLE_BINS];
  unsigned pass_ticket;   // arrivals of the workgroups of k_slot_sums (one rank: the last one takes the scalar decision)
  unsigned samp_ticket;   // arrivals of k_sample's workgroups (the last one decides)
  int rescaled, resc_bad;   // the coming search follows k_ps_rescale; the last rescaled prediction missed its range
  int want_sample;    // the prediction of the coming search is not trusted (theta moved, or rho was changed): sample first
  int sampled;        // the probes of the coming first pass were centred by the sampled prediction (diagnostics: dbg_sampled)
  int dbg_sampled;
  double samp_theta;  // the sampled estimate itself
  // ... minus the exact theta of the search it preceded: the error of a sampled estimate is dominated by WHICH entries are sampled,
  // and they are the same from one iteration to the next, so it persists (+0.007 ... +0.005 over twelve iterations of the headline
  // run while theta went from 8.6 to 0.41) -- the next sampled estimate is corrected by it
  double samp_bias;
  int samp_bias_ok;
  double samp_lo, samp_hi, samp_c;   // diagnostics: Newton / secant bounds of the sample's root, active sample count
};

// Optional sampled prediction in front of an l1 search (see k_sample): host-side switch and the pinned word through which
// k_l1_solve tells the host whether the coming search wants it (ProjScalars::want_sample)
// Collectives of a threshold search on a slab-decomposed grid (every rank sweeps its planes; the engine supplies them)
struct ChainHooks {
  int world = 1, rank = 0;
  void* user = nullptr;
  void (*allreduce_sum)(void* user, double* buf, size_t count, hipStream_t s) = nullptr;            // in place
  void (*allgather)(void* user, void* buf, size_t chunk, int dtype_is_f64, hipStream_t s) = nullptr;  // in place, chunk elements per rank
  void* gbuf = nullptr;          // world * (gcap + GATHER_HDR) TF elements
  long long gcap = 0;            // gathered magnitudes a rank may contribute
  long long fcap = 0;            // ... to the speculative exchange (fast segments: what a settled search gathers is small)
};

struct SampleCtl {
  int enable = 0;
  long long runs = 0;      // sampled runs of 64 grid points (0: 16384, 32768 from 2^26 grid points on)
  int* host_want = nullptr;
  // pinned word of the set (slab-decomposed grid): raised by k_gather_unpack when the magnitudes gathered inside the final
  // bracket over all ranks did not fit the exchange segments -- the host turns it into an error return (never NaN iterates)
  int* host_ovf = nullptr;
  // speculative exchange (slab-decomposed grid): pinned word of the set that k_spec_decide publishes, (seq << 2) | verdict bits
  unsigned* verdict = nullptr;
  unsigned seq = 0;
  int lean_done = 0;       // the set's lean first pass was taken by k_lean_multi: the chain launches the full one only (gated as ever)
  int lean_known = 0;      // the host has read from the set's pinned word that the coming first pass is a lean one
  long long compact_cap = 0;   // elements the set's compaction buffer holds (0: one per entry of the set's vector); what the exchange strings together must fit
};

template <typename T>
struct SetArgs {
  T *y, *l, *dy, *lh0, *y0, *s0, *l0;   // per-set state, padded layout nblk*N (y, l: current = y_old, l_old on entry)
  T *yo, *lo;                           // where the updated y, l go: y/l themselves, or the array holding the snapshot y0/l0
  T* v;                                 // scratch (v = x_hat - l/rho) for two-pass projectors
  const T *x, *m, *xold, *lb, *ub;
  int nblk;
  int dir[3];
  T ih[3];
  T rho, rho1, gamma;
  int prox;
  T plo, phi;
  const ProjScalars<T>* ps;
  int flags;
  int vsrc;                             // 1: read v from `v`; 2: read the already projected y from `v`
  int nblk_or1() const { return nblk > 0 ? nblk : 1; }
};

struct DecideArgs {        // what the scalar decision of a search needs besides the sums
  int prox, nospec;
  double pmin, pmax, capdiv, cap_max;
  long long true_len;
};
// the small steps of a slab-decomposed search for all sets of an iteration at once (kernels_proj.hip: k_spec_sums_pack, k_spec_finish)
constexpr int SPEC_MAX_SETS = 8;
template <typename T>
struct SpecPackSet {
  const ProjScalars<T>* ps;
  const double* partials;
  const T* maxpart;
  const T* compact;
  T* seg;                 // this rank's fast segment of the set
  int is_l1;
};
template <typename T>
struct SpecPackArgs {
  int nsets;
  int local = 0;          // one rank, no exchange: the header only (the gathered magnitudes stay where the pass left them)
  long long cap;
  SpecPackSet<T> s[SPEC_MAX_SETS];
};
template <typename T>
struct SpecFinishSet {
  ProjScalars<T>* ps;
  DecideArgs da;
  double* reg;            // the set's region of the staging buffer: receives the summed header
  const T* fseg0;         // the set's fast segment in rank 0's chunk
  T* compact;
  const double* partials;
  T radius;
  int* host_want;
  unsigned* verdict;
};
template <typename T>
struct SpecFinishArgs {
  int local = 0;          // one rank, no exchange: nothing to string together
  int nsets, world;
  long long fchunk;
  unsigned seq;
  double hw_max;
  int lean_on;
  long long coop_min;
  SpecFinishSet<T> s[SPEC_MAX_SETS];
};

template <typename T>
struct SampleSet {
  SetArgs<T> a;
  ProjScalars<T>* ps;
  double* partials;
  double* reg;            // the set's region of the sample staging buffer
  long long true_len;
};
template <typename T>
struct SampleMulti {
  int v_is_s = 0;         // the vector is s = A x itself (searches of the feasibility estimates: no y, no l)
  int ns;
  SampleSet<T> s[SPEC_MAX_SETS];
};
template <typename T>
struct RescaleMulti {
  int n;
  ProjScalars<T>* ps[SPEC_MAX_SETS];
  double factor[SPEC_MAX_SETS];
};

// the lean first passes of up to three l1 searches in one sweep (kernels_proj.hip, k_lean_multi): per set its arguments and buffers
template <typename T>
struct LeanSet {
  SetArgs<T> a;
  ProjScalars<T>* ps;
  T* compact;
  double* partials;
  T* maxpart;
};
constexpr int LEAN_MAX = 3;
template <typename T>
struct LeanMulti {
  int v_is_s = 0;         // the vector is s = A x itself (searches of the feasibility estimates)
  int ns;
  LeanSet<T> s[LEAN_MAX];
};

// ---- k_yl_multi: the y/l update of ALL sets in one z-marching sweep (kernels_multi.hip) --------------------------------------
constexpr int MULTI_MAXB = 8;     // operator blocks (a set has one per difference direction; identity: one)
template <typename T>
struct MultiBlk {
  const T *y, *l;                 // block q of the set's current iterate (base already offset by q * N)
  T *yo, *lo;                     // where the update goes: never the arrays y, l themselves (neighbours re-read the old iterate)
  T *lh0, *s0;                    // snapshots l_hat_0, s_0 (read and rewritten on Barzilai-Borwein iterations, written on the first)
  const T *y0, *l0;               // the snapshot pair y_0, l_0
  const ProjScalars<T>* ps;
  int dir;                        // -1: identity; 0 / 1 / 2: forward difference along that grid dimension
  int set;                        // index of the set the block belongs to (its group of partial slots)
  int first, last;                // first / last block of its set
  int dist;                       // the distance term
  int feas_el;                    // element-wise set (bounds, prox_l1): its feasibility estimate is taken in the sweep (F_FEAS)
  int in_rhs = 1;                 // the set's A'(rho y + l) goes into the fused right-hand side (0: a set behind one the sweep does not take -- the
                                  // sets are added in order, rhs_compose.jl:24-31, so the caller adds the rest with k_rhs)
  T ih, rho, rho1, gamma;
  int prox;
  T plo, phi;
};
template <typename T>
struct MultiArgs {
  int nblk;
  MultiBlk<T> b[MULTI_MAXB];
  const T *x, *m, *xold;
  const T* x0;                    // snapshot of x from which s_0 = A x_0 is recomputed (nullptr: s_0 is kept per set, MultiBlk::s0)
  T* x0w;                         // where the new snapshot goes on BB / first iterations (the two arrays take turns)
  T* rhs;                         // nullptr: no fused right-hand side
  double* partials;               // the engine's per-set partial array (SET_SLOTS groups)
  int flags;                      // F_FEAS | F_BB | F_FIRST (0: the lean variant)
  long long zlo, zhi;             // planes [zlo, zhi) of the last grid dimension (the rank's slab; the whole grid on one rank)
  long long zsum;                 // first plane whose sums / rhs belong to this rank: the planes in front of it are the
                                  // neighbour's last ones, recomputed (and stored) instead of being received
};

// One changed set of a fused Q update: Q[:,col(off_j)] += alpha * AtA_i[:,j] (CDS_scaled_add!.jl:16-22).
// ata == nullptr: the band values of A_i'A_i are regenerated from the operator descriptor.
// Minkowski mode (PARSDMM_precompute_distribute_Minkowski.jl): unknowns x = [u; v], every operator is one block row
// [A 0] (component 1), [0 A] (component 2) or [A A] (component 3, also the distance term [I I]).
template <typename T>
struct MkSet {
  T alpha;
  int nblk, comp;
  int dir[3];
  T ih[3];
};
template <typename T>
struct MkArgs {
  int nsets;
  MkSet<T> s[MAX_SETS];
};

template <typename T>
struct QSet {
  T alpha;
  const T* ata;
  int nblk;
  int dir[3];
  T ih[3];
  int nband;
  long long off[9];
};
template <typename T>
struct QArgs {
  int nsets;
  QSet<T> s[MAX_SETS];
};

template <typename T>
struct RhsSet {
  const T *y, *l;
  T rho;
  int nblk;
  int dir[3];
  T ih[3];
};
template <typename T>
struct RhsArgs {
  int nsets;
  RhsSet<T> s[MAX_SETS];
};

// State of one CG solve, device resident (mirrored to pinned host memory by the scalar kernels).
template <typename T>
struct CgState {
  double ss;        // float64 sum ||r||^2
  T rr;             // TF(ss)  == dot(r,z) of the reference (z aliases r)
  T gamma, alpha, beta;
  T nr0, tol, res_last;
  int done, flag, iters;
  T tol_ref;        // x_solve_tol_ref chain (argmin_x.jl:33-37)
  int it_outer;
  unsigned seq;     // number of this solve (ticket word, see publish_ticket)
};

// ---- launchers (explicitly instantiated for float and double in the .hip files) ----
template <typename T>
struct K {
  // CDS
  static void spmv(hipStream_t s, const Grid& g, long long N, const T* R, const CdsArgs& a, const T* x, T* y);
  // rows [r0, r1) of the product only (the z-slab of a sharded x-step; 0, N for the whole matrix); xold may be NULL
  static void spmv_dot(hipStream_t s, long long N, long long r0, long long r1, const T* R, const CdsArgs& a, const T* p, T* Ap,
                       double* partials, const CgState<T>* st);
  static void resid(hipStream_t s, long long N, long long r0, long long r1, const T* R, const CdsArgs& a, const T* x, const T* b,
                    T* r, T* p, T* xold, double* partials);
  // the same with a pending Q update applied on the fly (kernels_cds.hip); false when the z-marching form does not apply
  static bool resid_qupdate(hipStream_t s, const Grid& g, long long N, const T* R_old, T* R_new, const CdsArgs& a, const QArgs<T>& qa,
                            const T* x, const T* b, T* r, T* p, T* xold, double* partials);
  // the scalar step of CG iteration k (stop test, beta) + the product of iteration k+1 on p = r + beta p_old formed on the fly
  // (r, p_old carry a zero halo); writes p_new, Ap and the partials of p_new . Ap
  static void spmv_fused(hipStream_t s, long long N, const T* R, const CdsArgs& a, const T* r, const T* p_old, T* p_new, T* Ap,
                         double* partials, CgState<T>* st, CgState<T>* host, unsigned long long* ticket);
  static void sq_spmv(hipStream_t s, const Grid& g, const StencilQ<T>& q, const T* x, T* y);
  static void sq_spmv_dot(hipStream_t s, const Grid& g, const StencilQ<T>& q, const T* p, T* Ap, double* partials,
                          const CgState<T>* st);
  static void sq_resid(hipStream_t s, const Grid& g, const StencilQ<T>& q, const T* x, const T* b, T* r, T* p, T* xold,
                       double* partials);
  static void q_axpy(hipStream_t s, long long N, T* Qband, const T* Aband, T alpha);
  static void q_update(hipStream_t s, const Grid& g, long long r0, long long r1, const CdsArgs& q, const QArgs<T>& a, T* Q);
  static void gen_ata(hipStream_t s, const Grid& g, int nblk, const int* dir, const T* ih, int nband,
                      const long long* offs, T* R);
  // CG
  static void cg_begin(hipStream_t s, double* partials, CgState<T>* st, CgState<T>* host, int it_outer, T tol_ref,
                       unsigned seq, unsigned long long* ticket);
  static void cg_update_xr(hipStream_t s, long long N, const T* x_in, T* x, const T* r_in, T* r, const T* p, const T* Ap, double* partials,
                           CgState<T>* st, CgState<T>* host, int iter, unsigned long long* ticket, long long hlo = 0,
                           long long hhi = 0);      // hlo / hhi: elements in front of / behind the range whose x (p) follows along
  static void cg_update_p(hipStream_t s, long long N, T* p, const T* r, const double* partials, CgState<T>* st,
                          CgState<T>* host, unsigned long long* ticket, long long hlo = 0, long long hhi = 0);
  // sets
  static void rhs_compose(hipStream_t s, const Grid& g, const RhsArgs<T>& a, T* rhs, int accumulate);
  static void yl(hipStream_t s, const Grid& g, const SetArgs<T>& a, double* partials);
  static void adj_norm(hipStream_t s, const Grid& g, const SetArgs<T>& a, double* partials);
  // every set's y/l update (+ r_pri, r_dual sums, obj / evol sums, optionally rhs of the next iteration) in one sweep;
  // returns false (nothing launched) when the block list does not fit the kernel's instantiations
  // (probe_only: only say whether the list fits)
  static bool yl_multi(hipStream_t s, const Grid& g, const MultiArgs<T>& a, bool probe_only = false);
  static void fwd(hipStream_t s, const Grid& g, int nblk, const int* dir, const T* ih, const T* x, T* out);
  static void adj(hipStream_t s, const Grid& g, int nblk, const int* dir, const T* ih, const T* v, T* out);
  static void log3(hipStream_t s, long long N, const T* x, const T* m, const T* xold, double* partials);
  static void q_update_mk(hipStream_t s, const Grid& g, const CdsArgs& q, const MkArgs<T>& a, T* Q);
  static void mirror_bands(hipStream_t s, long long N, const CdsArgs& q, T* Q);
  static void sum_uv(hipStream_t s, long long N, const T* u, const T* v, T* w);
  // caller-supplied sparse operator: s = A x (CSR view), out (+)= A'(rho y + l), partial sum of (A' dy)^2
  static void csr_spmv(hipStream_t s, long long M, const long long* rowptr, const long long* col, const T* val, const T* x, T* out);
  static void csc_adj_rhs(hipStream_t s, long long N, const long long* colptr, const long long* row, const T* val, const T* y,
                          const T* l, T rho, T* out, int accumulate);
  static void csc_adj_norm(hipStream_t s, long long N, const long long* colptr, const long long* row, const T* val, const T* dy,
                           double* partials);
  static void rows_pack(hipStream_t s, const Grid& g, int dir, long long nrows, const T* pad, T* rows);
  static void rows_unpack(hipStream_t s, const Grid& g, int dir, long long nrows, const T* rows, T* pad);
  static void fin_sum(hipStream_t s, const double* partials, int nslots, double* out_dev, double* out_host, unsigned* ticket = nullptr,
                      unsigned long long* word = nullptr, unsigned long long seq = 0);
  static void copy_f64(hipStream_t s, const double* src, double* dst, int n);      // device -> pinned host, by a kernel
  // Scalars of the two-pass projectors (l1 threshold, l2 / annulus scale) of a vector that is either
  // produced on the fly by a set (v = x_hat - l/rho, or s = A x when v_is_s) or stored in an array.
  // Enqueues: first pass (sums + probe + speculative compaction), bracket, gated refinement / compaction, solve.
  static void ps_rescale(hipStream_t s, ProjScalars<T>* ps, double factor);
  static void proj_scalars_set(hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, ProjScalars<T>* ps,
                               double* partials, T* maxpart, T* compact, long long true_len, SampleCtl ctl = SampleCtl(),
                               const ChainHooks* hooks = nullptr);
  // one stage (0..3) of the same search, for a caller that runs the searches of several sets in lock step with ONE collective
  // between the stages (slab-decomposed iteration): reg = the set's region of the all-reduced staging buffer, gseg0 / chunk =
  // its segment in rank 0's chunk of the exchange buffer and the distance to the next rank's
  static void spec_sums_pack(hipStream_t s, const SpecPackArgs<T>& A);
  static void spec_finish(hipStream_t s, SpecFinishArgs<T>& A);
  static void ps_rescale_multi(hipStream_t s, const RescaleMulti<T>& A);
  static void sample_multi(int stage, hipStream_t s, const Grid& g, const SampleMulti<T>& A, long long runs, const ChainHooks* hk);
  static void lean_multi(hipStream_t s, const Grid& g, const LeanMulti<T>& m);
  // mode 0 / 1 / 2: the full first passes / gated refinement passes / gated compaction passes of up to LEAN_MAX searches in one sweep
  static void pass_multi(int mode, hipStream_t s, const Grid& g, const LeanMulti<T>& m, int v_is_s);
  // stages 1 (sums of a refinement pass + decision) and 3 (solve) of a search without their passes (those came from pass_multi)
  static void search_tail(int stage, hipStream_t s, const SetArgs<T>& a, ProjScalars<T>* ps, double* partials, T* maxpart, T* compact,
                          long long true_len, SampleCtl ctl, double* reg);
  static void proj_scalars_stage(int stage, hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, ProjScalars<T>* ps,
                                 double* partials, T* maxpart, T* compact, long long true_len, SampleCtl ctl, const ChainHooks* hooks,
                                 double* reg, T* gseg0, long long chunk);
  static void proj_scalars_arr(hipStream_t s, long long len, const T* v, int prox, T pmin, T pmax, ProjScalars<T>* ps,
                               double* partials, T* maxpart, T* compact, long long true_len);
  static void proj_scalars_arr_slab(hipStream_t s, long long len, const T* v, int prox, T pmin, T pmax, ProjScalars<T>* ps,
                                    double* partials, T* maxpart, T* compact, long long true_len, const ChainHooks* hooks,
                                    long long compact_len, int* host_ovf);
  // ||P(v)-v||^2 and ||v||^2 of the set-produced vector into partial slots dst[0..NB), dst[NB..2NB)
  static void proj_dist_set(hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, const ProjScalars<T>* ps,
                            double* dst);
  static void ps_init(hipStream_t s, ProjScalars<T>* ps, long long* cidx);
  // out[e] = v (or s = A x when v_is_s) of the set, padded layout: input of the library-backed projectors
  static void store_v(hipStream_t s, const Grid& g, const SetArgs<T>& a, int v_is_s, T* out);
};

// ||P(v)-v||^2, ||v||^2 (slots 0,1) and v = P(v) over a padded vector (pads skipped)
template <typename T>
void proj_dist_grid(hipStream_t s, const Grid& g, int nblk, const int* dir, long long len, const T* v, int prox, T plo,
                    T phi, const T* lb, const T* ub, const ProjScalars<T>* ps, double* partials);
template <typename T>
void proj_apply_grid(hipStream_t s, const Grid& g, int nblk, const int* dir, long long len, T* v, int prox, T plo,
                     T phi, const T* lb, const T* ub, const ProjScalars<T>* ps);

// x = prox_l2s(x, rho, m) element-wise (the distance term's prox, prox_l2s!.jl:3-6)
template <typename T>
void prox_l2s_dev(hipStream_t s, long long n, T* x, const T* m, T rho);

// nearest-neighbour grid transfer (multilevel): out (shape nf) <- in (shape nc)
template <typename T>
void resample_nn(hipStream_t s, const long long* nc, const long long* nf, const T* in, T* out);
// ... a set on TV / D2D / D3D: the reference's chunks of the row vector (kernels_sets.hip, k_resample_rows)
template <typename T>
void resample_nn_rows(hipStream_t s, const long long* nc, const long long* nf, int nblk, const int* dir, int b, long long cstride,
                      long long e0, long long e1, const T* in, T* out);
// ... between padded arrays, for the fine grid points [e0, e1) (kernels_sets.hip)
template <typename T>
void resample_nn_padded(hipStream_t s, const long long* nc, const long long* nf, const long long* cc, const long long* cf, long long e0,
                        long long e1, const T* in, T* out);

#define SIPX_HIP(expr)                                                                       \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

}  // namespace sipx
