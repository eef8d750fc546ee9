// Host engine behind the sipx C ABI: owns the device-resident PARSDMM state and drives the HIP
// kernels phase by phase, in the order of the reference's main loop (src/PARSDMM.jl:97-254).
// No CPU fallback exists: every numerical step below is a kernel launch.
#include "engine.h"
#include "comm.h"
#include "dist_dft.h"
#include "ext_proj.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <thread>
#include <exception>

namespace sipx {

const LaunchObserver*& launch_observer() {
  static thread_local const LaunchObserver* obs = nullptr;
  return obs;
}

namespace {
EnvKnobs g_env_knobs;
}
const EnvKnobs& env_knobs() { return g_env_knobs; }
void refresh_env_knobs() {
  auto num = [](const char* name, long long dflt) { const char* e = std::getenv(name); return e ? std::atoll(e) : dflt; };
  EnvKnobs k;
  k.cds_march = (int)num("SIPX_CDS_MARCH", 1);
  k.cds_march_zchunk = num("SIPX_CDS_MARCH_ZCHUNK", 0);
  k.multi_zchunk = num("SIPX_MULTI_ZCHUNK", 0);
  k.rhs_march = (int)num("SIPX_RHS_MARCH", 1);
  k.rhs_march_zchunk = num("SIPX_RHS_MARCH_ZCHUNK", 0);
  k.q_plan = (int)num("SIPX_Q_PLAN", 1);
  k.mark_stride = (int)std::max<long long>(0, num("SIPX_MARK_STRIDE", 0));
  k.trace_searches = (int)num("SIPX_TRACE_SEARCHES", 0);
  k.trace_kernels = (int)num("SIPX_TRACE_KERNELS", 0);
  g_env_knobs = k;
}
long long*& alloc_tally() {
  static thread_local long long* t = nullptr;
  return t;
}

namespace {

struct TallyGuard {
  long long* prev;
  explicit TallyGuard(long long* t) : prev(alloc_tally()) { alloc_tally() = t; }
  ~TallyGuard() { alloc_tally() = prev; }
};

// installs a context's observer for the duration of one entry point (the launchers consult it through launch_observer())
struct ObserverGuard {
  const LaunchObserver* prev;
  explicit ObserverGuard(const LaunchObserver* o) : prev(launch_observer()) { launch_observer() = o; }
  ~ObserverGuard() { launch_observer() = prev; }
};

constexpr int SLOTS = SET_SLOTS;      // reduction slots per set: 0..12 k_yl, 13 ||A'dy||^2, 14/15 two-pass feasibility

// what the tally counted for an allocation, so that freeing it while a tally is active takes the bytes back (the temporaries of
// sipx_finalize -- upload_rows' staging, the whole-size buffer of upload_rows_ranged -- used to stay in device_bytes_per_rank)
inline std::map<void*, long long>& tally_sizes() {
  static std::map<void*, long long> m;
  return m;
}
inline std::mutex& tally_mutex() {
  static std::mutex m;
  return m;
}
inline void tally_add(void* p, long long bytes) {
  if (long long* t = alloc_tally()) {
    *t += bytes;
    std::lock_guard<std::mutex> lk(tally_mutex());
    tally_sizes()[p] += bytes;
  }
}
inline void tally_release(void* p) {
  long long* t = alloc_tally();
  std::lock_guard<std::mutex> lk(tally_mutex());
  auto it = tally_sizes().find(p);
  if (it == tally_sizes().end()) return;
  if (t) *t -= it->second;
  tally_sizes().erase(it);
}
// bytes of every plain allocation (sipx_reset zeroes a context's arrays without knowing each one's length)
inline std::map<void*, size_t>& alloc_sizes() {
  static std::map<void*, size_t> m;
  return m;
}
template <typename T>
T* dalloc(size_t n, bool zero = true) {
  T* p = nullptr;
  if (n == 0) return p;
  SIPX_HIP(hipMalloc(&p, n * sizeof(T)));
  tally_add(p, (long long)(n * sizeof(T)));
  {
    std::lock_guard<std::mutex> lk(tally_mutex());
    alloc_sizes()[p] = n * sizeof(T);
  }
  if (zero) {
    // hipMemset is queued on the NULL stream; the engine stream is non-blocking, so wait here or the
    // zero-fill may land after kernels of the engine stream have already written the buffer.
    SIPX_HIP(hipMemset(p, 0, n * sizeof(T)));
    SIPX_HIP(hipStreamSynchronize(nullptr));
  }
  return p;
}
// SPARSE arrays (round 4, slab-decomposed contexts): the array keeps its GLOBAL index space -- the whole range is reserved in the
// virtual address space, so every kernel indexes it exactly as before -- but only the element ranges a rank touches (its planes,
// the halo planes around them) are backed by memory (hipMemAddressReserve / hipMemCreate / hipMemMap, 2 MiB granules).  A rank of
// eight then holds an eighth of every N-vector (plus three planes) instead of all of it: the decomposition grows the problem that
// fits, not only its speed.  An access outside the mapped ranges faults instead of reading stale data.
struct SparseBlock {
  size_t total = 0;
  std::vector<std::pair<size_t, size_t>> maps;                   // (offset, length) in bytes
  std::vector<hipMemGenericAllocationHandle_t> handles;
};
inline std::map<void*, SparseBlock>& sparse_registry() {
  static std::map<void*, SparseBlock> reg;
  return reg;
}
inline std::mutex& sparse_mutex() {
  static std::mutex m;
  return m;
}
constexpr size_t SPARSE_GRAN = 2ull << 20;
// ranges: [first, last) in BYTES of the array's address space; returns the base of the reservation (zero-filled where mapped)
inline void* sparse_alloc_bytes(size_t total_bytes, std::vector<std::pair<size_t, size_t>> ranges, int device) {
  const size_t total = (total_bytes + SPARSE_GRAN - 1) / SPARSE_GRAN * SPARSE_GRAN;
  for (auto& r : ranges) {
    r.first = r.first / SPARSE_GRAN * SPARSE_GRAN;
    r.second = std::min(total, (r.second + SPARSE_GRAN - 1) / SPARSE_GRAN * SPARSE_GRAN);
  }
  std::sort(ranges.begin(), ranges.end());
  std::vector<std::pair<size_t, size_t>> merged;
  for (const auto& r : ranges) {
    if (r.second <= r.first) continue;
    if (!merged.empty() && r.first <= merged.back().second) merged.back().second = std::max(merged.back().second, r.second);
    else merged.push_back(r);
  }
  void* base = nullptr;
  SIPX_HIP(hipMemAddressReserve(&base, total, SPARSE_GRAN, nullptr, 0));
  SparseBlock blk;
  blk.total = total;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  // Every mapping of a reservation has the SAME size, one granule: hipMemSetAccess of this runtime (ROCm 7.2) answers "invalid
  // argument" for a mapping whose size differs from the others inside one reservation (4 + 4 + 2 MiB fails at the third,
  // 2 + 4 at the second; uniform sizes are fine -- probed with scratch/vmm_test3).  2 MiB is the native large page.
  for (const auto& r : merged) {
    for (size_t off = r.first; off < r.second; off += SPARSE_GRAN) {
      const size_t len = SPARSE_GRAN;
      auto chk = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return;
        char msg[256];
        std::snprintf(msg, sizeof msg, "sparse array: %s failed (%s): reservation %zu bytes at %p, granule at %zu of range [%zu, %zu)", what,
                      hipGetErrorString(e), total, base, off, r.first, r.second);
        throw std::runtime_error(msg);
      };
      hipMemGenericAllocationHandle_t h;
      chk(hipMemCreate(&h, len, &prop, 0), "hipMemCreate");
      chk(hipMemMap((char*)base + off, len, 0, h, 0), "hipMemMap");
      chk(hipMemSetAccess((char*)base + off, len, &acc, 1), "hipMemSetAccess");
      blk.maps.push_back({off, len});
      blk.handles.push_back(h);
      tally_add(base, (long long)len);
    }
    SIPX_HIP(hipMemset((char*)base + r.first, 0, r.second - r.first));
  }
  SIPX_HIP(hipStreamSynchronize(nullptr));
  std::lock_guard<std::mutex> lk(sparse_mutex());
  sparse_registry()[base] = std::move(blk);
  return base;
}
// A device-to-host copy into memory the caller has just allocated spends most of its time in page faults (one per 4 KiB, taken
// one after the other by the copy's staging thread: 1 GiB arrives in 68 ms, in 20 ms once the pages exist -- round 5).  The
// destination of a large download is therefore touched first, by several threads at once: one write per page makes the kernel
// map it, and the copy that follows overwrites every byte.  (A destination that already has its pages loses a few hundred
// microseconds per GiB to this.)
inline void host_prefault(void* p, size_t bytes) {
  constexpr size_t PAGE = 4096, MIN_BYTES = 8u << 20;
  static const int nthreads = [] {
    const char* e = std::getenv("SIPX_PREFAULT_THREADS");        // 0: off (A/B switch)
    if (e) return std::max(0, std::atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(16u, hc > 1 ? hc / 2 : 1u);      // (9 GiB: 0.64 s without, 0.34 / 0.27 s with 4 / 16 threads)
  }();
  if (!p || bytes < MIN_BYTES || nthreads < 1) return;
  char* base = static_cast<char*>(p);
  const size_t first = (PAGE - (reinterpret_cast<uintptr_t>(base) & (PAGE - 1))) & (PAGE - 1);      // first page boundary inside
  if (first >= bytes) return;
  const size_t npages = (bytes - first + PAGE - 1) / PAGE;
  auto touch = [base, first, npages, bytes](size_t a, size_t b) {
    for (size_t k = a; k < b && k < npages; ++k) {
      volatile char* q = base + first + k * PAGE;
      if ((size_t)(q - base) < bytes) *q = 0;
    }
  };
  std::vector<std::thread> th;
  const size_t per = (npages + (size_t)nthreads - 1) / (size_t)nthreads;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(touch, (size_t)t * per, (size_t)(t + 1) * per);
  base[0] = 0;
  touch(0, per);
  for (auto& t : th) t.join();
}
inline void dfree(void* p) {
  if (!p) return;
  tally_release(p);
  {
    std::lock_guard<std::mutex> lk(sparse_mutex());
    auto it = sparse_registry().find(p);
    if (it != sparse_registry().end()) {
      for (size_t k = 0; k < it->second.maps.size(); ++k) {
        (void)hipMemUnmap((char*)p + it->second.maps[k].first, it->second.maps[k].second);
        (void)hipMemRelease(it->second.handles[k]);
      }
      (void)hipMemAddressFree(p, it->second.total);
      sparse_registry().erase(it);
      return;
    }
  }
  {
    std::lock_guard<std::mutex> lk(tally_mutex());
    alloc_sizes().erase(p);
  }
  (void)hipFree(p);
}
// zero-fill of an allocation made by dalloc or sparse_alloc_bytes (its mapped granules), queued on `s`
inline void dzero(void* p, hipStream_t s) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(sparse_mutex());
    auto it = sparse_registry().find(p);
    if (it != sparse_registry().end()) {
      for (const auto& mp : it->second.maps) SIPX_HIP(hipMemsetAsync((char*)p + mp.first, 0, mp.second, s));
      return;
    }
  }
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(tally_mutex());
    auto it = alloc_sizes().find(p);
    if (it == alloc_sizes().end()) throw std::runtime_error("internal: dzero of an allocation the engine did not make");
    bytes = it->second;
  }
  SIPX_HIP(hipMemsetAsync(p, 0, bytes, s));
}


// Julia maximum(): NaN-propagating
template <typename It>
double julia_maximum(It b, It e) {
  double m = -INFINITY;
  for (; b != e; ++b) {
    if (std::isnan(*b)) return NAN;
    m = std::max(m, (double)*b);
  }
  return m;
}

// Barzilai-Borwein scalar rule, reference src/adapt_rho_gamma.jl:55-126, all arithmetic in T.
template <typename T>
void bb_rule(T d_dHh_dlh, T n_d_H_hat, T n_d_l_hat, T n_d_l, T n_d_G_hat, T d_dGh_dl, bool adjust_rho,
             bool adjust_gamma, T& rho, T& gamma) {
  const T safeguard = sizeof(T) == 8 ? T(1e-10) : T(1e-6);   // :31-35
  const T eps_correlation = T(0.3);                          // :37
  bool alpha_reliable = false, beta_reliable = false;
  T alpha_correlation = 0, beta_correlation = 0;
  if ((n_d_H_hat * n_d_l_hat) > safeguard && (n_d_H_hat * n_d_H_hat) > safeguard && d_dHh_dlh > safeguard) {
    alpha_reliable = true;
    alpha_correlation = d_dHh_dlh / (n_d_H_hat * n_d_l_hat);
  }
  if ((n_d_G_hat * n_d_l) > safeguard && (n_d_G_hat * n_d_G_hat) > safeguard && d_dGh_dl > safeguard) {
    beta_reliable = true;
    beta_correlation = d_dGh_dl / (n_d_G_hat * n_d_l);
  }
  bool alpha_comp = false, beta_comp = false;
  T alpha_hat = 0, beta_hat = 0;
  if (alpha_reliable && alpha_correlation > eps_correlation) {
    alpha_comp = true;
    const T mg = d_dHh_dlh / (n_d_H_hat * n_d_H_hat);
    const T sd = (n_d_l_hat * n_d_l_hat) / d_dHh_dlh;
    alpha_hat = (T(2) * mg) > sd ? mg : sd - mg / T(2);
  }
  if (beta_reliable && beta_correlation > eps_correlation) {
    beta_comp = true;
    const T mg = d_dGh_dl / (n_d_G_hat * n_d_G_hat);
    const T sd = (n_d_l * n_d_l) / d_dGh_dl;
    beta_hat = (T(2) * mg) > sd ? mg : sd - mg / T(2);
  }
  if (adjust_rho) {
    if (alpha_comp && beta_comp) rho = std::sqrt(alpha_hat * beta_hat);
    else if (alpha_comp) rho = alpha_hat;
    else if (beta_comp) rho = beta_hat;
  }
  if (adjust_gamma) {
    if (alpha_comp && beta_comp) gamma = T(1) + ((T(2) * std::sqrt(alpha_hat * beta_hat)) / (alpha_hat + beta_hat));
    else if (alpha_comp) gamma = T(1.9);
    else if (beta_comp) gamma = T(1.1);
    else gamma = T(1.5);
  }
}

template <typename T>
struct SetState {
  int op = 0, prox = 0, nblk = 0, ncvx = 0;
  int nblk_or1() const { return nblk > 0 ? nblk : 1; }
  // caller-supplied sparse operator (SIPX_OP_CSC): CSC for the adjoint, a CSR copy for the forward product; s = A x is
  // materialised in sbuf and every set kernel then runs in its identity shape on a 1-D grid of M entries
  bool custom = false;
  std::vector<long long> h_colptr, h_rowval;
  std::vector<T> h_nzval;
  long long *d_colptr = nullptr, *d_rowval = nullptr, *d_rowptr = nullptr, *d_colidx = nullptr;
  T *d_nzval = nullptr, *d_rval = nullptr, *sbuf = nullptr;
  Grid gm;                           // the 1-D "grid" of the M rows
  int comp = 0;                      // Minkowski component: 0 none, 1 = [A 0], 2 = [0 A], 3 = [A A]
  int dir[3] = {0, 0, 0};
  T ih[3] = {0, 0, 0};
  long long Mtrue = 0, Mpad = 0;
  long long blk_rows[3] = {0, 0, 0};
  T plo = 0, phi = 0;
  bool ident = true, two_pass = false, is_dist = false, owned = true;
  bool in_sweep = false;             // the one-sweep update (k_yl_multi) takes this set; the others keep their per-set kernels
  // y, l: the arrays holding the current iterate; y0, l0: the other pair.  The snapshot (y_0, l_0) of the BB rule lives
  // in whichever pair `snap` names: on a snapshot iteration the update is written over the old snapshot (after it has
  // been read), on the others into the pair that is not the snapshot -- the reference's copies y_0 <- y, l_0 <- l never
  // happen (PARSDMM.jl:174-177,200-203).
  T *y = nullptr, *l = nullptr, *dy = nullptr, *lh0 = nullptr, *y0 = nullptr, *s0 = nullptr, *l0 = nullptr;
  T *y2 = nullptr, *l2 = nullptr;    // third pair of the one-sweep update (allocated on first need, see update_all_sets_in_one_sweep)
  int snap = -1;                     // -1: no snapshot yet; 0: (y, l) is also the snapshot; 1: (y0, l0) is
  T *lb = nullptr, *ub = nullptr, *ata = nullptr;
  std::vector<void*> halo_allocs;   // bases of the vectors allocated with a front halo
  int searches_done = 0;             // slab-decomposed l1 searches of this set so far (sizes the refinement rounds)
  int ext_kind = 0;                  // projector acting on a materialised vector (ext_proj.h)
  // Sharded solve: a rank / nuclear-norm set on the slices orthogonal to the last grid dimension is projected by ALL ranks,
  // each factorising the slices of its z-slab (`ext` is then built for the slab on every rank, owner or not)
  bool dist_ext = false;
  int owner_rank = 0;
  // Slab-decomposed solve (round 5, the long lists: BASELINE config 4): y, l of EVERY set live on the ranks' z-slabs; a set whose
  // projector cannot work from sums over the grid is projected on a materialised v --
  //   slab_ext: slice-wise rank / nuclear norm on the z-slices of the identity: every rank projects the slices of its own slab,
  //             nothing crosses the fabric;
  //   fan:      everything else (a projector behind a transform, cardinality): rank fan_owner gathers v, projects the whole
  //             array, scatters P(v) back (two fan exchanges of N w bytes for that set; the others stay slab-local).
  //   slab_card: cardinality of the whole array: a search of its own through the slab collectives (all-reduced probe counts, the
  //             pairs inside the final bracket all-gathered: launch_chain, kernels_proj.hip), then the per-set update.
  //   slab_dft:  the l1 ball behind the DFT on a 3-D grid: the transform itself is slab-decomposed (dist_dft.h: 2-D transforms of
  //             the rank's planes, one all-to-all, 1-D transforms along z; the threshold through the slab collectives).
  bool slab_ext = false, fan = false, slab_card = false, slab_dft = false;
  std::shared_ptr<DistDft<T>> ddft;
  int fan_owner = 0;
  bool fan_on_side = false;          // this update's projection was queued on the fan stream
  T* fanv = nullptr;                 // a gathered set's own whole-size vector (v, then P(v)): its owner projects it on the fan stream
  hipEvent_t fan_ev = nullptr;       // ... and records this when P(v) is ready
  ExtSpec spec;
  std::vector<T> host_basis;
  std::shared_ptr<ExtProj<T>> ext;
  ProjScalars<T>* ps = nullptr;    // scalars of prox_i (warm-started across iterations)
  ProjScalars<T>* psf = nullptr;   // scalars of the feasibility estimate P_i(A_i x)
  std::vector<long long> ata_off;
  std::vector<T> host_lb, host_ub, host_ata;
  double sums[SLOTS] = {0};
  bool bb_valid = false;
  T last_rho = T(-1), last_gamma = T(-1);   // parameters of the previous y/l update (speculation of the l1 search)
  // y/l updates of different sets are independent (the reference runs them on separate workers): the sets are dealt onto
  // a small pool of HIP streams, each with private search scratch, so the one-workgroup decide / solve kernels of one l1
  // search overlap with the streaming passes of another set instead of leaving the GPU idle
  hipStream_t st = nullptr;
  hipEvent_t ev = nullptr;
  long long cbuf_len = 0;            // elements cbuf holds
  double* ptmp = nullptr;
  T *mpart = nullptr, *cbuf = nullptr;
};

}  // namespace

template <typename T>
class Engine : public EngineBase {
 public:
  Engine(int ndim, const int64_t* n, const double* h, int device) : device_(device) {
    if (ndim != 2 && ndim != 3) throw std::runtime_error("ndim must be 2 or 3");
    SIPX_HIP(hipSetDevice(device));
    SIPX_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    SIPX_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    SIPX_HIP(hipEventCreateWithFlags(&ev_fork2_, hipEventDisableTiming));
    {
      const char* e = std::getenv("SIPX_SERIAL_SETS");
      set_streams_ = !(e && e[0] == '1');
      const char* f = std::getenv("SIPX_CDS_FULL");
      cds_full_ = f && f[0] == '1';
      const char* k = std::getenv("SIPX_SET_STREAMS");
      if (k && std::atoi(k) > 0) { n_set_streams_ = std::atoi(k); set_streams_forced_ = true; }
    }
    ndim_ = ndim;
    for (int a = 0; a < 3; ++a) {
      G_.n[a] = a < ndim ? n[a] : 1;
      if (G_.n[a] < 1) throw std::runtime_error("grid size must be positive");
      ih_[a] = a < ndim ? T(1) / T(h[a]) : T(0);   // entries +-1/h of get_discrete_Grad.jl:22-23,58-60
    }
    if (ndim == 3 && G_.n[2] == 1) ndim_ = 2;      // get_TD_operator.jl:30
    G_.N = G_.n[0] * G_.n[1] * G_.n[2];
    // Up to 2^22 grid points the kernels of the searches are too short for a second stream to hide anything: the cross-stream
    // dependencies (an event wait costs 15-25 us on this stack) outweigh the overlap.  2048^2, C2: 2130 -> 2300 it/s on one
    // stream; 256^3 gains 6 % from its three.  SIPX_SERIAL_SETS=0 / SIPX_SET_STREAMS=k keep the streams whatever the size.
    if (G_.N <= (1ll << 22) && !set_streams_forced_ && !std::getenv("SIPX_SERIAL_SETS")) set_streams_ = false;
    G_.st[0] = 1;
    G_.st[1] = G_.n[0];
    G_.st[2] = G_.n[0] * G_.n[1];
    if (G_.N >= (1ll << 31)) throw std::runtime_error("grids of 2^31 points or more are not supported");
    G_.set_fast_div();
  }
  ~Engine() override {
    (void)hipSetDevice(device_);
    if (lane_thr_.joinable()) lane_thr_.join();
    if (lane_st_) { (void)hipStreamSynchronize(lane_st_); (void)hipStreamDestroy(lane_st_); }
    for (hipEvent_t e : {lane_fork_, lane_ev_}) if (e) (void)hipEventDestroy(e);
    dfree(lane_v_);
    if (fan_st_) { (void)hipStreamSynchronize(fan_st_); (void)hipStreamDestroy(fan_st_); }
    if (fan_fork_) (void)hipEventDestroy(fan_fork_);
    for (void* p : {(void*)fan_ptmp_, (void*)fan_mpart_, (void*)fan_c_}) dfree(p);
    if (loose_owned_) { dfree(loose_v_); dfree(loose_w_); }
    (void)hipStreamSynchronize(stream_);
    for (auto& s : sets_) free_set(s);
    for (void* p : {(void*)xr_base_[0], (void*)xr_base_[1], (void*)xr_base_[2], (void*)w_base_, (void*)rhs_, (void*)m_base_, (void*)r_base_, (void*)p_base_, (void*)p2_base_, (void*)Ap_, (void*)Q_, (void*)Q2_,
                    (void*)scr_v_, (void*)scr_c_, (void*)scr_i_, (void*)scr_w_, (void*)part_cg_, (void*)part_tmp_, (void*)part_sets_,
                    (void*)maxpart_, (void*)cg_dev_, (void*)gbuf_, (void*)stage_, (void*)sstage_, (void*)fbuf_, (void*)agree_buf_})
      dfree(p);
    comm_.reset();
    if (cstream_) (void)hipStreamDestroy(cstream_);
    for (auto e : ev_c_) if (e) (void)hipEventDestroy(e);
    if (ev_sums_) (void)hipEventDestroy(ev_sums_);
    for (hipEvent_t e : open_ev_) if (e) (void)hipEventDestroy(e);
    if (ev_cgb_) (void)hipEventDestroy(ev_cgb_);
    if (ticket_) (void)hipHostFree((void*)ticket_);
    if (cg_host_) (void)hipHostFree(cg_host_);
    if (hres_) (void)hipHostFree(hres_);
    if (sums_word_) (void)hipHostFree((void*)sums_word_);
    dfree(sums_ticket_);
    if (hlean_) (void)hipHostFree((void*)hlean_);
    if (hovf_) (void)hipHostFree((void*)hovf_);
    if (hverd_) (void)hipHostFree((void*)hverd_);
    for (auto e : ev_) (void)hipEventDestroy(e);
    for (auto e : stat_ev_) (void)hipEventDestroy(e);
    for (auto e : cg_ev_) if (e) (void)hipEventDestroy(e);
    for (hipStream_t q : pool_) if (q != stream_) (void)hipStreamDestroy(q);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_fork2_) (void)hipEventDestroy(ev_fork2_);
    (void)hipStreamDestroy(stream_);
  }

  // ------------------------------------------------------------------------------------------
  int add_set(const sipx_set_desc* d, const void* ata_R, const int64_t* ata_off, int d_i) override {
    if (finalized_) throw std::runtime_error("sipx_add_set after sipx_finalize");
    SetState<T> s;
    if (d->op == SIPX_OP_CSC) configure_custom(s, d);
    else configure_op(s, d->op);
    configure_proj(s, d);
    if (s.custom && !ata_R) throw std::runtime_error("a custom sparse operator needs its A'A in CDS (ata_R, ata_off)");
    if (s.custom && (s.comp || s.ext_kind))
      throw std::runtime_error("custom sparse operators: Minkowski components and library-backed projectors are not available");
    if (ata_R && s.comp) throw std::runtime_error("Minkowski sets use descriptor-generated AtA (pass ata_R = NULL)");
    if (ata_R) {
      if (d_i < 1 || d_i > 9) throw std::runtime_error("AtA band count out of range (1..9 bands per set)");
      s.ata_off.assign(ata_off, ata_off + d_i);
      s.host_ata.assign((const T*)ata_R, (const T*)ata_R + (size_t)G_.N * d_i);
    } else {
      s.ata_off = default_ata_offsets(s);
    }
    sets_.push_back(std::move(s));
    return (int)sets_.size() - 1;
  }

  int64_t set_rows(int set) override {
    if (set == (int)sets_.size() && !finalized_) return G_.N;   // the distance term to come
    if (set < 0 || set >= (int)sets_.size()) throw std::runtime_error("set index out of range");
    return sets_[set].Mtrue;
  }
  void num_terms(int* p, int* pp) override {
    if (p) *p = p_n_;
    if (pp) *pp = pp_n_;
  }
  void set_owned(const int32_t* owned) override {
    if (finalized_) throw std::runtime_error("sipx_set_owned must precede sipx_finalize");
    owned_.assign(owned, owned + sets_.size() + 1);
  }

  void set_decomp(int mode) override {
    if (finalized_) throw std::runtime_error("sipx_set_decomp must precede sipx_finalize");
    if (mode != SIPX_DECOMP_SETS && mode != SIPX_DECOMP_SLAB && mode != SIPX_DECOMP_SLAB_FULL) throw std::runtime_error("unknown decomposition");
    slab_req_ = mode != SIPX_DECOMP_SETS;
    slab_full_req_ = mode == SIPX_DECOMP_SLAB_FULL;
  }

  void set_comm(Comm* c) override {
    std::unique_ptr<Comm> hold(c);
    if (finalized_) throw std::runtime_error("the communicator must be attached before sipx_finalize");
    counting_ = c ? new CountingComm(hold.release()) : nullptr;      // (owns c)
    comm_.reset(counting_);
  }
  void comm_info(int* nranks, int* rank, char* version, int version_len, int* decomposition) override {
    if (version && version_len > 0) version[0] = 0;
    if (nranks) *nranks = 1;
    if (rank) *rank = 0;
    if (decomposition) *decomposition = (finalized_ ? slab_ : (slab_req_ && comm_)) ? SIPX_DECOMP_SLAB : SIPX_DECOMP_SETS;
    if (comm_) comm_->info(nranks, rank, version, version_len > 0 ? (size_t)version_len : 0);
    else if (version && version_len > 0) std::snprintf(version, (size_t)version_len, "none");
  }
  void bind_device() override { SIPX_HIP(hipSetDevice(device_)); }
  void device_bytes(int64_t* context_bytes, int64_t* device_used, int64_t* device_total) override {
    SIPX_HIP(hipSetDevice(device_));
    size_t fr = 0, tot = 0;
    SIPX_HIP(hipMemGetInfo(&fr, &tot));
    if (context_bytes) *context_bytes = dev_bytes_;
    if (device_used) *device_used = (int64_t)(tot - fr);
    if (device_total) *device_total = (int64_t)tot;
  }
  void slab(int64_t* row0, int64_t* row1, int64_t* chunk) override {
    need_final();
    if (row0) *row0 = r0_;
    if (row1) *row1 = r1_;
    if (chunk) *chunk = comm_ ? chunk_ : Nx_;
  }

  void set_q_mode(int mode) override {
    if (finalized_) throw std::runtime_error("sipx_set_q_mode must precede sipx_finalize");
    if (mode != SIPX_Q_CDS && mode != SIPX_Q_STENCIL) throw std::runtime_error("unknown Q mode");
    stencil_q_ = mode == SIPX_Q_STENCIL;
  }

  // ------------------------------------------------------------------------------------------
  void finalize(const void* m, const double* rho_ini, int n_rho, double gamma_ini, int feasibility_only,
                int zero_ini_guess, const void* x0, const void* const* l0, const void* const* y0,
                double* feasibility_initial) override {
    if (finalized_) throw std::runtime_error("sipx_finalize called twice");
    SIPX_HIP(hipSetDevice(device_));
    refresh_env_knobs();                  // the launchers' A/B switches: read once per context, not per launch
    TallyGuard tally(&dev_bytes_);
    pp_n_ = (int)sets_.size();
    feasibility_only_ = feasibility_only != 0;
    if (!feasibility_only_) {             // PARSDMM_precompute_distribute.jl:17-26: identity operator for 1/2||x-m||^2
      SetState<T> s;
      configure_op(s, SIPX_OP_IDENTITY);
      s.prox = PX_DIST;
      s.is_dist = true;
      s.ata_off = {0};
      sets_.push_back(std::move(s));
    }
    p_n_ = (int)sets_.size();
    {
      int with = 0;
      for (int i = 0; i < pp_n_; ++i) with += sets_[i].comp != 0;
      if (with != 0 && with != pp_n_) throw std::runtime_error("either every set names its Minkowski component or none does");
      mk_ = with > 0;
      if (mk_ && !feasibility_only_) sets_.back().comp = 3;        // distance term [I I]
      if (mk_ && stencil_q_) throw std::runtime_error("the stencil form of Q is not available for Minkowski sets");
      if (mk_ && !owned_.empty()) throw std::runtime_error("set sharding is not available for Minkowski sets");
      if (mk_)
        for (auto& st : sets_) st.ata_off = default_ata_offsets(st);
    }
    Nx_ = mk_ ? 2 * G_.N : G_.N;
    if (Nx_ >= (1ll << 31)) throw std::runtime_error("2^31 unknowns or more are not supported");
    if (p_n_ > 99) throw std::runtime_error("at most 99 sets (PARSDMM_initialize.jl:217)");
    if (comm_) {                            // this context is one rank of a sharded solve
      if (mk_) throw std::runtime_error("the sharded solve is not available for Minkowski sets");
      if (stencil_q_) throw std::runtime_error("the sharded solve needs the CDS form of Q");
      if (owned_.empty()) {                 // PARSDMM_initialize.jl:78-80 deals one set per worker; here round robin
        owned_.resize(p_n_);
        for (int i = 0; i < p_n_; ++i) owned_[i] = (i % comm_->world) == comm_->rank;
      }
    }
    if (!owned_.empty())
      for (int i = 0; i < p_n_; ++i) sets_[i].owned = owned_[i] != 0;
    // Slab decomposition of the WHOLE iteration (sipx_set_decomp): every rank holds every set and works on its z-slab of the
    // globally indexed arrays; no N-vector crosses the fabric any more.  For the sets whose projector needs no
    // more than sums over the grid: element-wise ones, l1 / l2 balls and the annulus on the identity or D_x / D_y / D_z / TV.
    slab_ = slab_req_ && comm_ != nullptr;
    if (slab_) {
      if (mk_ || stencil_q_) throw std::runtime_error("the slab decomposition is not available for Minkowski contexts or the stencil form of Q");
      int nfan = 0;
      for (int i = 0; i < p_n_; ++i) {
        SetState<T>& s = sets_[i];
        if (s.custom)
          throw std::runtime_error("the slab decomposition has no form for a caller-supplied sparse operator (set " + std::to_string(i) + "): use the set decomposition");
        const bool sliced = (s.ext_kind == EXT_RANK || s.ext_kind == EXT_NUCLEAR) && s.spec.mode == SIPX_MODE_SLICE &&
                            s.spec.dir == ndim_ - 1 && s.ident;
        if (const char* fo = std::getenv("SIPX_FAN_OVERLAP")) fan_overlap_ = fo[0] != '0';
        const char* gc = std::getenv("SIPX_SLAB_CARD_GATHER");      // 1: cardinality through an owner rank as well (A/B switch, tests)
        const char* gd = std::getenv("SIPX_SLAB_DFT_GATHER");       // 1: the l1-DFT set through an owner rank (A/B switch, tests)
        if (sliced) s.slab_ext = true;
        else if (s.prox == PX_CARD && !s.ext_kind && !(gc && gc[0] == '1')) s.slab_card = true;
        else if (s.ext_kind == EXT_L1_DFT && ndim_ == 3 && s.ident && G_.n[0] >= 2 && !(gd && gd[0] == '1')) s.slab_dft = true;
        else if (s.ext_kind || s.prox == PX_CARD) { s.fan = true; s.fan_owner = (nfan++) % comm_->world; }
        if (s.fan && s.nblk > 1) throw std::runtime_error("internal: a gathered set with more than one operator block");
        slab_loose_ |= s.slab_ext || s.fan || s.slab_card || s.slab_dft;
        s.owned = true;
      }
      // (the searches with their collectives run on the engine stream, in one order on every rank; the y/l updates that
      // follow have no collectives inside and are dealt onto the set streams as usual)
    }
    const long long N = G_.N;
    // rho, gamma (PARSDMM_initialize.jl:58-63,107-114,159)
    rho_.resize(p_n_);
    gamma_.resize(p_n_);
    if (n_rho == 1) std::fill(rho_.begin(), rho_.end(), (T)rho_ini[0]);
    else if (n_rho == p_n_) for (int i = 0; i < p_n_; ++i) rho_[i] = (T)rho_ini[i];
    else throw std::runtime_error("rho_ini must have 1 or p entries");
    T g0 = (T)gamma_ini;
    any_ncvx_ = false;
    for (int i = 0; i < pp_n_; ++i) any_ncvx_ |= sets_[i].ncvx != 0;
    if (any_ncvx_) g0 = T(0.75);
    std::fill(gamma_.begin(), gamma_.end(), g0);

    // Q offsets first: the SpMV inputs (x, p) carry a zero halo of max|offset| on both sides so that the
    // kernels need no bounds checks on neighbour loads
    plan_Q_offsets();
    halo_ = 4;
    for (int b = 0; b < cds_.d; ++b) halo_ = std::max<long long>(halo_, std::llabs(cds_.off[b]));
    for (int a = 0; a < 3; ++a) halo_ = std::max<long long>(halo_, G_.st[a]);
    halo_ = (halo_ + 3) / 4 * 4;
    r0_ = 0; r1_ = Nx_;
    long long Npad = Nx_;
    if (comm_) {
      // z-slabs of the x-step: ceil(n_last / world) planes per rank (the last ranks may hold fewer, or none); the exchange
      // buffers (rhs, x) are padded to world equal chunks, the pad stays zero
      const long long nlast = G_.n[ndim_ - 1];
      plane_ = N / nlast;
      long long maxoff = 0;
      for (int b = 0; b < cds_.d; ++b) maxoff = std::max<long long>(maxoff, std::llabs(cds_.off[b]));
      if (maxoff > plane_) throw std::runtime_error("the sharded solve needs operators whose A'A reaches no further than one plane of the grid");
      const long long planes = (nlast + comm_->world - 1) / comm_->world;
      chunk_ = planes * plane_;
      Npad = chunk_ * comm_->world;
      r0_ = std::min<long long>(N, (long long)comm_->rank * chunk_);
      r1_ = std::min<long long>(N, (long long)(comm_->rank + 1) * chunk_);
      prev_ = (r0_ > 0 && r0_ < N) ? comm_->rank - 1 : -1;
      next_ = (r1_ < N && r1_ > r0_) ? comm_->rank + 1 : -1;
      qr0_ = r1_ > r0_ ? std::max<long long>(0, r0_ - maxoff) : 0;      // the symmetric read reaches |offset| rows back
      qr1_ = r1_ > r0_ ? r1_ : 0;
      SIPX_HIP(hipStreamCreateWithFlags(&cstream_, hipStreamNonBlocking));
      for (auto& e : ev_c_) SIPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    } else {
      qr0_ = 0; qr1_ = N;
    }
    Gr_ = G_;
    Gyl_ = G_;
    if (slab_) {
      Gr_.e0 = r0_; Gr_.e1 = r1_;
      // k_yl recomputes the last plane of the rank below (y, l of a difference along the slab direction are read one plane
      // back by the adjoint stencils): bit for bit what that rank computes, instead of an exchange of y, l and y - y_old
      Gyl_.e0 = (prev_ >= 0) ? r0_ - plane_ : r0_; Gyl_.e1 = r1_; Gyl_.s0 = r0_;
      if (r1_ <= r0_) { Gr_.e0 = Gr_.e1 = 0; Gyl_.e0 = Gyl_.e1 = 0; }
    }
    wlo_ = -halo_; whi_ = Npad + halo_;
    bool sparse_wanted = false;
    if (slab_ && !slab_full_req_) {
      const char* e = std::getenv("SIPX_SLAB_LOCAL");        // 0: full-size arrays on every rank, as before (A/B switch)
      int vmm = 0;
      (void)hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, device_);
      sparse_wanted = vmm != 0 && !(e && e[0] == '0');
      for (const auto& st : sets_)
        if (st.prox == PX_BOUNDS_VEC || (!st.two_pass && st.nblk > 0) || !st.host_ata.empty()) sparse_wanted = false;
      // (lists with materialised sets keep sparse arrays too: their vectors are addressed by global index -- loose_v_ / loose_w_ --
      //  and whole only on the rank that projects a gathered set.  SIPX_SLAB_LOOSE_SPARSE=0: whole arrays for such lists, A/B switch)
      if (slab_loose_)
        if (const char* ls = std::getenv("SIPX_SLAB_LOOSE_SPARSE")) if (ls[0] == '0') sparse_wanted = false;
    }
    if (comm_) {                                            // (every rank takes the same branch: the verdict is all-reduced)
      bool any_dft = false;
      for (const auto& st : sets_) any_dft |= st.slab_dft;
      comm_self_test(sparse_wanted, any_dft);
      if (selftest_alltoall_failed_) {       // the transposition of the slab-decomposed DFT does not work here: an owner rank projects such sets
        int nfan = 0;
        for (auto& st : sets_) nfan += st.fan ? 1 : 0;
        for (auto& st : sets_)
          if (st.slab_dft) { st.slab_dft = false; st.fan = true; st.fan_owner = (nfan++) % comm_->world; }
      }
    }
    // Everything from here to the initial feasibility allocates and uploads -- no collective.  A rank that fails in there (out of
    // device memory: the likeliest rank-local failure of a first multi-GPU run) used to throw by itself while the others went on
    // into the collectives of the initial feasibility and waited for it for ever; now every rank reports to one all-reduce on the
    // self-test's plain buffer and all of them throw together (bench.py then falls back to the next decomposition on every rank).
    std::string alloc_err;
    try {
    if (const char* f = std::getenv("SIPX_FINALIZE_FAIL_RANK"))       // tests: this rank "runs out of memory"
      if (comm_ && std::atoi(f) == comm_->rank) throw std::runtime_error("test hook: out of device memory");
    if (slab_ && !slab_full_req_) {
      slab_local_ = sparse_wanted && !selftest_mapped_failed_;
      for (const auto& st : sets_) {
        // per-element bound vectors arrive whole; the initial feasibility of an element-wise set on a difference operator
        // takes the whole-grid kernels: such lists keep full-size arrays
        if (st.prox == PX_BOUNDS_VEC || (!st.two_pass && st.nblk > 0) || !st.host_ata.empty()) slab_local_ = false;
      }
      if (slab_local_) {
        // what a rank's kernels touch: its planes, one plane behind (forward differences, the neighbour's copy of x and p), two
        // planes in front (the recomputed last plane of the rank below, and the plane the z-march loads in front of THAT one)
        const long long a = r1_ > r0_ ? r0_ : N, b = r1_ > r0_ ? r1_ : N;
        wlo_ = std::max<long long>(-halo_, a - 2 * plane_ - 64);
        whi_ = std::min<long long>(Npad + halo_, b + plane_ + 64);
      }
    }
    for (int k = 0; k < 3; ++k) { xr_base_[k] = galloc(Npad + 2 * halo_, halo_, 1, 0); xr_[k] = xr_base_[k] + halo_; }
    x_cur_ = 0; x_snap_ = -1;
    x_ = xr_[0]; xold_ = x_;                // (no x-step yet: x_old names x itself)
    p_base_ = galloc(Nx_ + 2 * halo_, halo_, 1, 0); p_ = p_base_ + halo_;
    rhs_ = galloc(Npad, 0, 1, 0);
    if (mk_) { w_base_ = dalloc<T>(N + 2 * halo_); w_ = w_base_ + halo_; }   // u + v, read through the stencils
    m_base_ = galloc(N + 2 * halo_, halo_, 1, 0); m_ = m_base_ + halo_;   // forward stencils of A m read past the end
    r_base_ = galloc(Nx_ + 2 * halo_, halo_, 1, 0); r_ = r_base_ + halo_;      // (halo: the fused CG product reads r through the bands)
    Ap_ = galloc(Nx_, 0, 1, 0);
    {
      // CG iterations from the second on as ONE kernel (scalar step + product on p = r + beta p_old formed on the fly,
      // k_cds_fused): one launch and a host round trip less per iteration -- what a launch-bound grid (2048^2) is made of --
      // against a product that reads two vectors instead of one.  SIPX_CG_FUSED=0 / 1 forces it off / on.
      const char* e = std::getenv("SIPX_CG_FUSED");
      const bool small = Nx_ <= (1ll << 23);
      // (the z-marching product has a fused form of its own, k_cds_march<MODE 3>: there the fusion also saves traffic -- 8 N w
      //  instead of the 9 of product + p-update -- so it is the default at every size for the matrices the march takes)
      cg_fused_ = !comm_ && !stencil_q_ && (e ? e[0] == '1' : (small || cds_.march != 0));
      if (cg_fused_) { p2_base_ = dalloc<T>(Nx_ + 2 * halo_); p2_ = p2_base_ + halo_; }
    }
    {
      const long long c0 = std::max<long long>(0, wlo_), c1 = std::min<long long>(N, whi_);      // (sparse arrays: the rank's share only)
      if (c1 > c0) SIPX_HIP(hipMemcpy(m_ + c0, (const T*)m + c0, (c1 - c0) * sizeof(T), hipMemcpyHostToDevice));
    }
    {
      // x0 mode of the one-sweep update: when EVERY y/l update of this context goes through the sweep (its block layout is
      // compiled in, no set needs the per-set kernels on feasibility iterations), s_0 = A x_0 is recomputed from a snapshot of
      // x instead of being stored per set: two N-vectors instead of one M_i-vector per set, and 8 N w less traffic on every
      // Barzilai-Borwein iteration of the headline list.  SIPX_X0_SNAPSHOT=0 keeps the per-set s_0 arrays (A/B switch, tests).
      const char* e = std::getenv("SIPX_X0_SNAPSHOT");
      const char* mu = std::getenv("SIPX_YL_MULTI");            // 0: one k_yl launch per set on every iteration (A/B switch, tests)
      yl_multi_ = !(mu && mu[0] == '0');
      if (const char* ra = std::getenv("SIPX_RESID_AHEAD")) resid_ahead_ = ra[0] != '0';      // A/B switch
      if (const char* qf = std::getenv("SIPX_Q_FUSED")) q_fused_ = qf[0] != '0';               // A/B switch
      // the lean first passes of the l1 searches in one sweep: pays where the re-reads of x are real traffic (512^3, settled
      // iterations: 133 -> 139 it/s); at 256^3 three concurrent per-set passes on their own streams are as fast or faster
      // (1028 against 1012 it/s settled, default window equal), so it is the default above 2^24 grid points only
      lean_multi_ = G_.N > (1ll << 24);
      if (const char* lm = std::getenv("SIPX_LEAN_MULTI")) lean_multi_ = lm[0] != '0';         // A/B switch, tests
      {
        // which sets the sweep takes: all of them (the layouts of C2 / C3 / C5 and of the short lists), or -- one rank only -- the
        // element-wise and l1 / l2 terms of a list with sets it cannot take (C4), provided the layout of that subset is compiled in
        MultiArgs<T> probe0;
        const bool any_sweep = sweep_applicable(0, probe0, true);
        has_loose_ = false;
        for (auto& st : sets_) {
          st.in_sweep = any_sweep && sweep_eligible(st);
          has_loose_ |= any_sweep && st.owned && !st.in_sweep;
        }
        const char* sp = std::getenv("SIPX_SWEEP_PARTIAL");      // 0: such lists keep one k_yl launch per set (A/B switch, tests)
        if (has_loose_ && sp && sp[0] == '0') {
          for (auto& st : sets_) st.in_sweep = false;
          has_loose_ = false;
          yl_multi_ = false;
        }
        sweep_partial_ = has_loose_;
      }
      MultiArgs<T> probe;
      x0_mode_ = !(e && e[0] == '0') && !has_loose_ && sweep_applicable(SIPX_YL_FEAS | SIPX_YL_BB, probe, true);
      // Set streams when the sweep does the updates: all that runs on them is the threshold / scale searches, chains of short
      // kernels whose latencies should overlap -- every searching set a stream of its own (up to three; one of them the engine
      // stream, so that its search starts without a cross-stream dependency), the other sets on the engine stream.  256^3, C3:
      // 715 -> 760 it/s with three instead of two; 512^3 unchanged; four lose (no search left on the engine stream); 2048^2 with
      // its one searching set keeps two.  Without the sweep the per-set y/l kernels run there too: two streams, sets dealt round robin.
      MultiArgs<T> probe2;
      search_streams_ = !set_streams_forced_ && sweep_applicable(SIPX_YL_BB, probe2, true);
      if (search_streams_) {
        int ntp = 0;
        for (const auto& st : sets_) ntp += st.two_pass ? 1 : 0;
        n_set_streams_ = std::max(2, std::min(3, ntp));
      }
      {
        // One rank and EVERY update through the sweep: the searches of all two-pass sets as one chain of launches on the engine
        // stream (batched_searches) -- no set streams at all.  SIPX_SEARCH_BATCH=0 keeps the per-set chains (A/B switch, tests).
        int ntp = 0;
        for (const auto& st : sets_) ntp += (st.two_pass && st.in_sweep) ? 1 : 0;
        const char* sb = std::getenv("SIPX_SEARCH_BATCH");
        MultiArgs<T> probe3;
        search_batch_ = !comm_ && !(sb && sb[0] == '0') && ntp >= 1 && ntp <= SPEC_MAX_SETS &&
                        sweep_applicable(SIPX_YL_FEAS | SIPX_YL_BB, probe3, true);
        if (search_batch_) set_streams_ = false;
        if (const char* fs = std::getenv("SIPX_FEAS_SAMPLE")) feas_sample_ = fs[0] != '0';
        if (const char* pm = std::getenv("SIPX_PASS_MULTI")) pass_multi_ = pm[0] == '1';
      }
      {
        MultiArgs<T> probe4;
        sweep_plain_ = sweep_applicable(0, probe4, true);
      }
      for (const auto& st : sets_) slab_dist_logs_ |= slab_ && !mk_ && st.is_dist;
      // (the snapshot of x of the x0 mode is a member of the ring of x buffers: nothing to allocate)
    }
    long long maxpad = N;
    for (auto& s : sets_) maxpad = std::max(maxpad, s.Mpad);
    if (comm_) {
      for (int i = 0; i < p_n_; ++i) {
        SetState<T>& s = sets_[i];
        s.owner_rank = i % comm_->world;
        const bool sliced = (s.ext_kind == EXT_RANK || s.ext_kind == EXT_NUCLEAR) && s.spec.mode == SIPX_MODE_SLICE &&
                            s.spec.dir == ndim_ - 1 && s.ident;
        if (!sliced || slab_) continue;               // (slab-decomposed: SetState::slab_ext, no exchange at all)
        if (s.owned != (s.owner_rank == comm_->rank))
          throw std::runtime_error("a slice-wise rank / nuclear set is projected by all ranks: it needs the default set ownership (set i on rank i mod world)");
        s.dist_ext = true;
        need_ext_ = true;
        maxpad = std::max(maxpad, Npad);            // v travels through the padded exchange layout of x
      }
    }
    if (slab_loose_) need_ext_ = true;              // (the feasibility estimates keep a copy of s = A x)
    if (slab_) {
      maxpad = std::max(maxpad, Npad);              // slabs of y, l are gathered through the padded exchange layout at download
      hooks_.world = comm_->world; hooks_.rank = comm_->rank; hooks_.user = comm_.get();
      hooks_.allreduce_sum = [](void* u, double* buf, size_t n, hipStream_t q) { static_cast<Comm*>(u)->allreduce_sum(buf, n, SIPX_F64, q); };
      hooks_.allgather = [](void* u, void* buf, size_t chunk, int f64, hipStream_t q) {
        static_cast<Comm*>(u)->allgather(buf, chunk, f64 ? SIPX_F64 : SIPX_F32, q);
      };
      // what a search may gather inside its final bracket over ALL ranks (and the size of a rank's full-size exchange segment:
      // the all-gather moves whole segments): 2^17 magnitudes up to 256^3, N / 512 above (512^3: 2^18), at most 2^20
      hooks_.gcap = std::min<long long>(std::min<long long>(1ll << 20, std::max<long long>(1ll << 17, G_.N / 512)), (maxpad + 3) / 4 * 4);
      if (const char* e = std::getenv("SIPX_GATHER_CAP"))                      // tests: a segment small enough to overflow
        if (std::atoll(e) >= 4) hooks_.gcap = std::atoll(e) / 4 * 4;
      int n2 = 0, nl1 = 0;
      for (auto& s : sets_) { n2 += s.two_pass ? 1 : 0; nl1 += (s.two_pass && s.prox == PX_L1) ? 1 : 0; }
      // the searches of all sets run in lock step: one staging buffer for their sums (one all-reduce per stage), one exchange
      // buffer with a segment per l1 set and rank (one all-gather)
      gbuf_ = dalloc<T>((size_t)comm_->world * std::max(nl1, 1) * (hooks_.gcap + GATHER_HDR));
      hooks_.gbuf = gbuf_;
      // speculative exchange: a small segment per two-pass set and rank (header with the rank's sums + what its first pass
      // gathered inside the speculative range: a few thousand magnitudes once theta moves slowly)
      // (a rank's share of what the full-size segment holds, times two for uneven shares; at least 16 K values)
      hooks_.fcap = std::min<long long>(hooks_.gcap, std::max<long long>(1ll << 14, (2 * hooks_.gcap / comm_->world + 3) / 4 * 4));
      if (const char* e = std::getenv("SIPX_GATHER_FAST_CAP"))
        if (std::atoll(e) >= 4) hooks_.fcap = std::min<long long>(hooks_.gcap, std::atoll(e) / 4 * 4);
      if (const char* e = std::getenv("SIPX_SPEC_EXCHANGE")) spec_exchange_ = std::atoi(e) != 0;
      if (const char* e = std::getenv("SIPX_SPEC_BATCH")) spec_batch_ = std::atoi(e) != 0;
      if (const char* e = std::getenv("SIPX_SLAB_LEAN_MULTI")) slab_lean_multi_ = std::atoi(e) != 0;
      fbuf_ = dalloc<T>((size_t)comm_->world * std::max(n2, 1) * (hooks_.fcap + fast_hdr<T>()));
      stage_ = dalloc<double>((size_t)std::max(n2, 1) * (PREP_SLOTS + 1 + 2 * comm_->world));
      sstage_ = dalloc<double>((size_t)std::max(n2, 1) * (2 * SAMPLE_BINS + 3));
    }
    const long long fullpad = maxpad;   // (what a whole vector of the exchange layout takes: the owner of a gathered set)
    if (slab_local_) {                 // the whole-array scratch is not needed: searches compact at most what the rank's planes hold
      int nbmax = 1;
      for (auto& s : sets_) nbmax = std::max(nbmax, s.nblk_or1());
      // (... or what the exchange of a search strings together from all ranks: at most gcap magnitudes, by the search's own rule)
      maxpad = std::min<long long>(maxpad, std::max<long long>((long long)nbmax * (whi_ - wlo_), hooks_.gcap + 64));
    }
    scr_v_ = dalloc<T>(maxpad);
    scr_c_ = dalloc<T>(maxpad);
    scr_c_len_ = maxpad;
    if (need_idx_) scr_i_ = dalloc<long long>(maxpad);
    if (need_ext_) scr_w_ = dalloc<T>(maxpad);
    // the vectors the materialised sets of a slab-decomposed list are stored into, addressed by GLOBAL index (v, P(v), s = A x and
    // its copy): the engine-wide scratch where that is whole; with sparse arrays a rank that projects a gathered set holds them
    // whole, every other rank its planes only
    loose_v_ = scr_v_; loose_w_ = scr_w_;
    if (slab_local_ && slab_loose_) {
      bool owner = false;
      for (auto& st : sets_) owner |= st.fan && st.fan_owner == comm_->rank;
      loose_whole_ = owner;
      loose_v_ = owner ? dalloc<T>(fullpad) : loose_alloc(Npad);
      loose_w_ = owner ? dalloc<T>(fullpad) : loose_alloc(Npad);
      loose_owned_ = true;
    }
    // (the reduced per-set sums sit right behind the CG partials: sharded, ONE all-reduce can carry both, see argmin_x_head)
    part_cg_ = dalloc<double>(2 * NB + (size_t)(p_n_ + 1) * SLOTS);
    part_tmp_ = dalloc<double>((size_t)(PREP_SLOTS + 2) * NB);
    part_sets_ = dalloc<double>((size_t)(p_n_ + 1) * SLOTS * NB);   // + one group of slots for whole-x sums
    maxpart_ = dalloc<T>(2 * NB);     // per-block max | per-block smallest non-zero magnitude
    cg_dev_ = dalloc<CgState<T>>(1);
    dres_ = part_cg_ + 2 * NB;
    SIPX_HIP(hipHostMalloc((void**)&cg_host_, 2 * sizeof(CgState<T>), hipHostMallocDefault));
    std::memset(cg_host_, 0, 2 * sizeof(CgState<T>));
    SIPX_HIP(hipHostMalloc((void**)&ticket_, 64, hipHostMallocDefault));
    std::memset((void*)ticket_, 0xff, 64);
    for (int k = 0; k < 2; ++k) SIPX_HIP(hipEventCreateWithFlags(&cg_ev_[k], hipEventDisableTiming));
    SIPX_HIP(hipEventCreateWithFlags(&ev_cgb_, hipEventDisableTiming));
    SIPX_HIP(hipEventCreateWithFlags(&ev_sums_, hipEventDisableTiming));
    SIPX_HIP(hipHostMalloc((void**)&hres_, sizeof(double) * (p_n_ + 1) * SLOTS, hipHostMallocDefault));
    std::memset(hres_, 0, sizeof(double) * (p_n_ + 1) * SLOTS);
    SIPX_HIP(hipHostMalloc((void**)&sums_word_, 64, hipHostMallocDefault));
    std::memset((void*)sums_word_, 0, 64);
    sums_ticket_ = dalloc<unsigned>(1);
    SIPX_HIP(hipHostMalloc((void**)&hlean_, sizeof(int) * (p_n_ + 1), hipHostMallocDefault));
    std::memset((void*)hlean_, 0, sizeof(int) * (p_n_ + 1));
    SIPX_HIP(hipHostMalloc((void**)&hovf_, sizeof(int) * (p_n_ + 1), hipHostMallocDefault));
    std::memset((void*)hovf_, 0, sizeof(int) * (p_n_ + 1));
    SIPX_HIP(hipHostMalloc((void**)&hverd_, sizeof(unsigned) * (p_n_ + 1), hipHostMallocDefault));
    std::memset((void*)hverd_, 0, sizeof(unsigned) * (p_n_ + 1));
    {
      const char* e = std::getenv("SIPX_L1_SAMPLE");
      l1_sample_ = !(e && e[0] == '0');
      const char* r = std::getenv("SIPX_L1_SAMPLE_RUNS");       // tests: sample small grids too
      l1_sample_runs_ = r ? std::atoll(r) : 0;
    }
    for (int k = 0; k < 2 * MAXMARK; ++k) {      // two sets of section marks: a step never waits for its own timing
      hipEvent_t e;
      SIPX_HIP(hipEventCreate(&e));
      ev_.push_back(e);
    }

    const bool warm = !zero_ini_guess;     // PARSDMM_initialize.jl:304-313
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>& s = sets_[i];
      // explicit AtA bands are kept on the device (every rank: Q is global); descriptor-generated
      // ones are never stored -- the fused Q update regenerates their values on the fly
      if (!s.host_ata.empty()) {
        s.ata = dalloc<T>((size_t)N * s.ata_off.size(), false);
        SIPX_HIP(hipMemcpy(s.ata, s.host_ata.data(), s.host_ata.size() * sizeof(T), hipMemcpyHostToDevice));
        s.host_ata.clear();
        s.host_ata.shrink_to_fit();
      }
      if ((s.dist_ext || s.slab_ext) && r1_ > r0_) {              // this rank's share of the slices: the projector on the slab grid
        ExtSpec sp = s.spec;
        const long long planes = (r1_ - r0_) / plane_;
        sp.G.n[ndim_ - 1] = planes;
        sp.G.N = planes * plane_;
        sp.dims[ndim_ - 1] = planes;
        s.ext = std::make_shared<ExtProj<T>>(sp, stream_);
      }
      if (!s.owned) continue;
      // vectors read through adjoint stencils (w[g - stride]) carry a zero front halo: no bounds checks in the kernels
      auto halloc = [&](long long n) {
        T* base = galloc(n + halo_, halo_, s.nblk_or1(), N);
        s.halo_allocs.push_back(base);
        return base + halo_;
      };
      s.y = halloc(s.Mpad); s.l = halloc(s.Mpad);
      s.lh0 = galloc(s.Mpad, 0, s.nblk_or1(), N);
      if (!x0_mode_) s.s0 = galloc(s.Mpad, 0, s.nblk_or1(), N);
      s.y0 = halloc(s.Mpad); s.l0 = halloc(s.Mpad);       // take turns with y, l as the current iterate: same halo
      if (!s.ident) s.dy = halloc(s.Mpad);
      // the third pair of the one-sweep update (two plain iterations in a row: the snapshot has to survive in the other pair
      // and the sweep never writes in place) -- allocated HERE, so that running out of memory is an error of sipx_finalize and
      // not of an iteration whose state has already advanced
      if (sweep_plain_ && s.in_sweep) { s.y2 = halloc(s.Mpad); s.l2 = halloc(s.Mpad); }
      if (s.custom) upload_custom(s);
      if (s.slab_dft) {
        const long long nn[3] = {G_.n[0], G_.n[1], G_.n[2]};
        s.ddft = std::make_shared<DistDft<T>>(nn, r0_ / plane_, r1_ / plane_, chunk_ / plane_, comm_->world, comm_->rank, (double)s.spec.pmax, stream_);
      }
      if (s.ext_kind && !s.dist_ext && !s.slab_ext && !s.slab_dft && !(s.fan && s.fan_owner != comm_->rank)) {      // (a gathered set: its owner only)
        s.spec.lb = s.host_lb.empty() ? nullptr : s.host_lb.data();
        s.spec.ub = s.host_ub.empty() ? nullptr : s.host_ub.data();
        s.spec.basis = s.host_basis.empty() ? nullptr : s.host_basis.data();
        s.ext = std::make_shared<ExtProj<T>>(s.spec, stream_);
        if (s.ext_kind == EXT_HISTOGRAM) { s.host_lb.clear(); s.host_ub.clear(); }
        s.host_basis.clear();
        s.host_basis.shrink_to_fit();
      }
      if (s.fan) {
        // the gathered sets are collected FIRST in an update (update_y_l), their owners project on a stream of their own while
        // every rank goes on with the sets of its own slab: each such set keeps its own whole-size vector, the fan stream its scratch
        s.fanv = (s.fan_owner == comm_->rank || !slab_local_) ? dalloc<T>(fullpad) : loose_alloc(Npad);
        SIPX_HIP(hipEventCreateWithFlags(&s.fan_ev, hipEventDisableTiming));
        if (s.fan_owner == comm_->rank && !fan_st_) {
          SIPX_HIP(hipStreamCreateWithFlags(&fan_st_, hipStreamNonBlocking));
          SIPX_HIP(hipEventCreateWithFlags(&fan_fork_, hipEventDisableTiming));
          fan_ptmp_ = dalloc<double>((size_t)(PREP_SLOTS + 2) * NB);
          fan_mpart_ = dalloc<T>(2 * NB);
          fan_c_ = dalloc<T>(fullpad);
        }
      }
      if (s.two_pass) {
        s.ps = dalloc<ProjScalars<T>>(1);
        s.psf = dalloc<ProjScalars<T>>(1);
        K<T>::ps_init(stream_, s.ps, scr_i_);
        K<T>::ps_init(stream_, s.psf, scr_i_);
      }
      if ((slab_ || search_batch_) && s.two_pass) {        // searches in lock step: every set keeps its own partial slots and gather buffer
        s.ptmp = dalloc<double>((size_t)(PREP_SLOTS + 2) * NB);
        s.mpart = dalloc<T>(2 * NB);
        s.cbuf_len = slab_local_ ? std::min<long long>(s.Mpad, std::max<long long>((long long)s.nblk_or1() * (whi_ - wlo_), hooks_.gcap + 64)) : s.Mpad;
        s.cbuf = dalloc<T>(s.cbuf_len);
      }
      const bool had_scratch = s.ptmp != nullptr;
      if (set_streams_ && !s.ext_kind && s.prox != PX_CARD) {     // those two share the engine-wide scratch: main stream
        if ((int)pool_.size() < n_set_streams_) {
          // the engine stream itself is the first of the set streams: the set dealt onto it starts right behind the x-step,
          // with no cross-stream dependency (about 20 us) on the critical path; that of the others hides behind its work
          hipStream_t q = stream_;
          if (!pool_.empty()) SIPX_HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
          pool_.push_back(q);
        }
        if (search_streams_) {               // searching sets: streams 1, 2, 0, 1, ... ; the others: the engine stream
          while ((int)pool_.size() < n_set_streams_) {
            hipStream_t q2 = stream_;
            if (!pool_.empty()) SIPX_HIP(hipStreamCreateWithFlags(&q2, hipStreamNonBlocking));
            pool_.push_back(q2);
          }
          s.st = s.two_pass ? pool_[(size_t)(++search_next_) % pool_.size()] : pool_[0];
        } else {
          s.st = pool_[pool_next_++ % pool_.size()];
        }
        SIPX_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
        if (s.two_pass && !had_scratch) {
          s.ptmp = dalloc<double>((size_t)(PREP_SLOTS + 2) * NB);
          s.mpart = dalloc<T>(2 * NB);
          s.cbuf = dalloc<T>(s.Mpad);
        }
      }
      if (s.prox == SIPX_PROJ_BOUNDS_VEC) {
        s.lb = dalloc<T>(s.Mpad); s.ub = dalloc<T>(s.Mpad);
        upload_rows(s, s.host_lb.data(), s.lb);
        upload_rows(s, s.host_ub.data(), s.ub);
      }
      if (warm && l0 && l0[i]) upload_rows_ranged(s, (const T*)l0[i], s.l);
      if (warm && y0 && y0[i]) upload_rows_ranged(s, (const T*)y0[i], s.y);
    }
    if (search_batch_) {                    // the sets' header segments and decision registers of the batched searches
      int n2 = 0;
      for (auto& s : sets_) n2 += s.two_pass ? 1 : 0;
      fbuf_ = dalloc<T>((size_t)n2 * fast_hdr<T>());
      stage_ = dalloc<double>((size_t)n2 * (PREP_SLOTS + 1 + 2));
    }
    if (warm && x0) {                        // Minkowski: [u; v], 2N entries (sparse arrays: the rank's share)
      const long long c0 = slab_local_ ? std::max<long long>(0, wlo_) : 0, c1 = slab_local_ ? std::min<long long>(Nx_, whi_) : Nx_;
      if (c1 > c0) SIPX_HIP(hipMemcpy(x_ + c0, (const T*)x0 + c0, (c1 - c0) * sizeof(T), hipMemcpyHostToDevice));
    }

    assemble_Q();
    if (q_fused_ && !stencil_q_ && !comm_ && cds_.march != 0) Q2_ = dalloc<T>((size_t)Nx_ * cds_.d);     // (SIPX_Q_FUSED=1 only)
    {
      // The lane: one rank, a list the sweep takes in part (C4) -- the slice-rank / nuclear-norm set, a chain of batched GEMMs and
      // small factorisations with host round trips in it (ext_proj.hip), runs on a stream of its own, queued by a host thread of
      // its own, beside the searches, the sweep and the other loose sets on the engine stream; its update only needs x.
      // SIPX_RANK_LANE=0: in turn on the engine stream (A/B switch, tests).
      const char* ln = std::getenv("SIPX_RANK_LANE");
      lane_set_ = -1;
      if (!comm_ && !mk_ && sweep_partial_ && !(ln && ln[0] == '0'))
        for (int i = 0; i < p_n_ && lane_set_ < 0; ++i) {
          const SetState<T>& s = sets_[i];
          if (s.owned && !s.in_sweep && !s.dist_ext && s.ident && !s.custom && s.ext && (s.ext_kind == EXT_RANK || s.ext_kind == EXT_NUCLEAR)) lane_set_ = i;
        }
      // Slab-decomposed (round 5): the slice-rank set of a long list projects the z-slices of the rank's own planes -- no collective
      // anywhere in its update -- so it takes the same lane, beside the lock-step searches, the sweep and the transform set with
      // their collectives on the engine stream.  A rank's share of C4 is where this pays most: a call on 64 slices is a chain of
      // small launches with host round trips in it that leaves most of the chip idle.
      if (comm_ && slab_ && slab_loose_ && !mk_ && sweep_partial_ && !(ln && ln[0] == '0'))
        for (int i = 0; i < p_n_ && lane_set_ < 0; ++i) {
          const SetState<T>& s = sets_[i];
          if (s.owned && !s.in_sweep && s.slab_ext && s.ident && !s.custom && s.ext && (s.ext_kind == EXT_RANK || s.ext_kind == EXT_NUCLEAR)) lane_set_ = i;      // (s.ext: a rank without planes has no projector and nothing to overlap)
        }
      if (lane_set_ >= 0) {
        // A stream of the highest priority: the runtime keeps its hardware queues per priority, so the lane can never share a
        // queue with the engine stream (streams of one priority are dealt onto four queues in turn; in the slab-decomposed
        // context the lane had landed on the engine stream's queue and ran strictly behind it: tools/lane_overlap.py, 0.00 ms
        // together), and the chain of small launches that is the iteration's critical path does not wait behind the sweep.
        // SIPX_LANE_PRIORITY=0: a plain stream (A/B switch).
        const char* lp = std::getenv("SIPX_LANE_PRIORITY");
        int pr_least = 0, pr_greatest = 0;
        SIPX_HIP(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
        if (lp && lp[0] == '0') SIPX_HIP(hipStreamCreateWithFlags(&lane_st_, hipStreamNonBlocking));
        else SIPX_HIP(hipStreamCreateWithPriority(&lane_st_, hipStreamNonBlocking, lp && lp[0] == 'l' ? pr_least : pr_greatest));
        SIPX_HIP(hipEventCreateWithFlags(&lane_fork_, hipEventDisableTiming));
        SIPX_HIP(hipEventCreateWithFlags(&lane_ev_, hipEventDisableTiming));
        // (slab-decomposed: a vector of the exchange layout addressed by global index, like loose_v_)
        if (slab_) lane_v_ = (slab_local_ && !loose_whole_) ? loose_alloc(Npad) : dalloc<T>((size_t)std::max<long long>(fullpad, sets_[lane_set_].Mpad));
        else lane_v_ = dalloc<T>((size_t)sets_[lane_set_].Mpad);
      }
    }
    } catch (const std::exception& ex) {
      if (!comm_ || comm_->world <= 1 || !agree_buf_) throw;
      alloc_err = ex.what();
    }
    if (comm_ && comm_->world > 1 && agree_buf_) {
      const double flag = alloc_err.empty() ? 0.0 : 1.0;
      double sum = flag;
      try {
        SIPX_HIP(hipMemcpy(agree_buf_, &flag, sizeof(double), hipMemcpyHostToDevice));
        comm_->allreduce_sum(agree_buf_, 1, SIPX_F64, stream_);
        SIPX_HIP(hipStreamSynchronize(stream_));
        SIPX_HIP(hipMemcpy(&sum, agree_buf_, sizeof(double), hipMemcpyDeviceToHost));
      } catch (const std::exception& ex) {
        throw std::runtime_error(std::string("sipx_finalize: the ranks could not agree on the outcome of their allocations (") + ex.what() + ")" +
                                 (alloc_err.empty() ? "" : "; this rank: " + alloc_err));
      }
      if (sum != 0.0)
        throw std::runtime_error("sipx_finalize failed on " + std::to_string((int)sum) + " of " + std::to_string(comm_->world) + " ranks while allocating" +
                                 (alloc_err.empty() ? std::string(" (not on this one)") : ": " + alloc_err));
    }
    finalized_ = true;

    initial_feasibility(feasibility_initial);
  }

  // ------------------------------------------------------------------------------------------
  // sipx_reset (round 5): the SAME sets on the SAME grid with a new model m (and, optionally, a new warm start and rho_ini) --
  // what the reference's callers do when they wrap PARSDMM as a projector inside an outer loop
  // (examples/constrained_freq_FWI_simple.jl:468, examples/Constraint_examples_2D.jl:222-223).  Nothing is allocated, no plan,
  // handle, stream or event is created: every array of the context is zero-filled, m is uploaded, rho / gamma return to their
  // initial values, Q is assembled again (the solve updated it incrementally: Q_update!.jl:45-48), every warm start of the
  // searches and of the library-backed projectors is forgotten, the initial feasibility is taken again.  The context then is in
  // the state sipx_finalize leaves, and a solve on it gives the bits a new context gives (tests/test_gpu_round5.py).
  void reset(const void* m, const double* rho_ini, int n_rho, double gamma_ini, int zero_ini_guess, const void* x0,
             const void* const* l0, const void* const* y0, double* feasibility_initial) override {
    need_final();
    if (comm_) throw std::runtime_error("sipx_reset is not available for a rank of a sharded solve (build a new context)");
    SIPX_HIP(hipSetDevice(device_));
    if (lane_thr_.joinable()) lane_thr_.join();
    lane_err_ = nullptr;
    SIPX_HIP(hipStreamSynchronize(stream_));
    for (hipStream_t q : pool_) if (q && q != stream_) SIPX_HIP(hipStreamSynchronize(q));
    if (lane_st_) SIPX_HIP(hipStreamSynchronize(lane_st_));
    if (cstream_) SIPX_HIP(hipStreamSynchronize(cstream_));
    const long long N = G_.N;
    // ---- arrays
    for (int k = 0; k < 3; ++k) dzero(xr_base_[k], stream_);
    dzero(p_base_, stream_); dzero(rhs_, stream_); dzero(r_base_, stream_); dzero(Ap_, stream_);
    dzero(p2_base_, stream_); dzero(w_base_, stream_); dzero(Q2_, stream_);
    dzero(part_cg_, stream_); dzero(part_tmp_, stream_); dzero(part_sets_, stream_); dzero(maxpart_, stream_);
    dzero(cg_dev_, stream_); dzero(sums_ticket_, stream_);
    dzero(fbuf_, stream_); dzero(stage_, stream_); dzero(sstage_, stream_); dzero(gbuf_, stream_);
    dzero(scr_v_, stream_); dzero(scr_c_, stream_); dzero(scr_w_, stream_);
    if (scr_i_) dzero(scr_i_, stream_);
    dzero(lane_v_, stream_);
    SIPX_HIP(hipMemcpyAsync(m_, m, N * sizeof(T), hipMemcpyHostToDevice, stream_));
    x_cur_ = 0; x_snap_ = -1;
    x_ = xr_[0]; xold_ = x_;
    // ---- rho, gamma (PARSDMM_initialize.jl:58-63,107-114,159)
    if (n_rho == 1) std::fill(rho_.begin(), rho_.end(), (T)rho_ini[0]);
    else if (n_rho == p_n_) for (int i = 0; i < p_n_; ++i) rho_[i] = (T)rho_ini[i];
    else throw std::runtime_error("rho_ini must have 1 or p entries");
    T g0 = (T)gamma_ini;
    if (any_ncvx_) g0 = T(0.75);
    std::fill(gamma_.begin(), gamma_.end(), g0);
    // ---- sets
    const bool warm = !zero_ini_guess;
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>& s = sets_[i];
      for (void* b : s.halo_allocs) dzero(b, stream_);
      dzero(s.lh0, stream_); dzero(s.s0, stream_);
      dzero(s.sbuf, stream_);
      dzero(s.ptmp, stream_); dzero(s.mpart, stream_); dzero(s.cbuf, stream_);
      s.snap = -1;
      s.searches_done = 0;
      std::fill(s.sums, s.sums + SLOTS, 0.0);
      s.bb_valid = false;
      s.last_rho = T(-1); s.last_gamma = T(-1);
      if (s.ps) K<T>::ps_init(stream_, s.ps, scr_i_);
      if (s.psf) K<T>::ps_init(stream_, s.psf, scr_i_);
      if (s.ext) { s.ext->set_stream(stream_); s.ext->reset(); }
      if (warm && l0 && l0[i]) upload_rows_ranged(s, (const T*)l0[i], s.l);
      if (warm && y0 && y0[i]) upload_rows_ranged(s, (const T*)y0[i], s.y);
    }
    if (warm && x0) SIPX_HIP(hipMemcpyAsync(x_, x0, Nx_ * sizeof(T), hipMemcpyHostToDevice, stream_));
    // ---- pinned words and the scalars that live beside them
    std::memset(cg_host_, 0, 2 * sizeof(CgState<T>));
    std::memset((void*)ticket_, 0xff, 64);
    std::memset(hres_, 0, sizeof(double) * (p_n_ + 1) * SLOTS);
    std::memset((void*)sums_word_, 0, 64);
    std::memset((void*)hlean_, 0, sizeof(int) * (p_n_ + 1));
    std::memset((void*)hovf_, 0, sizeof(int) * (p_n_ + 1));
    std::memset((void*)hverd_, 0, sizeof(unsigned) * (p_n_ + 1));
    sums_seq_ = 0; cg_seq_ = 0; spec_seq_ = 0;
    spec_searches_ = spec_fallbacks_ = spec_rounds_ = 0;
    batch_searches_ = batch_fallbacks_ = 0;
    head_done_ = false; rs_pending_ = false; sums_pending_ = false; defer_sums_ = false; merge_sums_ = false; merged_nslots_ = 0;
    q_pending_ = false; q_defer_ = false; rhs_fused_ = false; fuse_rhs_ = false; have_log_sums_ = false;
    obj_ss_ = evo_ss_ = xx_ss_ = 0;
    sums_flags_ = 0; word_sums_ = false;
    for (bool& v : open_valid_) v = false;
    mark_weight_[0] = mark_weight_[1] = 1.0;
    nmark_[0] = nmark_[1] = 0;
    mark_step_[0] = mark_step_[1] = 0;
    run_ = Run();
    assemble_Q();
    initial_feasibility(feasibility_initial);
  }

  void initial_feasibility(double* feasibility_initial) {
    // initial feasibility ||P_i(A_i m) - A_i m|| / (||A_i m|| + 100 eps)   (PARSDMM_initialize.jl:97-99)
    feas_init_.assign(pp_n_, 0.0);
    for (int i = 0; i < pp_n_; ++i) {
      SetState<T>& s = sets_[i];
      if (s.dist_ext || s.slab_ext || s.slab_dft) {           // every rank: its slab of slices / its planes of the transform
        dist_feasibility(s, m_, part_sets_ + ((size_t)i * SLOTS + SL_FE2) * NB, i);
        continue;
      }
      if (s.fan) {                              // the set's owner, on the gathered array
        SetArgs<T> a = set_args(s, rho_[i], gamma_[i], 0);
        a.x = m_;
        fan_feasibility(s, a, part_sets_ + ((size_t)i * SLOTS + SL_FE2) * NB);
        continue;
      }
      if (!s.owned) continue;
      if (slab_local_ && !s.two_pass && !s.ext_kind && !s.custom) {      // an element-wise set on the identity: every rank its planes
        SetArgs<T> a = set_args(s, rho_[i], gamma_[i], 0);
        a.x = m_;
        K<T>::proj_dist_set(stream_, Gr_, a, 1, (const ProjScalars<T>*)nullptr, part_sets_ + ((size_t)i * SLOTS + SL_FE2) * NB);
        continue;
      }
      if (slab_ && !s.two_pass && comm_->rank != 0) continue;      // an element-wise set: rank 0 takes the whole grid (one-off)
      double* dst = part_sets_ + ((size_t)i * SLOTS + SL_FE2) * NB;
      // Minkowski: TD_OP[i] * [m; 0] = A m for components 1 and 3, A 0 = 0 for component 2 (w_ is still all zero here)
      const T* mm = s.comp == 2 ? w_ : m_;
      if (s.ext_kind) {
        SetArgs<T> a = set_args(s, rho_[i], gamma_[i], 0);
        a.x = mm;
        ext_feasibility(s, a, dst);
      } else if (s.custom) {                                   // s = A m materialised, then as for an identity set
        K<T>::csr_spmv(stream_, s.Mtrue, s.d_rowptr, s.d_colidx, s.d_rval, mm, s.sbuf);
        if (s.two_pass) {
          SetArgs<T> a = set_args(s, rho_[i], gamma_[i], 0);
          a.x = s.sbuf;
          K<T>::proj_scalars_set(stream_, s.gm, a, 1, s.psf, part_tmp_, maxpart_, scr_c_, s.Mtrue);
          K<T>::proj_dist_set(stream_, s.gm, a, 1, s.psf, dst);
        } else {
          proj_dist_grid<T>(stream_, s.gm, 0, s.dir, s.Mpad, s.sbuf, s.prox, s.plo, s.phi, s.lb, s.ub, nullptr, dst);
        }
      } else if (s.two_pass) {
        SetArgs<T> a = set_args(s, rho_[i], gamma_[i], 0);
        a.x = mm;                                              // s = A m produced on the fly
        SampleCtl cf;
        cf.host_ovf = (int*)hovf_ + i;
        cf.compact_cap = scr_c_len_;
        K<T>::proj_scalars_set(stream_, Gr_, a, 1, s.psf, part_tmp_, maxpart_, scr_c_, s.Mtrue, cf, hooks());
        K<T>::proj_dist_set(stream_, Gr_, a, 1, s.psf, dst);
      } else {
        K<T>::fwd(stream_, G_, s.nblk, s.dir, s.ih, mm, scr_v_);
        proj_dist_grid<T>(stream_, G_, s.nblk, s.dir, s.Mpad, scr_v_, s.prox, s.plo, s.phi, s.lb, s.ub, nullptr, dst);
      }
    }
    reduce_set_sums(p_n_ * SLOTS);
    SIPX_HIP(hipStreamSynchronize(stream_));
    for (int i = 0; i < p_n_ && slab_; ++i)
      if (hovf_[i]) {
        hovf_[i] = 0;
        throw std::runtime_error("initial feasibility of set " + std::to_string(i) + ": the magnitudes inside the final bracket of the l1 search, gathered over all ranks, exceed the exchange segment of " +
                                 std::to_string(hooks_.gcap) + " values per rank (slab decomposition) -- use the set decomposition for this problem");
      }
    for (int i = 0; i < pp_n_; ++i) {
      if (!sets_[i].owned && !comm_) continue;     // sharded: the all-reduced sums of every set are here
      feas_init_[i] = (double)feas_value(hres_[i * SLOTS + SL_FE2], hres_[i * SLOTS + SL_SS2]);
    }
    if (feasibility_initial)
      for (int i = 0; i < pp_n_; ++i) feasibility_initial[i] = feas_init_[i];
  }

  // ------------------------------------------------------------------------------------------
  void rhs_compose(const double* rho) override {
    need_final();
    ObserverGuard og(observer());
    head_done_ = false;           // a residual product queued ahead belonged to the right-hand side that is replaced here
    if (rs_pending_) {          // a right-hand side that was never consumed: let its exchange finish before rhs is rewritten
      SIPX_HIP(hipStreamWaitEvent(stream_, ev_c_[1], 0));
      rs_pending_ = false;
    }
    if (mk_) {    // rhs = sum_i [A 0]'w_i / [0 A]'w_i / [A A]'w_i, w_i = rho_i y_i + l_i, sets added in order per half
      for (int half = 0; half < 2; ++half) {
        T* out = rhs_ + (long long)half * G_.N;
        int launched = 0;
        for (int pass = 0; pass < 2; ++pass) {            // own component first (it precedes the sum sets in TD_OP order)
          RhsArgs<T> a;
          a.nsets = 0;
          for (int i = 0; i < p_n_; ++i) {
            const SetState<T>& s = sets_[i];
            if (s.comp != (pass == 0 ? half + 1 : 3)) continue;
            RhsSet<T>& r = a.s[a.nsets++];
            r.y = s.y; r.l = s.l; r.rho = (T)rho[i]; r.nblk = s.nblk;
            for (int q = 0; q < 3; ++q) { r.dir[q] = s.dir[q]; r.ih[q] = s.ih[q]; }
            if (a.nsets == MAX_SETS) {
              K<T>::rhs_compose(stream_, G_, a, out, launched++ > 0);
              a.nsets = 0;
            }
          }
          if (a.nsets > 0) K<T>::rhs_compose(stream_, G_, a, out, launched++ > 0);
        }
        if (launched == 0) SIPX_HIP(hipMemsetAsync(out, 0, G_.N * sizeof(T), stream_));
      }
      return;
    }
    RhsArgs<T> a;
    a.nsets = 0;
    int launched = 0;
    for (int i = 0; i < p_n_; ++i) {
      const SetState<T>& s = sets_[i];
      if (!s.owned || s.custom) continue;
      RhsSet<T>& r = a.s[a.nsets++];
      r.y = s.y; r.l = s.l; r.rho = (T)rho[i]; r.nblk = s.nblk;
      for (int q = 0; q < 3; ++q) { r.dir[q] = s.dir[q]; r.ih[q] = s.ih[q]; }
      if (a.nsets == MAX_SETS) {
        K<T>::rhs_compose(stream_, Gr_, a, rhs_, launched++ > 0);
        a.nsets = 0;
      }
    }
    if (a.nsets > 0 || launched == 0) K<T>::rhs_compose(stream_, Gr_, a, rhs_, launched > 0);
    for (int i = 0; i < p_n_; ++i) {       // caller-supplied sparse operators: rhs += A_i'(rho_i y_i + l_i), one launch each
      const SetState<T>& s = sets_[i];
      if (s.owned && s.custom)
        K<T>::csc_adj_rhs(stream_, G_.N, s.d_colptr, s.d_rowval, s.d_nzval, s.y, s.l, (T)rho[i], rhs_, 1);
    }
    if (comm_ && !slab_) {
      // the (+) reduction of the partial right-hand sides (rhs_compose.jl:17-20), delivered by z-slab: every rank receives
      // the rows its part of the x-step needs.  It runs on the communication stream; whatever the engine stream is given
      // next (the Q update, the log-only kernels of the previous y/l update) overlaps with it, argmin_x joins.
      SIPX_HIP(hipEventRecord(ev_c_[0], stream_));
      SIPX_HIP(hipStreamWaitEvent(cstream_, ev_c_[0], 0));
      comm_->reduce_scatter_sum(rhs_, (size_t)chunk_, dtype_code(), cstream_);
      SIPX_HIP(hipEventRecord(ev_c_[1], cstream_));
      rs_pending_ = true;
    }
  }

  // First part of the x-step: the residual product r_0 = rhs - Q x (which also keeps x_old and serves the tolerance chain), and,
  // sharded, the grouped call that makes its sums global and hands the boundary planes of p_1 = r_0 to the neighbours.  The
  // whole-solve loop queues it AHEAD -- right behind the y/l update of the previous iteration, before the host waits for that
  // iteration's sums -- whenever the right-hand side is known by then (rho cannot change): the device runs the product while the
  // host reads the sums and evaluates the stop rule (a stop leaves x, y, l untouched: p, x_old and the partials are scratch),
  // and, sharded, the all-reduce of the per-set sums rides in the same call (dres_ sits behind part_cg_).
  void argmin_x_head() {
    if (rs_pending_) {                       // the reduce-scatter of rhs (communication stream) has to have landed
      SIPX_HIP(hipStreamWaitEvent(stream_, ev_c_[1], 0));
      rs_pending_ = false;
    }
    const long long r0 = r0_, r1 = r1_;
    const int dt = dtype_code();
    // the initial residual goes straight into the p buffer (p_1 = r_0, cg.jl:57): the first iteration reads it from there
    // as both r and p and writes r_1 into the r buffer, so the copy p <- r is never made
    bool done = false;
    if (q_pending_ && !stencil_q_ && !comm_ && r0 == 0 && r1 == Nx_) {
      if (!Q2_) throw std::runtime_error("internal: the second copy of Q (SIPX_Q_FUSED) was not allocated at sipx_finalize");
      done = K<T>::resid_qupdate(stream_, G_, Nx_, Q_, Q2_, cds_, q_pending_args_, x_, rhs_, p_, (T*)nullptr, (T*)nullptr, part_cg_);
      if (done) { std::swap(Q_, Q2_); q_pending_ = false; }
    }
    if (!done) {
      flush_q_pending();
      // (x_old is not written: the x-step leaves x_k behind in its own buffer, see the ring of x buffers)
      if (stencil_q_) K<T>::sq_resid(stream_, G_, sq_, x_, rhs_, p_, (T*)nullptr, (T*)nullptr, part_cg_);
      else K<T>::resid(stream_, Nx_, r0, r1, Q_, cds_, x_, rhs_, p_, (T*)nullptr, (T*)nullptr, part_cg_);
    }
    if (comm_) {         // ||r_0||^2, ||rhs||^2 block partials [+ the per-set sums of the y/l update queued just before]; p_1 = r_0 is in p_
      comm_->allreduce_with_halo(part_cg_, (size_t)(2 * NB + merged_nslots_), SIPX_F64, p_ + r0, p_ + r0 - plane_, prev_, p_ + r1 - plane_,
                                 p_ + r1, next_, (size_t)plane_, dt, stream_);
      if (merged_nslots_ > 0) K<T>::copy_f64(stream_, dres_, hres_, merged_nslots_);
      merged_nslots_ = 0;
    }
    head_done_ = true;
  }

  void argmin_x(int it, double* tol_ref_io, int64_t* cg_it, double* cg_relres, int* cg_flag) override {
    need_final();
    ObserverGuard og(observer());
    if (!head_done_) argmin_x_head();
    head_done_ = false;
    const long long r0 = r0_, r1 = r1_, nloc = r1 - r0;      // rows of this rank's part of the x-step (all of them unless sharded)
    const int dt = dtype_code();
    // Sharded (either decomposition): the rows of the x-step are split by z-slab and a product needs the boundary planes of p
    // from the two neighbours.  They are never sent: a rank keeps copies of the neighbours' boundary planes of p AND x and
    // takes them through the same updates (x += alpha p, p = r + beta p: same scalars, same operands, same bits), which only
    // needs the boundary planes of r -- and those do not depend on the all-reduce of ||r||^2 that follows the same kernel, so
    // the two travel in ONE grouped call.  Per CG iteration: all-reduce (p.Ap), then {all-reduce (||r||^2) + planes of r};
    // before the first: {all-reduce (||r_0||^2, ||rhs||^2) + planes of r_0 = p_1} (argmin_x_head); no exchange of x after the solve.
    const long long hlo = (comm_ && prev_ >= 0) ? plane_ : 0, hhi = (comm_ && next_ >= 0) ? plane_ : 0;
    const unsigned seq = ++cg_seq_;
    K<T>::cg_begin(stream_, part_cg_, cg_dev_, cg_host_, it, (T)*tol_ref_io, seq, (unsigned long long*)ticket_);
    // One CG iteration = product + p.Ap -> x, r update + ||r||^2 -> p update.
    // The host enqueues iteration k+1 as soon as the ticket word says that k did not converge, which workgroup 0 of the
    // p-update publishes before it starts streaming: the GPU does not idle on the round trip and nothing is launched
    // for an iteration that does not run.
    // where this x-step writes x: a ring buffer that holds neither the current x nor the snapshot of the x0 mode
    int tgt = 0;
    while (tgt == x_cur_ || tgt == x_snap_) ++tgt;
    T* const xn = xr_[tgt];
    auto enqueue = [&](int k) {
      CgState<T>* mirror = cg_host_ + (k & 1);
      if (stencil_q_) K<T>::sq_spmv_dot(stream_, G_, sq_, p_, Ap_, part_cg_, cg_dev_);
      else K<T>::spmv_dot(stream_, Nx_, r0, r1, Q_, cds_, p_, Ap_, part_cg_, cg_dev_);
      if (comm_) comm_->allreduce_sum(part_cg_, NB, SIPX_F64, stream_);
      K<T>::cg_update_xr(stream_, nloc, (k == 1 ? x_ : xn) + r0, xn + r0, (k == 1 ? p_ : r_) + r0, r_ + r0, p_ + r0, Ap_ + r0, part_cg_, cg_dev_, mirror, k,
                         (unsigned long long*)ticket_, hlo, hhi);
      if (comm_)
        comm_->allreduce_with_halo(part_cg_ + NB, NB, SIPX_F64, r_ + r0, r_ + r0 - plane_, prev_, r_ + r1 - plane_, r_ + r1, next_,
                                   (size_t)plane_, dt, stream_);
      K<T>::cg_update_p(stream_, nloc, p_ + r0, r_ + r0, part_cg_, cg_dev_, mirror, (unsigned long long*)ticket_, hlo, hhi);
    };
    // Fused form (cg_fused_): iteration 1 = product on p_1 (= r_0, in p_), x / r update, then the fused kernel of iteration 2
    // queued at once; iteration k >= 2 = x / r update on the p_k and A p_k the fused kernel left, then the fused kernel of
    // iteration k + 1.  p_k lives in p_ for odd k, in p2_ for even k.
    auto enqueue_fused = [&](int k) {
      CgState<T>* mirror = cg_host_ + (k & 1);
      T* pk = (k & 1) ? p_ : p2_;
      T* pn = (k & 1) ? p2_ : p_;
      if (k == 1) K<T>::spmv_dot(stream_, Nx_, r0, r1, Q_, cds_, p_, Ap_, part_cg_, cg_dev_);
      K<T>::cg_update_xr(stream_, nloc, k == 1 ? x_ : xn, xn, k == 1 ? p_ : r_, r_, pk, Ap_, part_cg_, cg_dev_, mirror, k, (unsigned long long*)ticket_);
      K<T>::spmv_fused(stream_, Nx_, Q_, cds_, r_, pk, pn, Ap_, part_cg_, cg_dev_, mirror, (unsigned long long*)ticket_);
    };
    // (no event is recorded inside this loop: a record costs the stream about 5 us, more than the p-update of a small grid)
    // The first iteration is queued before the verdict of k_cg_begin is back: it is almost always needed, and its kernels
    // return at once on the device-side `done` flag when it is not (zero right-hand side, cg.jl:51; x already solves the
    // system to the tolerance, cg.jl:73-76) -- such a launch is not counted as a sample of the dominant kernel.
    // The first outer iteration of a solve is the one place where that is common (zero start: rhs = 0): there the verdict is
    // awaited first, so that rocprofv3's per-kernel averages hold launches with work only.
    const size_t stat0 = samples_.size();
    const bool ahead = it > 1;
    auto enq = [&](int k) { if (cg_fused_) enqueue_fused(k); else enqueue(k); };
    if (ahead) enq(1);
    bool done = wait_ticket(seq, 0, cg_host_);
    const bool done_at_begin = done;
    CgState<T> fin;
    if (done) {
      fin = cg_host_[0];
      drop_samples_from(stat0);          // the iteration queued ahead returned at once: not a sample of its kernels
      if (fin.flag == -9) SIPX_HIP(hipMemsetAsync(xn + r0 - hlo, 0, (nloc + hlo + hhi) * sizeof(T), stream_));   // cg.jl:51 (the copies of the neighbours' planes too)
    } else {
      const int maxIter = 1000;                       // argmin_x.jl:39
      int iter = 1;
      if (!ahead) enq(1);
      // (Queueing iteration k+1 before the verdict of k -- under a kernel name of its own -- was measured on the small
      // grid where the p-update is shorter than the round trip, 2048^2: 1940 against 2040 it/s.  Not adopted.)
      for (;;) {
        done = wait_ticket(seq, iter, cg_host_ + (iter & 1));
        if (done || iter == maxIter) break;
        enq(++iter);
      }
      fin = cg_host_[iter & 1];         // written before the ticket (release / acquire): complete
    }
    cg_host_[0] = fin;
    // Did the solve write x?  Zero right-hand side (flag -9): x = 0, into the target buffer.  x already good enough (cg.jl:73-76),
    // or the very first step refused (flag -2 at iteration 1, cg.jl:91-93): x is what it was, and x_old names the same buffer.
    const bool wrote = fin.flag == -9 || (!done_at_begin && !(fin.flag == -2 && fin.iters <= 1));
    if (wrote) {
      xold_ = x_;
      x_cur_ = tgt;
      x_ = xn;
    } else {
      xold_ = x_;
    }
    if (comm_) {
      // obj / evol_x sums over the slab (x_old is only kept for the slab), then x is completed on every rank
      // (slab-decomposed with a distance term: the y/l update of that set forms the same three sums over the same rows -- x, m
      //  and x_old of the slab -- and they travel in the same all-reduce; no pass of its own)
      if (!slab_dist_logs_) K<T>::log3(stream_, nloc, x_ + r0, m_ + r0, xold_ + r0, part_sets_ + (size_t)p_n_ * SLOTS * NB);
      // slab-decomposed: the planes of x next to the slab (forward differences read one plane up, the recomputed plane below
      // needs one down) were kept current by the CG updates themselves -- nothing to exchange
      if (!slab_) comm_->allgather(x_, (size_t)chunk_, dt, stream_);
    }
    *tol_ref_io = (double)cg_host_->tol_ref;
    *cg_it = cg_host_->iters;
    *cg_relres = (double)cg_host_->res_last;
    if (cg_flag) *cg_flag = cg_host_->flag;
  }

  void update_y_l(int it, int flags, const double* rho, const double* gamma, double* r_pri, double* r_dual,
                  double* feas) override {
    need_final();
    ObserverGuard og(observer());
    (void)it;
    rhs_fused_ = false;
    MultiArgs<T> ma;
    const bool sweep = sweep_applicable(flags, ma);
    if (x0_mode_ && !sweep) throw std::runtime_error("internal: an x0-mode context met an update the one-sweep kernel does not take");
    // PARTIAL sweep (round 4): set lists with terms the sweep cannot take -- a projector behind a transform or a factorisation,
    // cardinality (BASELINE config 4: l1 behind the DFT, slice rank, cardinality on D_z) -- have their element-wise and l1 / l2 terms
    // updated by the sweep all the same (x read once for them, r_dual in the same pass); the other sets ("loose") keep their
    // per-set kernels below, on the engine stream.  The fused right-hand side then holds the sets in front of the first loose
    // one (MultiBlk::in_rhs; the sets are added in order, rhs_compose.jl:24-31) and k_rhs adds the rest.
    const bool loose_only = sweep && !slab_;
    // (the all-kernel statistics window keeps everything on the engine stream: its event pairs time one kernel at a time)
    const bool lane_now = (loose_only || (slab_ && sweep)) && lane_set_ >= 0 && stats_mode_ != 2;
    struct LaneGuard {                       // whatever ends this call, the lane's host thread is joined first
      Engine<T>* e;
      ~LaneGuard() {
        if (!e->lane_thr_.joinable()) return;              // (lane_join has run: nothing is left over)
        // an exception on the caller's thread between lane_start and lane_join: the lane's work is waited for, its projector
        // goes back to the engine stream, its own error (if any) is dropped in favour of the one that is propagating.  The
        // iterate of the lane set may be half updated: the solve is over, sipx_reset gives the context a defined state again.
        e->lane_thr_.join();
        if (e->lane_st_) (void)hipStreamSynchronize(e->lane_st_);
        try { e->sets_[e->lane_set_].ext->set_stream(e->stream_); } catch (...) {}
        e->lane_err_ = nullptr;
        e->lane_sample_ = -1;
      }
    } lane_guard{this};
    if (lane_now) lane_start(flags, rho, gamma);
    if (loose_only) {
      if (search_batch_) batched_searches(flags, rho, gamma);
      else sweep_searches(flags, rho, gamma);
      sweep_launch(flags, rho, gamma, ma);
      if (!has_loose_) {
        reduce_set_sums(p_n_ * SLOTS);
        sums_flags_ = flags;
        sums_pending_ = true;
        if (!defer_sums_) {
          SIPX_HIP(hipEventRecord(ev_sums_, stream_));
          sums_event_ = ev_sums_;
          collect_set_sums(rho, r_pri, r_dual, feas);
        }
        return;
      }
    }
    if (mk_) K<T>::sum_uv(stream_, G_.N, x_, x_ + G_.N, w_);
    if (set_streams_ && !slab_ && !loose_only) SIPX_HIP(hipEventRecord(ev_fork_, stream_));     // x (and u + v) are final: the sets may start
    if (slab_) {
      // Slab-decomposed iteration: the threshold / scale searches of ALL sets in lock step -- every rank sweeps its planes,
      // ONE all-reduce makes the probe sums of all sets global (twice: first pass, gated refinement), ONE all-gather strings
      // the gathered magnitudes of all l1 sets together; every rank then solves the same small problems (same bits).
      std::vector<int> tp;
      for (int i = 0; i < p_n_; ++i)
        if (sets_[i].two_pass && !sets_[i].fan && !sets_[i].slab_card) tp.push_back(i);
      if (!tp.empty()) {
        const size_t RS = (size_t)(PREP_SLOTS + 1 + 2 * comm_->world);
        const long long seg = hooks_.gcap + GATHER_HDR;
        int nl1 = 0;
        for (int i : tp) nl1 += sets_[i].prox == PX_L1 ? 1 : 0;
        const long long chunk = (long long)std::max(nl1, 1) * seg;
        std::vector<SetArgs<T>> args(tp.size());
        std::vector<T*> gseg(tp.size(), nullptr);
        int k1 = 0;
        const bool batch = spec_exchange_ && spec_batch_ && (int)tp.size() <= SPEC_MAX_SETS;
        RescaleMulti<T> rs;
        rs.n = 0;
        for (size_t j = 0; j < tp.size(); ++j) {
          SetState<T>& s = sets_[tp[j]];
          args[j] = set_args(s, (T)rho[tp[j]], (T)gamma[tp[j]], flags);
          if (s.prox == PX_L1) gseg[j] = gbuf_ + (long long)(k1++) * seg;
          if (s.prox == PX_L1 && s.last_rho > T(0) && s.last_rho != args[j].rho) {    // v rescaled: theta moves like 1/rho
            if (batch && rs.n < SPEC_MAX_SETS) { rs.ps[rs.n] = s.ps; rs.factor[rs.n++] = (double)s.last_rho / (double)args[j].rho; }
            else K<T>::ps_rescale(stream_, s.ps, (double)s.last_rho / (double)args[j].rho);
          }
        }
        K<T>::ps_rescale_multi(stream_, rs);          // (one launch for all of them)
        // sampled prediction of theta (kernels_proj.hip, k_sample) for the l1 sets whose last search asked for it: every rank
        // samples its planes, ONE all-reduce adds the histograms (float64 holding exact integers), every rank decides alike
        std::vector<SampleCtl> ctl(tp.size());
        bool any_sample = false;
        const size_t SS = (size_t)(2 * SAMPLE_BINS + 3);
        for (size_t j = 0; j < tp.size(); ++j) {
          SetState<T>& s = sets_[tp[j]];
          ctl[j].host_want = (int*)hlean_ + tp[j];
          ctl[j].host_ovf = (int*)hovf_ + tp[j];
          ctl[j].compact_cap = sets_[tp[j]].cbuf_len;
          ctl[j].runs = l1_sample_runs_;
          const bool rescaled = s.prox == PX_L1 && s.last_rho > T(0) && s.last_rho != args[j].rho;
          ctl[j].enable = l1_sample_ && s.prox == PX_L1 && (rescaled || (hlean_[tp[j]] & 0xff) != 0);
          any_sample |= ctl[j].enable != 0;
        }
        if (any_sample && batch && Gr_.n[0] % 4 == 0) {      // every sampling set in one launch per stage
          SampleMulti<T> sm;
          sm.ns = 0;
          for (size_t j = 0; j < tp.size(); ++j) {
            if (!ctl[j].enable || sets_[tp[j]].prox != PX_L1) continue;
            SampleSet<T>& S = sm.s[sm.ns++];
            S.a = args[j]; S.a.ps = sets_[tp[j]].ps; S.ps = sets_[tp[j]].ps; S.partials = sets_[tp[j]].ptmp;
            S.reg = sstage_ + j * SS; S.true_len = sets_[tp[j]].Mtrue;
          }
          K<T>::sample_multi(10, stream_, Gr_, sm, l1_sample_runs_, &hooks_);
          comm_->allreduce_sum(sstage_, tp.size() * SS, SIPX_F64, stream_);
          K<T>::sample_multi(11, stream_, Gr_, sm, l1_sample_runs_, &hooks_);
        } else if (any_sample) {
          for (int stage = 10; stage <= 11; ++stage) {
            for (size_t j = 0; j < tp.size(); ++j)
              if (ctl[j].enable)
                K<T>::proj_scalars_stage(stage, stream_, Gr_, args[j], 0, sets_[tp[j]].ps, sets_[tp[j]].ptmp, sets_[tp[j]].mpart,
                                         sets_[tp[j]].cbuf, sets_[tp[j]].Mtrue, ctl[j], &hooks_, sstage_ + j * SS, gseg[j], chunk);
            if (stage == 10) comm_->allreduce_sum(sstage_, tp.size() * SS, SIPX_F64, stream_);
          }
        }
        for (size_t j = 0; j < tp.size(); ++j) ctl[j].enable = 0;
        // Refinement rounds (a gated probe pass + one all-reduce each): the bracket has to shrink until what it holds, over
        // all ranks, fits the exchange segments -- a rank cannot keep what does not fit.  How many rounds are enqueued follows
        // the previous search of each l1 set (pinned word written by k_l1_solve; the same on every rank): two more than it
        // used, all of them while nothing is known (the first searches of a solve); a search that needs more ends in the
        // error return of collect_set_sums, never in a wrong theta.
        int rounds = 1;
        for (size_t j = 0; j < tp.size(); ++j) {
          SetState<T>& s = sets_[tp[j]];
          if (s.prox != PX_L1) continue;
          const int used = (hlean_[tp[j]] >> 8) & 0xff;
          const int want = s.searches_done < 2 ? 6 : std::min(6, std::max(2, used + 2));
          rounds = std::max(rounds, want);
          s.searches_done += 1;
        }
        if (const char* e = std::getenv("SIPX_L1_ROUNDS_MIN")) rounds = std::min(6, std::max(rounds, std::atoi(e)));      // a problem whose brackets shrink slowly
        if (const char* e = std::getenv("SIPX_L1_ROUNDS_MAX")) rounds = std::max(1, std::min(rounds, std::atoi(e)));      // tests: force an overflow
        if (spec_exchange_) {
          spec_exchange_searches(tp, args, ctl, gseg, chunk, 0, false, it, rs.n == 0 && !any_sample);
        } else {
        const int order[4] = {0, 1, 2, 3};
        for (int si = 0; si < 4; ++si) {
          const int stage = order[si];
          const int reps = stage == 1 ? rounds : 1;
          for (int rep = 0; rep < reps; ++rep) {
            const int st = (stage == 1 && rep > 0) ? 4 : stage;
            for (size_t j = 0; j < tp.size(); ++j) {
              SetState<T>& s = sets_[tp[j]];
              K<T>::proj_scalars_stage(st, stream_, Gr_, args[j], 0, s.ps, s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], &hooks_,
                                       stage_ + j * RS, gseg[j], chunk);
            }
            if (stage < 2 && (stage == 0 || nl1 > 0)) comm_->allreduce_sum(stage_, tp.size() * RS, SIPX_F64, stream_);
          }
          if (stage == 2 && nl1 > 0) comm_->allgather(gbuf_, (size_t)chunk, dtype_code(), stream_);
        }
        }
      }
    }
    if (slab_ && sweep) {
      // slab-decomposed: the searches ran in lock step above; every set's update in one sweep over the rank's planes (plus
      // the last plane of the rank below, recomputed), then the feasibility searches of the two-pass sets with their
      // collectives, on the engine stream, in one order on every rank
      for (int i = 0; i < p_n_; ++i)
        if (sets_[i].two_pass) { sets_[i].last_rho = (T)rho[i]; sets_[i].last_gamma = (T)gamma[i]; }
      sweep_launch(flags, rho, gamma, ma);
      if ((flags & SIPX_YL_FEAS) && spec_exchange_) {
        // the feasibility estimates ||P_i(A_i x) - A_i x|| of the two-pass sets: their searches (on v = A_i x itself, each set's
        // second scalar state) in lock step through the same exchange -- one all-gather for all of them -- then the distances
        std::vector<int> tf;
        for (int i = 0; i < pp_n_; ++i)
          if (sets_[i].two_pass && !sets_[i].fan && !sets_[i].slab_card) tf.push_back(i);
        if (!tf.empty()) {
          const long long seg = hooks_.gcap + GATHER_HDR;
          int nl1 = 0;
          for (int i = 0; i < p_n_; ++i) nl1 += (sets_[i].two_pass && sets_[i].prox == PX_L1) ? 1 : 0;
          const long long chunk = (long long)std::max(nl1, 1) * seg;
          std::vector<SetArgs<T>> fa(tf.size());
          std::vector<SampleCtl> fc(tf.size());
          std::vector<T*> fg(tf.size(), nullptr);
          int k1 = 0;
          for (size_t j = 0; j < tf.size(); ++j) {
            fa[j] = set_args(sets_[tf[j]], (T)rho[tf[j]], (T)gamma[tf[j]], flags);
            fc[j].host_ovf = (int*)hovf_ + tf[j];
            fc[j].compact_cap = sets_[tf[j]].cbuf_len;
            if (sets_[tf[j]].prox == PX_L1) fg[j] = gbuf_ + (long long)(k1++) * seg;
          }
          spec_exchange_searches(tf, fa, fc, fg, chunk, 1, true, it);
          for (size_t j = 0; j < tf.size(); ++j)
            K<T>::proj_dist_set(stream_, Gr_, fa[j], 1, sets_[tf[j]].psf, part_sets_ + ((size_t)tf[j] * SLOTS + SL_FE2) * NB);
        }
      } else if (flags & SIPX_YL_FEAS) {
        for (int i = 0; i < pp_n_; ++i) {
          SetState<T>& s = sets_[i];
          if (!s.two_pass || s.fan || s.slab_card) continue;
          SetArgs<T> a = set_args(s, (T)rho[i], (T)gamma[i], flags);
          SampleCtl cf;
          cf.host_ovf = (int*)hovf_ + i;
          cf.compact_cap = s.cbuf ? s.cbuf_len : scr_c_len_;
          K<T>::proj_scalars_set(stream_, Gr_, a, 1, s.psf, s.ptmp ? s.ptmp : part_tmp_, s.mpart ? s.mpart : maxpart_, s.cbuf ? s.cbuf : scr_c_,
                                 s.Mtrue, cf, hooks());
          K<T>::proj_dist_set(stream_, Gr_, a, 1, s.psf, part_sets_ + ((size_t)i * SLOTS + SL_FE2) * NB);
        }
      }
    }
    if (set_streams_ && slab_ && !sweep) SIPX_HIP(hipEventRecord(ev_fork_, stream_));      // the searches are done: the updates may start
    for (int i = 0; i < p_n_ && slab_loose_; ++i)          // the gathered sets first: their owners project beside what follows
      if (sets_[i].fan) fan_begin(sets_[i], set_args(sets_[i], (T)rho[i], (T)gamma[i], flags));
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>& s = sets_[i];
      if (!s.owned || s.dist_ext) continue;
      if ((loose_only || (slab_ && sweep)) && s.in_sweep) continue;             // (updated by the sweep above)
      if (lane_now && i == lane_set_) continue;           // (on its lane, lane_start)
      SetArgs<T> a = set_args(s, (T)rho[i], (T)gamma[i], flags);
      if (mk_) a.x = s.comp == 1 ? x_ : (s.comp == 2 ? x_ + G_.N : w_);
      double* part = part_sets_ + (size_t)i * SLOTS * NB;
      hipStream_t q = (s.st && !loose_only) ? s.st : stream_;
      // slab-decomposed: a feasibility search carries collectives -- those stay on the engine stream, in one order on every rank
      if (slab_ && (flags & SIPX_YL_FEAS) && s.two_pass && i < pp_n_) q = stream_;
      double* ptmp = s.ptmp ? s.ptmp : part_tmp_;
      T* mpart = s.mpart ? s.mpart : maxpart_;
      T* cbuf = s.cbuf ? s.cbuf : scr_c_;
      if (q != stream_) SIPX_HIP(hipStreamWaitEvent(q, ev_fork_, 0));
      // where y_new, l_new go (see SetState): snapshot iterations overwrite the old snapshot, the others stay off it
      const bool snapshot = (flags & (SIPX_YL_BB | SIPX_YL_FIRST)) != 0;
      const bool first = (flags & SIPX_YL_FIRST) != 0;
      bool to_other;
      if (snapshot) to_other = !first && s.snap != 0;     // first: in place, (y, l) becomes the snapshot; no snapshot yet:
      else to_other = s.snap == 0;                        // the zero-filled other pair stands in for it, as before
      a.yo = to_other ? s.y0 : s.y;
      a.lo = to_other ? s.l0 : s.l;
      const Grid& gs = s.custom ? s.gm : Gr_;          // (the rank's slab when the whole iteration is slab-decomposed)
      const Grid& gy = s.custom ? s.gm : Gyl_;
      if (s.custom) {     // s = A x once, then the identity-shaped kernels on the M entries of s
        K<T>::csr_spmv(q, s.Mtrue, s.d_rowptr, s.d_colidx, s.d_rval, a.x, s.sbuf);
        a.x = s.sbuf;
        a.flags |= F_STORE_DY;
      }
      if (s.slab_ext) {   // slab-decomposed, slices of the rank's own planes: nothing crosses the fabric
        q = stream_;
        K<T>::store_v(q, Gr_, a, 0, loose_v_);
        if (r1_ > r0_) {
          ObsScope obs(KID_EXT, stream_, 0.0);
          s.ext->project(loose_v_ + r0_, false, ptmp, mpart, cbuf);
        }
        a.v = loose_v_;
        a.vsrc = 2;
      } else if (s.slab_dft) {   // slab-decomposed transform: every rank its planes, one all-to-all each way, the threshold over all ranks
        q = stream_;
        K<T>::store_v(q, Gr_, a, 0, loose_v_);
        {
          ObsScope obs(KID_EXT, stream_, 0.0);
          s.ddft->project(loose_v_ + r0_, false, comm_.get(), &hooks_, part_tmp_, maxpart_, scr_c_, scr_c_len_, (int*)hovf_ + i);
        }
        a.v = loose_v_;
        a.vsrc = 2;
      } else if (s.fan) { // slab-decomposed, a projector that needs the whole array: its owner has gathered v (fan_begin, in front of
        q = stream_;      // this loop) and scatters P(v)
        fan_return(s);
        a.v = s.fanv;
        a.vsrc = 2;
      } else if (s.ext_kind) {   // library-backed projector: materialise v, project it in place, hand y to the fused update
        K<T>::store_v(q, G_, a, 0, scr_v_);
        {
          ObsScope obs(KID_EXT, stream_, 0.0);
          s.ext->project(scr_v_, false, ptmp, mpart, cbuf);
        }
        a.vsrc = 2;
      }
      if (s.fan) {                        // (projected above)
        s.last_rho = a.rho;
        s.last_gamma = a.gamma;
      } else if (s.two_pass && slab_ && !s.slab_card) {          // (searched above, in lock step with the other sets)
        s.last_rho = a.rho;
        s.last_gamma = a.gamma;
      } else if (s.two_pass) {   // threshold / scale of prox_i from one pass that produces v on the fly (nothing stored)
        SetArgs<T> ap = a;
        const bool rescaled = a.prox == PX_L1 && s.last_rho > T(0) && s.last_rho != a.rho;
        if (rescaled)                                                               // v rescaled: theta moves like 1/rho
          K<T>::ps_rescale(q, s.ps, (double)s.last_rho / (double)a.rho);
        // Sampled prediction of theta in front of the search (kernels_proj.hip, k_sample) when the previous search of this
        // set asked for it (theta moved, fallback sweeps were needed) or rho was changed: its verdict sits in pinned memory, written by k_l1_solve before the sums of that
        // iteration reached the host.  (A stale word costs time only: the kernels check the device-side state themselves.)
        SampleCtl ctl;
        ctl.host_want = (int*)hlean_ + i;
        ctl.host_ovf = (int*)hovf_ + i;
        ctl.runs = l1_sample_runs_;
        ctl.enable = l1_sample_ && !slab_ && a.prox == PX_L1 && !s.custom && (rescaled || (hlean_[i] & 0xff) != 0);
        K<T>::proj_scalars_set(q, gs, ap, 0, s.ps, ptmp, mpart, cbuf, s.Mtrue, ctl, hooks());
        s.last_rho = a.rho;
        s.last_gamma = a.gamma;
      }
      K<T>::yl(q, (s.slab_ext || s.slab_dft || (s.fan && s.ident)) ? gs : gy, a, part);      // (no adjoint stencil reads the plane below: nothing to recompute)
      if (s.custom) K<T>::csc_adj_norm(q, G_.N, s.d_colptr, s.d_rowval, s.d_nzval, s.dy, part + (size_t)SL_ADJ * NB);
      else if (!s.ident) K<T>::adj_norm(q, gs, a, part + (size_t)SL_ADJ * NB);
      if ((flags & SIPX_YL_FEAS) && (s.slab_ext || s.slab_dft) && i < pp_n_) dist_feasibility(s, x_, part + (size_t)SL_FE2 * NB, i);
      else if ((flags & SIPX_YL_FEAS) && s.fan && i < pp_n_) fan_feasibility(s, a, part + (size_t)SL_FE2 * NB);
      else if ((flags & SIPX_YL_FEAS) && s.ext_kind && i < pp_n_) ext_feasibility(s, a, part + (size_t)SL_FE2 * NB);
      if ((flags & SIPX_YL_FEAS) && s.two_pass && !s.fan && i < pp_n_) {
        // ||P_i(s) - s|| with s = A_i x produced on the fly; its own warm-started scalars (psf)
        SampleCtl cf;
        cf.host_ovf = (int*)hovf_ + i;
        cf.compact_cap = s.cbuf ? s.cbuf_len : scr_c_len_;
        K<T>::proj_scalars_set(q, gs, a, 1, s.psf, ptmp, mpart, cbuf, s.Mtrue, cf, hooks());
        K<T>::proj_dist_set(q, gs, a, 1, s.psf, part + (size_t)SL_FE2 * NB);
      }
      if (to_other) { std::swap(s.y, s.y0); std::swap(s.l, s.l0); }     // (y, l) always names the current iterate
      if (snapshot) s.snap = 0;
      else if (to_other && s.snap == 0) s.snap = 1;
    }
    // Sets whose projector is shared by all ranks (slice-wise rank / nuclear norm), in set order on every rank, after the
    // rank's own sets are queued: the owner materialises v = x_hat - l / rho and scatters it by slab (seven links at once:
    // 7/8 of N w bytes leave it, a broadcast would move seven times that), every rank projects the slices of its slab, a
    // gather returns P(v), the owner finishes the update with it.  (Engine stream: the
    // collectives are ordered against the rank's other work; what runs on the second set stream overlaps.)
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>& s = sets_[i];
      if (!s.dist_ext) continue;
      const int dt = dtype_code();
      double* part = part_sets_ + (size_t)i * SLOTS * NB;
      SetArgs<T> a;
      bool to_other = false;
      const bool snapshot = (flags & (SIPX_YL_BB | SIPX_YL_FIRST)) != 0;
      if (s.owned) {
        a = set_args(s, (T)rho[i], (T)gamma[i], flags);
        const bool first = (flags & SIPX_YL_FIRST) != 0;
        if (snapshot) to_other = !first && s.snap != 0;
        else to_other = s.snap == 0;
        a.yo = to_other ? s.y0 : s.y;
        a.lo = to_other ? s.l0 : s.l;
        K<T>::store_v(stream_, G_, a, 0, scr_v_);
      }
      comm_->scatter(scr_v_, (size_t)chunk_, dt, s.owner_rank, stream_);
      if (s.ext) {
        ObsScope obs(KID_EXT, stream_, 0.0);
        s.ext->project(scr_v_ + r0_, false, part_tmp_, maxpart_, scr_c_);
      }
      comm_->gather(scr_v_, (size_t)chunk_, dt, s.owner_rank, stream_);
      if (s.owned) {
        a.vsrc = 2;
        K<T>::yl(stream_, G_, a, part);
        if (to_other) { std::swap(s.y, s.y0); std::swap(s.l, s.l0); }
        if (snapshot) s.snap = 0;
        else if (to_other && s.snap == 0) s.snap = 1;
      }
      if ((flags & SIPX_YL_FEAS) && i < pp_n_) dist_feasibility(s, x_, part + (size_t)SL_FE2 * NB);
    }
    if (lane_now) lane_join(flags);
    if ((loose_only || (slab_ && sweep)) && rhs_fused_) {
      // the sweep wrote the sum over the sets in front of the first loose one; the rest, in order
      RhsArgs<T> ra;
      ra.nsets = 0;
      bool behind = false;
      for (int i = 0; i < p_n_; ++i) {
        const SetState<T>& s = sets_[i];
        behind |= !s.in_sweep;
        if (!behind) continue;
        RhsSet<T>& r = ra.s[ra.nsets++];
        r.y = s.y; r.l = s.l; r.rho = (T)rho[i]; r.nblk = s.nblk;
        for (int qd = 0; qd < 3; ++qd) { r.dir[qd] = s.dir[qd]; r.ih[qd] = s.ih[qd]; }
        if (ra.nsets == MAX_SETS) {
          K<T>::rhs_compose(stream_, Gr_, ra, rhs_, 1);
          ra.nsets = 0;
        }
      }
      if (ra.nsets > 0) K<T>::rhs_compose(stream_, Gr_, ra, rhs_, 1);
    }
    for (size_t k = 0; k < pool_.size() && !(slab_ && sweep) && !loose_only; ++k) {   // join: the reductions below see every set
      if (pool_[k] == stream_) continue;
      SetState<T>* last = nullptr;                                      // (one event per set stream: after its last set)
      for (int i = 0; i < p_n_; ++i)
        if (sets_[i].owned && sets_[i].st == pool_[k]) last = &sets_[i];
      if (!last) continue;
      SIPX_HIP(hipEventRecord(last->ev, last->st));
      SIPX_HIP(hipStreamWaitEvent(stream_, last->ev, 0));
    }
    // Minkowski: evol_x runs over all 2N unknowns (PARSDMM.jl:145); the distance-term kernel only saw u + v
    if (mk_) K<T>::log3(stream_, Nx_, x_, (const T*)nullptr, xold_, part_sets_ + (size_t)p_n_ * SLOTS * NB);
    reduce_set_sums((p_n_ + ((mk_ || comm_) ? 1 : 0)) * SLOTS);
    sums_flags_ = flags;
    sums_pending_ = true;
    if (!defer_sums_) {
      SIPX_HIP(hipEventRecord(ev_sums_, stream_));
      sums_event_ = ev_sums_;
      collect_set_sums(rho, r_pri, r_dual, feas);
    }
  }

  // The lane set's y/l update (see sipx_finalize): x is final on the engine stream; store v, project it, update y and l on the
  // lane stream, queued by a host thread (the projector waits for its own stream between its steps).  The pointer rotation of
  // the set happens here, on the caller's thread; the thread only launches.
  void lane_start(int flags, const double* rho, const double* gamma) {
    SetState<T>& s = sets_[lane_set_];
    SetArgs<T> a = set_args(s, (T)rho[lane_set_], (T)gamma[lane_set_], flags);
    const bool snapshot = (flags & (SIPX_YL_BB | SIPX_YL_FIRST)) != 0;
    const bool first = (flags & SIPX_YL_FIRST) != 0;
    const bool to_other = snapshot ? (!first && s.snap != 0) : (s.snap == 0);
    a.yo = to_other ? s.y0 : s.y;
    a.lo = to_other ? s.l0 : s.l;
    a.v = lane_v_;
    lane_args_ = a;
    SIPX_HIP(hipEventRecord(lane_fork_, stream_));
    SIPX_HIP(hipStreamWaitEvent(lane_st_, lane_fork_, 0));
    s.ext->set_stream(lane_st_);
    lane_err_ = nullptr;
    // the lane's kernels are launched by a thread without a launch observer (the sample list is not shared between threads): the
    // whole update of the set is booked from HERE as one inclusive interval on the lane stream, closed in lane_join -- in the
    // all-kernel statistics the slice-rank set of C4 used to be missing altogether
    lane_sample_ = -1;
    if (stats_mode_ >= 2) {
      const size_t i = samples_.size();
      samples_.push_back(KSample{KID_EXT, 0.0, 0.0});
      SIPX_HIP(hipEventRecord(stat_event(2 * i), lane_st_));
      lane_sample_ = (long long)i;
    }
    double* part = part_sets_ + (size_t)lane_set_ * SLOTS * NB;
    double* ptmp = s.ptmp ? s.ptmp : part_tmp_;
    T* mpart = s.mpart ? s.mpart : maxpart_;
    T* cbuf = s.cbuf ? s.cbuf : scr_c_;
    ExtProj<T>* ext = s.ext.get();
    lane_thr_ = std::thread([this, a, part, ptmp, mpart, cbuf, ext]() {
      try {
        SIPX_HIP(hipSetDevice(device_));
        K<T>::store_v(lane_st_, slab_ ? Gr_ : G_, a, 0, lane_v_);
        if (!slab_) ext->project(lane_v_, false, ptmp, mpart, cbuf);
        else if (r1_ > r0_) ext->project(lane_v_ + r0_, false, ptmp, mpart, cbuf);      // (the slices of the rank's own planes)
        SetArgs<T> a2 = a;
        a2.vsrc = 2;
        K<T>::yl(lane_st_, slab_ ? Gr_ : Gyl_, a2, part);
        SIPX_HIP(hipEventRecord(lane_ev_, lane_st_));
      } catch (...) {
        lane_err_ = std::current_exception();
      }
    });
    if (to_other) { std::swap(s.y, s.y0); std::swap(s.l, s.l0); }     // (y, l) always names the current iterate
    if (snapshot) s.snap = 0;
    else if (to_other && s.snap == 0) s.snap = 1;
  }
  void lane_join(int flags) {
    SetState<T>& s = sets_[lane_set_];
    if (lane_thr_.joinable()) lane_thr_.join();
    s.ext->set_stream(stream_);
    if (lane_sample_ >= 0 && (size_t)lane_sample_ < samples_.size())
      SIPX_HIP(hipEventRecord(stat_event(2 * (size_t)lane_sample_ + 1), lane_st_));
    lane_sample_ = -1;
    if (lane_err_) {
      (void)hipStreamSynchronize(lane_st_);
      std::exception_ptr e = lane_err_;
      lane_err_ = nullptr;
      std::rethrow_exception(e);
    }
    SIPX_HIP(hipStreamWaitEvent(stream_, lane_ev_, 0));
    if ((flags & SIPX_YL_FEAS) && lane_set_ < pp_n_) {
      if (slab_) dist_feasibility(s, x_, part_sets_ + ((size_t)lane_set_ * SLOTS + SL_FE2) * NB, lane_set_);
      else {
        SetArgs<T> a = lane_args_;
        a.v = scr_v_;
        ext_feasibility(s, a, part_sets_ + ((size_t)lane_set_ * SLOTS + SL_FE2) * NB);
      }
    }
  }


  // ---- communicator self-test (round 5) -------------------------------------------------------------------------
  // RcclComm has only ever run with a world of one here (a gpurun box has one GPU), and the slab decomposition hands RCCL halo
  // planes that live in hipMemMap-backed memory.  Before any array of the context is allocated every rank therefore runs the
  // communicator's operations once on KNOWN data -- the grouped all-reduce + neighbour exchange, the in-place reduce-scatter and
  // all-gather (the offsets RcclComm computes), the fan scatter / gather, the all-to-all, all on plain memory; then the neighbour exchange once
  // more with its four buffers inside a mapped granule between two unmapped ones -- compares what arrived with what must have
  // arrived, and the ranks agree on the outcome through one more all-reduce on plain memory.  Wrong data from a base operation
  // is an error of sipx_finalize on every rank alike (the caller may attach another communicator: bench.py goes on with
  // torch.distributed callbacks); a failure of the mapped exchange alone switches THIS context to full-size arrays on every
  // rank.  SIPX_COMM_SELFTEST=0 skips it.  The same code runs through the callback communicator (tests: 2-4 ranks on one GPU).
  void comm_self_test(bool want_mapped, bool want_a2a = false) {
    const char* e = std::getenv("SIPX_COMM_SELFTEST");
    if (e && e[0] == '0') {
      selftest_ = "skipped (SIPX_COMM_SELFTEST=0)";
      if (!agree_buf_) agree_buf_ = dalloc<double>(64);
      return;
    }
    Comm& c = *comm_;
    const int W = c.world, R = c.rank, dt = dtype_code();
    const size_t chunk = 1024, hc = 256;
    const int prev = R > 0 ? R - 1 : -1, next = R + 1 < W ? R + 1 : -1;
    auto val = [](int r, size_t e) { return (double)((r + 1) * 256 + (int)(e & 255)); };
    std::vector<T> h(std::max<size_t>(W * chunk, 4 * hc));
    std::vector<double> hr(64);
    T* buf = dalloc<T>(W * chunk);
    T* hal = dalloc<T>(4 * hc);
    double* red = dalloc<double>(64);
    T* vm_base = nullptr;
    std::string why;
    auto put = [&](T* dst, size_t n) { SIPX_HIP(hipMemcpy(dst, h.data(), n * sizeof(T), hipMemcpyHostToDevice)); };
    auto get = [&](const T* src, size_t n) {
      SIPX_HIP(hipStreamSynchronize(stream_));
      SIPX_HIP(hipMemcpy(h.data(), src, n * sizeof(T), hipMemcpyDeviceToHost));
    };
    auto fill_halo = [&](T* base) {            // [send_prev | send_next | recv_prev | recv_next], hc entries each
      for (size_t j = 0; j < hc; ++j) { h[j] = (T)((R + 1) * 4096 + (int)j); h[hc + j] = (T)((R + 1) * 4096 + 2048 + (int)j); h[2 * hc + j] = h[3 * hc + j] = T(-1); }
      put(base, 4 * hc);
    };
    auto check_halo = [&](const T* base, const char* what) {
      get(base, 4 * hc);
      for (size_t j = 0; j < hc; ++j) {
        if (prev >= 0 && h[2 * hc + j] != (T)(R * 4096 + 2048 + (int)j)) { why = std::string(what) + ": the plane received from the rank below is not what it sent"; return false; }
        if (next >= 0 && h[3 * hc + j] != (T)((R + 2) * 4096 + (int)j)) { why = std::string(what) + ": the plane received from the rank above is not what it sent"; return false; }
        if (prev < 0 && h[2 * hc + j] != T(-1)) { why = std::string(what) + ": a receive buffer without a neighbour was written"; return false; }
      }
      return true;
    };
    auto red_fill = [&]() {
      for (int i = 0; i < 64; ++i) hr[i] = (double)((R + 1) * 1000 + i);
      SIPX_HIP(hipMemcpy(red, hr.data(), 64 * sizeof(double), hipMemcpyHostToDevice));
    };
    auto red_check = [&](const char* what) {
      SIPX_HIP(hipStreamSynchronize(stream_));
      SIPX_HIP(hipMemcpy(hr.data(), red, 64 * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = 0; i < 64; ++i)
        if (hr[i] != 1000.0 * W * (W + 1) / 2 + (double)W * i) { why = std::string(what) + ": the all-reduced sums are wrong"; return false; }
      return true;
    };
    bool ok = true, mapped_ok = true, a2a_ok = true;
    try {
      // A: grouped all-reduce + neighbour exchange, plain memory
      red_fill();
      fill_halo(hal);
      c.allreduce_with_halo(red, 64, SIPX_F64, hal, hal + 2 * hc, prev, hal + hc, hal + 3 * hc, next, hc, dt, stream_);
      ok = red_check("all-reduce + neighbour exchange") && check_halo(hal, "all-reduce + neighbour exchange");
      // B: reduce-scatter in place
      if (ok) {
        for (size_t q = 0; q < W * chunk; ++q) h[q] = (T)val(R, q);
        put(buf, W * chunk);
        c.reduce_scatter_sum(buf, chunk, dt, stream_);
        get(buf, W * chunk);
        for (size_t q = R * chunk; q < (R + 1) * chunk && ok; ++q)
          if (h[q] != (T)(256.0 * W * (W + 1) / 2 + (double)W * (double)(q & 255))) { ok = false; why = "reduce-scatter (in place): the rank's range does not hold the sums"; }
      }
      // C: all-gather in place
      if (ok) {
        for (size_t q = 0; q < W * chunk; ++q) h[q] = (q / chunk == (size_t)R) ? (T)val(R, q) : T(-3);
        put(buf, W * chunk);
        c.allgather(buf, chunk, dt, stream_);
        get(buf, W * chunk);
        for (size_t q = 0; q < W * chunk && ok; ++q)
          if (h[q] != (T)val((int)(q / chunk), q)) { ok = false; why = "all-gather (in place): a range does not hold its rank's values"; }
      }
      // D: fan scatter from rank 0, fan gather to the last rank
      if (ok) {
        for (size_t q = 0; q < W * chunk; ++q) h[q] = R == 0 ? (T)(val((int)(q / chunk), q) + 7.0) : T(-5);
        put(buf, W * chunk);
        c.scatter(buf, chunk, dt, 0, stream_);
        get(buf, W * chunk);
        for (size_t q = R * chunk; q < (R + 1) * chunk && ok; ++q)
          if (h[q] != (T)(val(R, q) + 7.0)) { ok = false; why = "scatter: the rank's range is not what the root sent"; }
      }
      if (ok) {
        for (size_t q = 0; q < W * chunk; ++q) h[q] = (q / chunk == (size_t)R) ? (T)(val(R, q) + 11.0) : T(-7);
        put(buf, W * chunk);
        c.gather(buf, chunk, dt, W - 1, stream_);
        get(buf, W * chunk);
        if (R == W - 1)
          for (size_t q = 0; q < W * chunk && ok; ++q)
            if (h[q] != (T)(val((int)(q / chunk), q) + 11.0)) { ok = false; why = "gather: a range at the root is not what its rank sent"; }
      }
    } catch (const std::exception& ex) {
      ok = false;
      why = std::string("an operation failed: ") + ex.what();
    }
    // D2 (only where a set needs it: the slab-decomposed DFT): all-to-all -- range d of what rank r sends carries (r, d).  A failure
    // here does not fail sipx_finalize: the set goes through an owner rank instead (two fan exchanges), on every rank alike.
    if (ok && want_a2a) {
      T* a2a = nullptr;
      try {
        a2a = dalloc<T>(3 * W * chunk);
        for (size_t q = 0; q < W * chunk; ++q) h[q] = (T)((R + 1) * 64 + (int)(q / chunk) + (double)(q & 15) / 16.0);
        put(a2a, W * chunk);
        c.alltoall(a2a, a2a + W * chunk, a2a + 2 * W * chunk, chunk, dt, stream_);
        get(a2a + W * chunk, W * chunk);
        for (size_t q = 0; q < W * chunk && a2a_ok; ++q)
          if (h[q] != (T)(((int)(q / chunk) + 1) * 64 + R + (double)(q & 15) / 16.0)) { a2a_ok = false; a2a_why_ = "a range is not what its rank sent to this one"; }
      } catch (const std::exception& ex) {
        a2a_ok = false;
        a2a_why_ = ex.what();
      }
      dfree(a2a);
    }
    // E: the neighbour exchange out of / into hipMemMap-backed memory (one mapped granule between two that are not)
    if (ok && want_mapped) {
      try {
        vm_base = (T*)sparse_alloc_bytes(3 * SPARSE_GRAN, {{SPARSE_GRAN, 2 * SPARSE_GRAN}}, device_);
        T* vm = (T*)((char*)vm_base + SPARSE_GRAN) + 64;
        red_fill();
        fill_halo(vm);
        c.allreduce_with_halo(red, 64, SIPX_F64, vm, vm + 2 * hc, prev, vm + hc, vm + 3 * hc, next, hc, dt, stream_);
        std::string keep = why;
        mapped_ok = red_check("mapped memory") && check_halo(vm, "mapped memory");
        if (!mapped_ok) mapped_why_ = why;
        why = keep;
      } catch (const std::exception& ex) {
        mapped_ok = false;
        mapped_why_ = ex.what();
      }
    }
    if (const char* f = std::getenv("SIPX_COMM_SELFTEST_FAIL")) {      // tests: "mapped" / "base", optionally ":rank" (one rank only)
      const char* colon = std::strchr(f, ':');
      if (!colon || std::atoi(colon + 1) == R) {
        if (!std::strncmp(f, "mapped", 6) && want_mapped) { mapped_ok = false; mapped_why_ = "test hook"; }
        if (!std::strncmp(f, "base", 4)) { ok = false; why = "test hook"; }
        if (!std::strncmp(f, "alltoall", 8) && want_a2a) { a2a_ok = false; a2a_why_ = "test hook"; }
      }
    }
    // the verdict, the same on every rank
    bool all_ok = ok, all_mapped = mapped_ok, all_a2a = a2a_ok;
    try {
      hr[0] = 4096.0 + (ok ? 0.0 : 1.0);
      hr[1] = 4096.0 + (mapped_ok ? 0.0 : 1.0);
      hr[2] = 4096.0 + (a2a_ok ? 0.0 : 1.0);
      SIPX_HIP(hipMemcpy(red, hr.data(), 3 * sizeof(double), hipMemcpyHostToDevice));
      c.allreduce_sum(red, 3, SIPX_F64, stream_);
      SIPX_HIP(hipStreamSynchronize(stream_));
      SIPX_HIP(hipMemcpy(hr.data(), red, 3 * sizeof(double), hipMemcpyDeviceToHost));
      all_ok = hr[0] == 4096.0 * W;
      all_mapped = hr[1] == 4096.0 * W;
      all_a2a = hr[2] == 4096.0 * W;
    } catch (const std::exception& ex) {
      all_ok = false;
      if (why.empty()) why = std::string("the verdict's all-reduce failed: ") + ex.what();
    }
    dfree(buf); dfree(hal);
    if (agree_buf_) dfree(red); else agree_buf_ = red;      // (kept: the ranks agree on the outcome of their allocations through it)
    if (vm_base) dfree(vm_base);
    if (!all_ok)
      throw std::runtime_error("communicator self-test failed (" + std::string(c.kind()) + ", rank " + std::to_string(R) + " of " + std::to_string(W) +
                               "): " + (why.empty() ? std::string("another rank reports wrong data") : why));
    selftest_ = "passed";
    if (want_mapped && !all_mapped) {
      selftest_ = "passed; exchange out of hipMemMap-backed memory failed (" + (mapped_why_.empty() ? std::string("on another rank") : mapped_why_) +
                  "): full-size arrays";
      selftest_mapped_failed_ = true;
    }
    if (want_a2a && !all_a2a) {
      selftest_ += "; all-to-all failed (" + (a2a_why_.empty() ? std::string("on another rank") : a2a_why_) + "): the l1-DFT set through an owner rank";
      selftest_alltoall_failed_ = true;
    }
  }

  // An array over the grid: `total` elements, entry g of block q at front + q * bstride + g.  Full size, or (slab_local_) backed
  // by memory for the grid points [wlo_, whi_) of every block only.
  T* galloc(long long total, long long front, int nblk, long long bstride) {
    if (!slab_local_) return dalloc<T>((size_t)total);
    std::vector<std::pair<size_t, size_t>> rg;
    for (int q = 0; q < std::max(nblk, 1); ++q) {
      const long long lo = std::max<long long>(0, front + (long long)q * bstride + wlo_);
      const long long hi = std::min<long long>(total, front + (long long)q * bstride + whi_);
      if (hi > lo) rg.push_back({(size_t)lo * sizeof(T), (size_t)hi * sizeof(T)});
    }
    return (T*)sparse_alloc_bytes((size_t)total * sizeof(T), rg, device_);
  }

  // A rank's part of a whole vector of the exchange layout (sparse arrays, a materialised set this rank does not project whole):
  // backed for its chunk of the layout -- what a fan gather sends and a scatter receives -- and for the planes its kernels touch
  // around its slab.
  T* loose_alloc(long long Npad) {
    const long long c0 = (long long)comm_->rank * chunk_, c1 = c0 + chunk_;
    const long long lo = std::min(std::max<long long>(0, wlo_), c0), hi = std::min(Npad, std::max(std::max<long long>(0, whi_), c1));
    std::vector<std::pair<size_t, size_t>> rg{{(size_t)lo * sizeof(T), (size_t)hi * sizeof(T)}};
    return (T*)sparse_alloc_bytes((size_t)Npad * sizeof(T), rg, device_);
  }

  // does the sweep take this context / iteration?  (asked before any search is queued; fills the layout part of `ma`)
  bool sweep_applicable(int flags, MultiArgs<T>& ma, bool planning = false) {
    if (!yl_multi_ || mk_ || (comm_ && !slab_)) return false;
    const bool feas = (flags & SIPX_YL_FEAS) != 0;
    ma.nblk = 0;
    ma.rhs = nullptr;
    ma.flags = flags;
    const long long nlast = G_.n[ndim_ - 1];
    ma.zlo = 0; ma.zhi = nlast; ma.zsum = 0;
    if (slab_ && !planning) {                   // the rank's planes, plus the last plane of the rank below (recomputed, see Gyl_)
      ma.zsum = r0_ / plane_;
      ma.zlo = ma.zsum - ((prev_ >= 0 && r1_ > r0_) ? 1 : 0);
      ma.zhi = r1_ / plane_;
      if (r1_ <= r0_) ma.zlo = ma.zhi = ma.zsum = 0;
    }
    bool behind = false;                        // a loose set has been seen: what follows is not part of the fused right-hand side
    for (int i = 0; i < p_n_; ++i) {
      const SetState<T>& s = sets_[i];
      if (planning ? !sweep_eligible(s) : !s.in_sweep) {
        // a set the sweep cannot take: it keeps its per-set kernels (one rank only; never a caller-supplied sparse operator,
        // whose right-hand side term is added out of order)
        // (... or slab-decomposed, the sets projected on a materialised v: SetState::slab_ext, fan)
        if ((slab_ && !(s.slab_ext || s.fan || s.slab_card || s.slab_dft)) || (comm_ && !slab_) || s.custom || !s.owned || s.dist_ext || (planning ? false : !sweep_partial_)) return false;
        behind = true;
        continue;
      }
      if (ma.nblk + s.nblk_or1() > MULTI_MAXB) return false;
      // (the sweep takes the feasibility estimate of an element-wise set on the identity only)
      if (feas && i < pp_n_ && !s.two_pass && s.nblk > 0) return false;
      for (int qb = 0; qb < s.nblk_or1(); ++qb) {
        MultiBlk<T>& B = ma.b[ma.nblk++];
        B.dir = s.nblk == 0 ? -1 : s.dir[qb];
        B.last = qb == s.nblk_or1() - 1;
        B.dist = s.is_dist ? 1 : 0;
        B.prox = s.prox;
        B.in_rhs = behind ? 0 : 1;
      }
    }
    if (ma.nblk == 0) return false;
    return K<T>::yl_multi(stream_, G_, ma, true);
  }
  static bool sweep_eligible(const SetState<T>& s) {
    return s.owned && !s.custom && !s.ext_kind && !s.dist_ext &&
           (s.prox == PX_BOUNDS || s.prox == PX_L1 || s.prox == PX_PROX_L1 || s.prox == PX_L2 || s.prox == PX_ANNULUS || s.prox == PX_DIST);
  }

  // Threshold / scale searches of the two-pass sets `tp` of a slab-decomposed grid, all in lock step, through the SPECULATIVE
  // EXCHANGE (used for the searches of the y/l update, feas_ps = false, and for those of the feasibility estimates, on the sets'
  // second scalar state and v = A x itself).  gseg[j] / chunk: the set's segment in rank 0's chunk of the full-size exchange
  // buffer; ctl[j]: the set's pinned words.
  void spec_exchange_searches(const std::vector<int>& tp, std::vector<SetArgs<T>>& args, std::vector<SampleCtl>& ctl,
                              const std::vector<T*>& gseg, long long chunk, int v_is_s, bool feas_ps, int it, bool lean_group = false) {
    const size_t RS = (size_t)(PREP_SLOTS + 1 + 2 * comm_->world);
    auto PS = [&](int i) { return feas_ps ? sets_[i].psf : sets_[i].ps; };
    // SPECULATIVE EXCHANGE (kernels_proj.hip, k_spec_pack): first pass of every set, then ONE all-gather carrying every
    // rank's probe sums and the magnitudes it gathered inside the speculative range.  Every rank adds the sums up itself,
    // decides, and -- when the range held theta, the rule once rho and gamma move slowly -- solves from the gathered values:
    // one collective per iteration for all the searches.  Whether a set needs its fallback (refinement rounds with an
    // all-reduce each, then the full-size exchange) the host reads from a pinned word per set; the decision kernel,
    // the unpacking and the solve are queued before that wait, so the device does not idle on it.
    const long long fseg = hooks_.fcap + fast_hdr<T>();
    const long long fchunk = (long long)tp.size() * fseg;
    const unsigned seq = ++spec_seq_ & 0x3fffffffu;
    // The sets' chains of small kernels (slot sums, packing; decision, unpacking, solve) run side by side on the set
    // streams -- every set has its own partial slots, gather buffer and segments -- the collective itself on the engine stream.
    auto fork = [&](hipEvent_t ev) { if (set_streams_) SIPX_HIP(hipEventRecord(ev, stream_)); };
    auto join = [&]() {
      for (size_t k = 0; k < pool_.size(); ++k) {
        if (pool_[k] == stream_) continue;
        SetState<T>* last = nullptr;
        for (int i : tp)
          if (sets_[i].st == pool_[k]) last = &sets_[i];
        if (!last) continue;
        SIPX_HIP(hipEventRecord(last->ev, last->st));
        SIPX_HIP(hipStreamWaitEvent(stream_, last->ev, 0));
      }
    };
    if (spec_batch_ && (int)tp.size() <= SPEC_MAX_SETS) {
      // Batched form (the default): every set's first pass on the engine stream, then TWO launches for all sets -- sums of
      // the partial slots + packing, and, after the all-gather, decision + unpacking + solve (one workgroup per set).  A rank's
      // share of the grid is small when the ranks are many and the iteration is then bound by the launches the host can
      // issue: 21 small kernels and four cross-stream dependencies of three searches become 8 launches on one stream.
      SpecPackArgs<T> pk;
      SpecFinishArgs<T> fa;
      pk.nsets = fa.nsets = (int)tp.size();
      pk.cap = hooks_.fcap;
      fa.world = comm_->world; fa.fchunk = fchunk; fa.seq = seq;
      // (the lean first passes of the l1 sets in one sweep: everything that prepares a search -- rescaling, sampled prediction --
      //  was queued on this very stream before, so the device-side state a lean pass reads is final when the sweep starts)
      (void)lean_group;
      if (!feas_ps && !v_is_s && Gr_.n[0] % 4 == 0 && slab_lean_multi_) {
        LeanMulti<T> lm;
        lm.ns = 0;
        std::vector<size_t> who;
        for (size_t j = 0; j < tp.size() && lm.ns < LEAN_MAX; ++j) {
          SetState<T>& s = sets_[tp[j]];
          if (s.prox != PX_L1 || (args[j].flags & F_NOSPEC)) continue;
          LeanSet<T>& L = lm.s[lm.ns++];
          L.a = args[j]; L.a.ps = s.ps; L.ps = s.ps; L.compact = s.cbuf; L.partials = s.ptmp; L.maxpart = s.mpart;
          who.push_back(j);
        }
        if (lm.ns >= 2) {
          K<T>::lean_multi(stream_, Gr_, lm);
          for (size_t j : who) {
            ctl[j].lean_done = 1;
            ctl[j].lean_known = (hlean_[tp[j]] >> 16) & 1;       // published by the set's last solve: its full first pass need not be launched
          }
        }
      }
      for (size_t j = 0; j < tp.size(); ++j) {
        SetState<T>& s = sets_[tp[j]];
        ctl[j].verdict = (unsigned*)hverd_ + tp[j];
        ctl[j].seq = seq;
        K<T>::proj_scalars_stage(13, stream_, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], &hooks_,
                                 stage_ + j * RS, gseg[j], chunk);
        SpecPackSet<T>& P = pk.s[j];
        P.ps = PS(tp[j]); P.partials = s.ptmp; P.maxpart = s.mpart; P.compact = s.cbuf;
        P.seg = fbuf_ + (long long)comm_->rank * fchunk + (long long)j * fseg;
        P.is_l1 = s.prox == PX_L1 ? 1 : 0;
        SpecFinishSet<T>& F = fa.s[j];
        F.ps = PS(tp[j]);
        F.da = DecideArgs{args[j].prox, (args[j].flags & F_NOSPEC) ? 1 : 0, (double)args[j].plo, (double)args[j].phi, 64.0, (double)hooks_.gcap, s.Mtrue};
        F.reg = stage_ + j * RS;
        F.fseg0 = fbuf_ + (long long)j * fseg;
        F.compact = s.cbuf; F.partials = s.ptmp; F.radius = args[j].phi;
        F.host_want = ctl[j].host_want; F.verdict = ctl[j].verdict;
      }
      K<T>::spec_sums_pack(stream_, pk);
      comm_->allgather(fbuf_, (size_t)fchunk, dtype_code(), stream_);
      K<T>::spec_finish(stream_, fa);
    } else {
    fork(ev_fork_);
    for (size_t j = 0; j < tp.size(); ++j) {
      SetState<T>& s = sets_[tp[j]];
      hipStream_t q = (set_streams_ && s.st) ? s.st : stream_;
      if (q != stream_) SIPX_HIP(hipStreamWaitEvent(q, ev_fork_, 0));
      ctl[j].verdict = (unsigned*)hverd_ + tp[j];
      ctl[j].seq = seq;
      K<T>::proj_scalars_stage(0, q, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], &hooks_,
                               stage_ + j * RS, gseg[j], chunk);
      K<T>::proj_scalars_stage(5, q, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], &hooks_,
                               stage_ + j * RS, fbuf_ + (long long)j * fseg, fchunk);
    }
    join();
    comm_->allgather(fbuf_, (size_t)fchunk, dtype_code(), stream_);
    fork(ev_fork2_);
    for (size_t j = 0; j < tp.size(); ++j) {
      SetState<T>& s = sets_[tp[j]];
      hipStream_t q = (set_streams_ && s.st) ? s.st : stream_;
      if (q != stream_) SIPX_HIP(hipStreamWaitEvent(q, ev_fork2_, 0));
      K<T>::proj_scalars_stage(6, q, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], &hooks_,
                               stage_ + j * RS, fbuf_ + (long long)j * fseg, fchunk);
    }
    join();
    }
    std::vector<size_t> fb;                  // the sets whose search goes on (the same on every rank)
    bool refine = false;
    for (size_t j = 0; j < tp.size(); ++j) {
      const unsigned w = wait_verdict(hverd_ + tp[j], seq);
      if (w & 1u) fb.push_back(j);
      refine |= (w & 2u) != 0;
    }
    spec_searches_ += (long long)tp.size();
    spec_fallbacks_ += (long long)fb.size();
    static const bool spec_debug = std::getenv("SIPX_SPEC_DEBUG") != nullptr;
    if (spec_debug) {
      for (size_t j : fb) dump_ps(tp[j], stage_ + j * RS);
      std::fprintf(stderr, "[sipx spec] it %d:", it);
      for (size_t j = 0; j < tp.size(); ++j) std::fprintf(stderr, " set %d verdict %u", tp[j], (unsigned)(hverd_[tp[j]] & 3u));
      std::fprintf(stderr, "\n");
    }
    if (!fb.empty()) {
      // Fallback (the summed first-pass sums of every set are in its region of stage_, where k_spec_decide left them):
      // refinement rounds -- gated probe pass, ONE all-reduce, decision -- for as long as some set's bracket holds more than
      // the exchange segments take (the host reads that from the sets' pinned words after every round: exactly as many
      // all-reduces as are needed, at most L1_REFINES_SLAB), then the compaction of every final bracket and the full-size
      // all-gather.  The sets the exchange settled take no part.
      int max_rounds = 6;
      if (const char* e = std::getenv("SIPX_L1_ROUNDS_MAX")) max_rounds = std::max(0, std::min(6, std::atoi(e)));      // tests: force an overflow
      for (int rep = 0; refine && rep < max_rounds; ++rep) {
        const unsigned rseq = ++spec_seq_ & 0x3fffffffu;
        for (size_t j : fb) {
          SetState<T>& s = sets_[tp[j]];
          K<T>::proj_scalars_stage(8, stream_, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], &hooks_,
                                   stage_ + j * RS, gseg[j], chunk);
        }
        comm_->allreduce_sum(stage_, tp.size() * RS, SIPX_F64, stream_);
        for (size_t j : fb) {
          SetState<T>& s = sets_[tp[j]];
          ctl[j].seq = rseq;
          K<T>::proj_scalars_stage(9, stream_, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], &hooks_,
                                   stage_ + j * RS, gseg[j], chunk);
        }
        refine = false;
        for (size_t j : fb)
          if (sets_[tp[j]].prox == PX_L1) refine |= (wait_verdict(hverd_ + tp[j], rseq) & 2u) != 0;
        spec_rounds_ += 1;
        if (spec_debug) {
          std::fprintf(stderr, "[sipx spec]   round %d -> refine %d\n", rep + 1, (int)refine);
          for (size_t j : fb) dump_ps(tp[j], stage_ + j * RS);
        }
      }
      for (int stage : {12, 3}) {
        for (size_t j : fb) {
          SetState<T>& s = sets_[tp[j]];
          K<T>::proj_scalars_stage(stage, stream_, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], &hooks_,
                                   stage_ + j * RS, gseg[j], chunk);
        }
        if (stage == 12) comm_->allgather(gbuf_, (size_t)chunk, dtype_code(), stream_);
      }
    }
  }

  // One rank, the sweep does the updates: the threshold / scale searches of ALL two-pass sets as ONE chain of launches on the
  // engine stream (round 4) -- rescaling after a change of rho and the sampled prediction for every set that asks for them (one
  // launch each), the lean first passes in one sweep that reads x once (k_lean_multi), a full first pass per set only where the
  // host does not know that the pass will be a lean one, then two launches for all sets: sums of the partial slots (+ header),
  // decision + solve (one workgroup per set) -- the batched form of the slab-decomposed search without its exchange.  Whether
  // a set needs its fallback sweeps (theta left the speculative range) the host reads from one pinned word per set, published
  // by the decision before the solve starts: the gated launches of the per-set chains -- three no-ops per search in the common
  // case, each a few microseconds on the critical path -- are only made when they have work, and no stream forks or joins.
  // Per iteration of the headline list: 5 launches instead of 27 on three streams, x read once instead of three times.
  // Same decisions (decide_body with cap_max = 0), same gathered values, same double-double solve: theta bit for bit.
  void batched_searches(int flags, const double* rho, const double* gamma) {
    std::vector<int> tp;
    for (int i = 0; i < p_n_; ++i)
      if (sets_[i].two_pass && sets_[i].in_sweep) tp.push_back(i);
    if (tp.empty()) return;
    run_batched(tp, false, flags, rho, gamma);
    if (flags & SIPX_YL_FEAS) {               // ||P_i(s) - s|| with s = A_i x itself: the sets' second scalar state
      std::vector<int> tf;
      for (int i : tp)
        if (i < pp_n_) tf.push_back(i);
      if (!tf.empty()) {
        run_batched(tf, true, flags, rho, gamma);
        for (int i : tf) {
          SetArgs<T> a = set_args(sets_[i], (T)rho[i], (T)gamma[i], flags);
          K<T>::proj_dist_set(stream_, Gr_, a, 1, sets_[i].psf, part_sets_ + ((size_t)i * SLOTS + SL_FE2) * NB);
        }
      }
    }
  }
  void run_batched(const std::vector<int>& tp, bool feas_ps, int flags, const double* rho, const double* gamma) {
    const size_t RS = (size_t)(PREP_SLOTS + 1 + 2);
    const long long fseg = fast_hdr<T>();
    const int v_is_s = feas_ps ? 1 : 0;
    const unsigned seq = ++spec_seq_ & 0x3fffffffu;
    std::vector<SetArgs<T>> args(tp.size());
    std::vector<SampleCtl> ctl(tp.size());
    auto PS = [&](int i) { return feas_ps ? sets_[i].psf : sets_[i].ps; };
    const bool vec = Gr_.n[0] % 4 == 0;
    RescaleMulti<T> rs;
    rs.n = 0;
    SampleMulti<T> sm;
    sm.ns = 0;
    for (size_t j = 0; j < tp.size(); ++j) {
      SetState<T>& s = sets_[tp[j]];
      args[j] = set_args(s, (T)rho[tp[j]], (T)gamma[tp[j]], flags);
      ctl[j].verdict = (unsigned*)hverd_ + tp[j];
      ctl[j].seq = seq;
      if (feas_ps) {
        // the search of a feasibility estimate comes every tenth iteration: its own last theta is ten iterations old and missed
        // the range every time (three fallbacks of two sweeps each per such iteration: 7 % of the 512^3 window) -- a sampled
        // estimate first, whenever the set's last such search asked for one (device side: ProjScalars::want_sample)
        if (feas_sample_ && l1_sample_ && s.prox == PX_L1 && vec) {
          SampleSet<T>& S = sm.s[sm.ns++];
          S.a = args[j]; S.a.ps = s.psf; S.ps = s.psf; S.partials = s.ptmp; S.reg = nullptr; S.true_len = s.Mtrue;
        }
        continue;
      }
      ctl[j].host_want = (int*)hlean_ + tp[j];
      ctl[j].runs = l1_sample_runs_;
      const bool l1 = s.prox == PX_L1;
      const bool rescaled = l1 && s.last_rho > T(0) && s.last_rho != args[j].rho;      // v rescaled: theta moves like 1/rho
      if (rescaled) { rs.ps[rs.n] = s.ps; rs.factor[rs.n++] = (double)s.last_rho / (double)args[j].rho; }
      if (l1_sample_ && l1 && vec && (rescaled || (hlean_[tp[j]] & 0xff) != 0)) {
        SampleSet<T>& S = sm.s[sm.ns++];
        S.a = args[j]; S.a.ps = s.ps; S.ps = s.ps; S.partials = s.ptmp; S.reg = nullptr; S.true_len = s.Mtrue;
      }
      s.last_rho = args[j].rho;
      s.last_gamma = args[j].gamma;
    }
    K<T>::ps_rescale_multi(stream_, rs);
    sm.v_is_s = v_is_s;
    if (sm.ns > 0) K<T>::sample_multi(10, stream_, Gr_, sm, l1_sample_runs_, nullptr);
    // the passes: groups of up to LEAN_MAX sets per launch (x read once per group) -- the lean first passes of the l1 sets whose
    // device-side state asks for one, then the full first passes of everybody else; each kernel returns at once when no set
    // of its group wants it.  (A grid whose lines are no multiple of four points keeps one scalar pass per set.)
    auto group = [&](const std::vector<size_t>& who, size_t from) {
      LeanMulti<T> L;
      L.ns = 0;
      L.v_is_s = v_is_s;
      for (size_t k = from; k < who.size() && L.ns < LEAN_MAX; ++k) {
        const size_t j = who[k];
        SetState<T>& s = sets_[tp[j]];
        LeanSet<T>& S = L.s[L.ns++];
        S.a = args[j]; S.a.ps = PS(tp[j]); S.ps = PS(tp[j]); S.compact = s.cbuf; S.partials = s.ptmp; S.maxpart = s.mpart;
      }
      return L;
    };
    std::vector<size_t> all(tp.size());
    for (size_t j = 0; j < tp.size(); ++j) all[j] = j;
    // (pass_multi_: the FULL first passes and the fallback passes of a group in one sweep as well -- measured and NOT the
    //  default: eight probes for three sets make that kernel ALU-bound at 161-173 VGPRs, 512^3 first passes 800-1000 us against
    //  3 x 285 us one after the other, refinement 1009 against 3 x 264; 512^3 116.2 against 118.7 it/s.  SIPX_PASS_MULTI=1, tests)
    if (vec && pass_multi_) {
      std::vector<size_t> l1s;
      for (size_t j = 0; j < tp.size(); ++j)
        if (sets_[tp[j]].prox == PX_L1 && !(args[j].flags & F_NOSPEC)) l1s.push_back(j);
      for (size_t k = 0; k < l1s.size(); k += LEAN_MAX) K<T>::lean_multi(stream_, Gr_, group(l1s, k));
      for (size_t k = 0; k < all.size(); k += LEAN_MAX) K<T>::pass_multi(0, stream_, Gr_, group(all, k), v_is_s);
    } else {
      // the lean first passes of the l1 sets in one sweep (x read once), a full first pass per set only where the host does not
      // know from the set's pinned word that the pass will be a lean one (it returns at once when the device-side state says lean)
      if (vec) {
        std::vector<size_t> l1s;
        for (size_t j = 0; j < tp.size(); ++j)
          if (sets_[tp[j]].prox == PX_L1 && !(args[j].flags & F_NOSPEC)) l1s.push_back(j);
        for (size_t k = 0; k < l1s.size(); k += LEAN_MAX) K<T>::lean_multi(stream_, Gr_, group(l1s, k));
        for (size_t j : l1s) {
          ctl[j].lean_done = 1;
          ctl[j].lean_known = feas_ps ? 0 : ((hlean_[tp[j]] >> 16) & 1);      // published by the set's last solve; nothing since can have taken the flag back
        }
      }
      for (size_t j = 0; j < tp.size(); ++j) {
        SetState<T>& s = sets_[tp[j]];
        K<T>::proj_scalars_stage(13, stream_, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], nullptr,
                                 stage_ + j * RS, nullptr, 0);
      }
    }
    SpecPackArgs<T> pk;
    SpecFinishArgs<T> fa;
    pk.nsets = fa.nsets = (int)tp.size();
    pk.local = fa.local = 1;
    pk.cap = 0;
    fa.world = 1; fa.fchunk = (long long)tp.size() * fseg; fa.seq = seq;
    for (size_t j = 0; j < tp.size(); ++j) {
      SetState<T>& s = sets_[tp[j]];
      SpecPackSet<T>& P = pk.s[j];
      P.ps = PS(tp[j]); P.partials = s.ptmp; P.maxpart = s.mpart; P.compact = s.cbuf;
      P.seg = fbuf_ + (long long)j * fseg;
      P.is_l1 = s.prox == PX_L1 ? 1 : 0;
      SpecFinishSet<T>& F = fa.s[j];
      F.ps = PS(tp[j]);
      F.da = DecideArgs{args[j].prox, (args[j].flags & F_NOSPEC) ? 1 : 0, (double)args[j].plo, (double)args[j].phi, 64.0, 0.0, s.Mtrue};
      F.reg = stage_ + j * RS;
      F.fseg0 = fbuf_ + (long long)j * fseg;
      F.compact = s.cbuf; F.partials = s.ptmp; F.radius = args[j].phi;
      F.host_want = ctl[j].host_want; F.verdict = ctl[j].verdict;
    }
    K<T>::spec_sums_pack(stream_, pk);
    K<T>::spec_finish(stream_, fa);
    // Fallback of the sets whose pinned verdict asks for it (theta left the speculative range, or the range gathered too much):
    // refinement pass + decision (where the decision asked for one), compaction of the bracket, solve -- the passes of all
    // such sets in one sweep each.
    std::vector<size_t> fb;
    bool refine = false;
    for (size_t j = 0; j < tp.size(); ++j) {
      const unsigned w = wait_verdict(hverd_ + tp[j], seq);
      batch_searches_ += 1;
      if (!(w & 1u)) continue;
      batch_fallbacks_ += 1;
      if (env_knobs().trace_searches)
        std::fprintf(stderr, "[sipx search] search %lld (set %d%s): fallback%s\n", batch_searches_, tp[j], feas_ps ? ", feasibility" : "", (w & 2u) ? " with refinement" : "");
      fb.push_back(j);
      refine |= (w & 2u) != 0;
    }
    static const bool spec_debug = std::getenv("SIPX_SPEC_DEBUG") != nullptr;
    if (spec_debug && !feas_ps) {                // (diagnostics: the state every search of the chain ended its first stage with; synchronises)
      std::fprintf(stderr, "[sipx spec] batched chain, search seq %u\n", seq);
      for (size_t j = 0; j < tp.size(); ++j) dump_ps(tp[j], stage_ + j * RS);
    }
    if (fb.empty()) return;
    auto tail = [&](int stage) {
      for (size_t j : fb) {
        SetState<T>& s = sets_[tp[j]];
        K<T>::search_tail(stage, stream_, args[j], PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], stage_ + j * RS);
      }
    };
    if (vec && pass_multi_) {
      if (refine) {
        for (size_t k = 0; k < fb.size(); k += LEAN_MAX) K<T>::pass_multi(1, stream_, Gr_, group(fb, k), v_is_s);
        tail(1);
      }
      for (size_t k = 0; k < fb.size(); k += LEAN_MAX) K<T>::pass_multi(2, stream_, Gr_, group(fb, k), v_is_s);
      tail(3);
    } else {
      for (size_t j : fb) {
        SetState<T>& s = sets_[tp[j]];
        for (int stage : {1, 2, 3}) {
          if (stage == 1 && !refine) continue;
          K<T>::proj_scalars_stage(stage, stream_, Gr_, args[j], v_is_s, PS(tp[j]), s.ptmp, s.mpart, s.cbuf, s.Mtrue, ctl[j], nullptr,
                                   stage_ + j * RS, nullptr, 0);
        }
      }
    }
  }

  // one rank (or no communicator): the searches of the two-pass sets on the set streams, joined before the sweep
  void sweep_searches(int flags, const double* rho, const double* gamma) {
    const bool feas = (flags & SIPX_YL_FEAS) != 0;
    // The lean first passes of the l1 searches in ONE sweep (k_lean_multi: x read once): on iterations where no search of the
    // group is rescaled or sampled first (the device-side state a lean pass needs is then final when the engine stream gets
    // here), for two or three l1 sets on stencil operators of this grid.  Each set's chain then launches its full first
    // pass only, which returns at once for the sets the sweep served.
    std::vector<char> lean_done(p_n_, 0);
    if (lean_multi_ && Gr_.n[0] % 4 == 0) {
      LeanMulti<T> lm;
      lm.ns = 0;
      bool ok = true;
      std::vector<int> who;
      for (int i = 0; i < p_n_ && ok; ++i) {
        SetState<T>& s = sets_[i];
        if (!s.two_pass || !s.in_sweep || s.prox != PX_L1 || s.custom || s.ext_kind) continue;
        SetArgs<T> a = set_args(s, (T)rho[i], (T)gamma[i], flags);
        if (a.flags & F_NOSPEC) continue;
        const bool rescaled = s.last_rho > T(0) && s.last_rho != a.rho;
        const bool sampled = l1_sample_ && (rescaled || (hlean_[i] & 0xff) != 0);
        if (rescaled || sampled) { ok = false; break; }
        if (lm.ns == LEAN_MAX) break;
        LeanSet<T>& L = lm.s[lm.ns++];
        L.a = a; L.a.ps = s.ps; L.ps = s.ps;
        L.compact = s.cbuf ? s.cbuf : scr_c_;
        L.partials = s.ptmp ? s.ptmp : part_tmp_;
        L.maxpart = s.mpart ? s.mpart : maxpart_;
        who.push_back(i);
      }
      bool own = true;                     // every set of the group needs scratch of its own
      for (int i : who) own &= sets_[i].ptmp != nullptr && sets_[i].mpart != nullptr && sets_[i].cbuf != nullptr;
      if (ok && own && lm.ns >= 2) {
        K<T>::lean_multi(stream_, Gr_, lm);
        for (int i : who) lean_done[i] = 1;
      }
    }
    if (set_streams_) SIPX_HIP(hipEventRecord(ev_fork_, stream_));       // x is final: the searches may start
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>& s = sets_[i];
      if (!s.two_pass || !s.in_sweep) continue;
      SetArgs<T> a = set_args(s, (T)rho[i], (T)gamma[i], flags);
      hipStream_t q = s.st ? s.st : stream_;
      double* ptmp = s.ptmp ? s.ptmp : part_tmp_;
      T* mpart = s.mpart ? s.mpart : maxpart_;
      T* cbuf = s.cbuf ? s.cbuf : scr_c_;
      double* part = part_sets_ + (size_t)i * SLOTS * NB;
      if (q != stream_) SIPX_HIP(hipStreamWaitEvent(q, ev_fork_, 0));
      const bool rescaled = a.prox == PX_L1 && s.last_rho > T(0) && s.last_rho != a.rho;
      if (rescaled) K<T>::ps_rescale(q, s.ps, (double)s.last_rho / (double)a.rho);
      SampleCtl ctl;
      ctl.host_want = (int*)hlean_ + i;
      ctl.host_ovf = (int*)hovf_ + i;
      ctl.runs = l1_sample_runs_;
      ctl.enable = l1_sample_ && a.prox == PX_L1 && (rescaled || (hlean_[i] & 0xff) != 0);
      ctl.lean_done = lean_done[i];
      K<T>::proj_scalars_set(q, Gr_, a, 0, s.ps, ptmp, mpart, cbuf, s.Mtrue, ctl, nullptr);
      s.last_rho = a.rho;
      s.last_gamma = a.gamma;
      if (feas && i < pp_n_) {                 // ||P_i(s) - s|| with s = A_i x produced on the fly; its own warm-started scalars
        SampleCtl cf;                          // (a sampled estimate first, as in the batched chain: run_batched)
        cf.runs = l1_sample_runs_;
        cf.enable = feas_sample_ && l1_sample_ && a.prox == PX_L1;
        K<T>::proj_scalars_set(q, Gr_, a, 1, s.psf, ptmp, mpart, cbuf, s.Mtrue, cf, nullptr);
        K<T>::proj_dist_set(q, Gr_, a, 1, s.psf, part + (size_t)SL_FE2 * NB);
      }
    }
    for (size_t k = 0; k < pool_.size(); ++k) {                         // join: theta / scale of every set are known
      if (pool_[k] == stream_) continue;
      SetState<T>* last = nullptr;
      for (int i = 0; i < p_n_; ++i)
        if (sets_[i].two_pass && sets_[i].in_sweep && sets_[i].st == pool_[k]) last = &sets_[i];
      if (!last) continue;
      SIPX_HIP(hipEventRecord(last->ev, last->st));
      SIPX_HIP(hipStreamWaitEvent(stream_, last->ev, 0));
    }
  }

  // The y/l update of EVERY set in one sweep over the grid (kernels_multi.hip), on the engine stream, once the threshold /
  // scale of every two-pass set is known: one kernel updates all sets, forms the r_pri / r_dual / obj sums -- on
  // Barzilai-Borwein iterations the six BB sums and the snapshot refresh, every tenth iteration the feasibility estimates of
  // the element-wise sets -- and, when the caller has announced that rho cannot change before the next iteration (fuse_rhs_),
  // writes the right-hand side of that iteration.
  // The sweep never updates in place (neighbouring tiles re-read the OLD y, l of a few points): it writes into the pair that
  // holds the old snapshot (BB / first iteration: it is read first), into the free pair, or -- when both other pairs are
  // taken, i.e. the snapshot must survive and sits in the other pair -- into a third pair, allocated on first need.
  void sweep_launch(int flags, const double* rho, const double* gamma, MultiArgs<T>& ma) {
    const bool first = (flags & SIPX_YL_FIRST) != 0, bb = (flags & SIPX_YL_BB) != 0 && !first;
    ma.nblk = 0;
    std::vector<int> target(p_n_, 0);          // 0: the other pair (y0, l0); 2: the third pair
    bool behind = false;
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>& s = sets_[i];
      if (!s.in_sweep) { behind = true; target[i] = -1; continue; }
      SetArgs<T> a = set_args(s, (T)rho[i], (T)gamma[i], flags);
      target[i] = (!first && !bb && s.snap == 1) ? 2 : 0;
      if (target[i] == 2 && !s.y2) throw std::runtime_error("internal: the third y / l pair of the sweep was not allocated at sipx_finalize");
      T* ty = target[i] == 2 ? s.y2 : s.y0;
      T* tl = target[i] == 2 ? s.l2 : s.l0;
      for (int qb = 0; qb < s.nblk_or1(); ++qb) {
        MultiBlk<T>& B = ma.b[ma.nblk++];
        const long long off = (long long)qb * G_.N;
        B.y = s.y + off; B.l = s.l + off;
        B.yo = ty + off; B.lo = tl + off;
        B.lh0 = s.lh0 + off; B.s0 = s.s0 + off;
        B.y0 = a.y0 + off; B.l0 = a.l0 + off;         // the snapshot pair (set_args)
        B.ps = s.ps;
        B.dir = s.nblk == 0 ? -1 : s.dir[qb];
        B.set = i;
        B.first = qb == 0;
        B.last = qb == s.nblk_or1() - 1;
        B.dist = s.is_dist ? 1 : 0;
        B.feas_el = (i < pp_n_ && (s.prox == PX_BOUNDS || s.prox == PX_PROX_L1)) ? 1 : 0;
        B.ih = s.nblk == 0 ? T(0) : s.ih[qb];
        B.rho = a.rho; B.rho1 = a.rho1; B.gamma = a.gamma;
        B.prox = s.prox; B.plo = s.plo; B.phi = s.phi;
        B.in_rhs = behind ? 0 : 1;
      }
    }
    ma.x = x_; ma.m = m_; ma.xold = xold_;
    // x0 mode: s_0 = A x_0 from the ring buffer that held x on the last BB / first iteration; the new snapshot is x itself --
    // the coming x-steps leave its buffer alone (argmin_x)
    ma.x0 = (x0_mode_ && x_snap_ >= 0) ? xr_[x_snap_] : (x0_mode_ ? x_ : nullptr);
    ma.x0w = nullptr;
    ma.rhs = fuse_rhs_ ? rhs_ : nullptr;
    ma.partials = part_sets_;
    if (!K<T>::yl_multi(stream_, G_, ma)) throw std::runtime_error("internal: the fused y/l sweep refused a block list it was prepared for");
    rhs_fused_ = ma.rhs != nullptr;
    if (x0_mode_ && (first || bb)) x_snap_ = x_cur_;
    for (int i = 0; i < p_n_; ++i) {           // (y, l) always names the current iterate; snap says where the snapshot sits
      SetState<T>& s = sets_[i];
      if (target[i] < 0) continue;
      if (target[i] == 2) {
        std::swap(s.y, s.y2); std::swap(s.l, s.l2);                     // the snapshot stays in (y0, l0)
      } else {
        std::swap(s.y, s.y0); std::swap(s.l, s.l0);
        if (first || bb) s.snap = 0;                                    // the pair just written IS the new snapshot
        else if (s.snap == 0) s.snap = 1;                               // the snapshot stayed behind in what is now (y0, l0)
      }
    }
  }

  // second half of update_y_l: waits for the reduced sums and turns them into the per-set scalars.  The whole-solve loop
  // calls it late (defer_sums_), after it has queued the work of the next iteration that cannot depend on them.
  void collect_set_sums(const double* rho, double* r_pri, double* r_dual, double* feas) {
    if (!sums_pending_) throw std::runtime_error("no y/l update is pending");
    sums_pending_ = false;
    const int flags = sums_flags_;
    if (sums_by_word_) wait_word(sums_word_, sums_seq_);
    else SIPX_HIP(hipEventSynchronize(sums_event_));
    have_log_sums_ = false;
    if (slab_) {
      // a threshold search whose final bracket, over all ranks, held more magnitudes than the exchange segments: every rank saw
      // the same segment headers and raised its word, so every rank leaves here with the same error -- never with NaN iterates
      for (int i = 0; i < p_n_; ++i) {
        if (!hovf_[i]) continue;
        for (int k = 0; k <= p_n_; ++k) hovf_[k] = 0;
        ProjScalars<T> h;                    // what the search knew when it gave up (diagnostics): the update's search or the feasibility estimate's
        SIPX_HIP(hipStreamSynchronize(stream_));
        char dbg[512];
        int at = 0;
        for (const ProjScalars<T>* d : {sets_[i].ps, sets_[i].psf}) {
          if (!d) continue;
          SIPX_HIP(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
          if (!h.gather_overflow) continue;
          at += std::snprintf(dbg + at, sizeof dbg - at, " [%s: bracket (%.9g, %.9g], refinement rounds %d, ||v||_1 %.6g, radius %.6g, speculative range (%.9g, %.9g]]",
                              d == sets_[i].ps ? "y/l update" : "feasibility estimate", h.lo, h.hi, h.rounds_used, h.asum, (double)sets_[i].phi, h.spec_lo, h.spec_hi);
        }
        dbg[at] = 0;
        throw std::runtime_error("l1 threshold search of set " + std::to_string(i) + ": the magnitudes inside the final bracket, gathered over all "
                                 "ranks, exceed the exchange segment of " + std::to_string(hooks_.gcap) + " values per rank "
                                 "(slab decomposition) after the refinement rounds of the search; the iterate of this step is not valid -- "
                                 "use the set decomposition for this problem" + std::string(dbg));
      }
    }
    const bool all = comm_ != nullptr;       // sharded: the all-reduced sums of every set are here, on every rank
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>& s = sets_[i];
      if (r_pri) r_pri[i] = 0;
      if (r_dual) r_dual[i] = 0;
      if (!s.owned && !all) continue;
      const double* h = hres_ + (size_t)i * SLOTS;
      std::copy(h, h + SLOTS, s.sums);
      if (r_pri) r_pri[i] = (double)(T)std::sqrt(h[SL_RPRI]);                                   // update_y_l.jl:81
      // :84; in Minkowski mode [A A]'d = [A'd; A'd] doubles the sum of squares
      if (r_dual) r_dual[i] = (double)((T)rho[i] * (T)std::sqrt((s.comp == 3 ? 2.0 : 1.0) * (s.ident ? h[SL_DY] : h[SL_ADJ])));
      s.bb_valid = (flags & (SIPX_YL_BB | SIPX_YL_FIRST)) != 0;
      if (s.is_dist) {
        obj_ss_ = h[SL_OBJ]; evo_ss_ = h[SL_EVO]; xx_ss_ = h[SL_XX];     // obj = 1/2 ||TD_OP[end] x - m||^2  PARSDMM.jl:139-143
        if (mk_) { evo_ss_ = hres_[(size_t)p_n_ * SLOTS + SL_EVO]; xx_ss_ = hres_[(size_t)p_n_ * SLOTS + SL_XX]; }
        have_log_sums_ = true;
      }
    }
    if (all && !slab_dist_logs_) {           // the slab sums of argmin_x: the distance-term kernel saw x_old on its own slab only
      const double* g = hres_ + (size_t)p_n_ * SLOTS;
      obj_ss_ = g[SL_OBJ]; evo_ss_ = g[SL_EVO]; xx_ss_ = g[SL_XX];
      have_log_sums_ = true;
    }
    if ((flags & SIPX_YL_FEAS) && feas) {
      for (int i = 0; i < pp_n_; ++i) {
        feas[i] = 0;
        if (!sets_[i].owned && !all) continue;
        const double* h = hres_ + (size_t)i * SLOTS;
        feas[i] = (sets_[i].two_pass || sets_[i].ext_kind) ? (double)feas_value(h[SL_FE2], h[SL_SS2])
                                                            : (double)feas_value(h[SL_FE], h[SL_SS]);
      }
    }
  }

  void log_scalars(double* obj, double* evol_x) override {
    need_final();
    ObserverGuard og(observer());
    if (!have_log_sums_) {
      K<T>::log3(stream_, G_.N, mk_ ? w_ : x_, m_, xold_, part_sets_);
      if (mk_) K<T>::log3(stream_, Nx_, x_, (const T*)nullptr, xold_, part_sets_ + (size_t)SLOTS * NB);
      K<T>::fin_sum(stream_, part_sets_, (mk_ ? 2 : 1) * SLOTS, nullptr, hres_);
      SIPX_HIP(hipStreamSynchronize(stream_));
      obj_ss_ = hres_[SL_OBJ]; evo_ss_ = hres_[SL_EVO]; xx_ss_ = hres_[SL_XX];
      if (mk_) { evo_ss_ = hres_[SLOTS + SL_EVO]; xx_ss_ = hres_[SLOTS + SL_XX]; }
    }
    const T nd = (T)std::sqrt(obj_ss_);
    *obj = (double)(T(0.5) * (nd * nd));                                   // PARSDMM.jl:140
    *evol_x = (double)((T)std::sqrt(evo_ss_) / (T)std::sqrt(xx_ss_));      // PARSDMM.jl:145
  }

  void adapt_rho_gamma(int adjust_rho, int adjust_gamma, double* rho_io, double* gamma_io) override {
    need_final();
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>& s = sets_[i];
      if (!s.owned && !comm_) continue;
      if (!s.bb_valid)
        throw std::runtime_error("sipx_adapt_rho_gamma: call sipx_update_y_l with SIPX_YL_BB (or _FIRST) first");
      T rho = (T)rho_io[i], gamma = (T)gamma_io[i];
      const double* h = s.sums;
      bb_rule<T>((T)h[SL_HL], (T)std::sqrt(h[SL_HH]), (T)std::sqrt(h[SL_LH]), (T)std::sqrt(h[SL_DL]),
                 (T)std::sqrt(h[SL_GG]), (T)h[SL_GL], adjust_rho != 0, adjust_gamma != 0, rho, gamma);
      rho_io[i] = (double)rho;
      gamma_io[i] = (double)gamma;
    }
  }

  void q_update(const double* rho_new, const double* rho_old) override {
    need_final();
    ObserverGuard og(observer());
    if (stencil_q_) {
      stencil_weights(rho_new);
      return;
    }
    if (mk_) {
      std::vector<T> al(p_n_);
      for (int i = 0; i < p_n_; ++i) al[i] = rho_new[i] == rho_old[i] ? T(0) : (T)rho_new[i] - (T)rho_old[i];
      mk_q_update(al);
      return;
    }
    flush_q_pending();
    QArgs<T> a;
    a.nsets = 0;
    for (int i = 0; i < p_n_; ++i) {
      if (rho_new[i] == rho_old[i]) continue;                       // ind_updated, PARSDMM.jl:230
      push_qset(a, sets_[i], (T)rho_new[i] - (T)rho_old[i]);        // Q_update!.jl:47
    }
    if (q_defer_ && q_fused_ && cds_.march != 0 && !comm_ && a.nsets > 0) {
      bool generated = true;
      for (int k = 0; k < a.nsets; ++k) generated &= a.s[k].ata == nullptr;
      if (generated) {                       // applied by the residual product of the coming x-step
        q_pending_args_ = a;
        q_pending_ = true;
        return;
      }
    }
    K<T>::q_update(stream_, G_, qr0_, qr1_, cds_, a, Q_);
  }
  void flush_q_pending() {
    if (!q_pending_) return;
    q_pending_ = false;
    K<T>::q_update(stream_, G_, qr0_, qr1_, cds_, q_pending_args_, Q_);
  }

  // Warm start of this (finer) level from a solved coarser one, device to device: x by nearest-neighbour resampling of the
  // grid (PARSDMM_multi_level.jl:61-67), l_i and y_i set by set as interpolate_y_l does (interpolate_y_l.jl:16-94) -- a
  // single-block operator resamples its TD_n-shaped array; TV splits its rows into chunks of the sizes of [D_x; D_y; D_z]
  // blocks (although the storage is [D_z; D_y; D_x]: replicated as written) and resamples each.  The reference's row
  // order is recovered from / restored to the padded layout by the pack / unpack kernels of the import / export path.
  void warm_start_from(EngineBase* coarse_base) override {
    need_final();
    auto* c = dynamic_cast<Engine<T>*>(coarse_base);
    if (!c || !c->finalized_) throw std::runtime_error("warm start: the coarse context must be a finalized context of the same precision");
    if (c->device_ != device_) throw std::runtime_error("warm start: both levels must live on one device");
    if (c->p_n_ != p_n_ || c->ndim_ != ndim_ || mk_ || c->mk_) throw std::runtime_error("warm start: the two levels must hold the same sets");
    if ((comm_ != nullptr) != (c->comm_ != nullptr) || (comm_ && !(slab_ && c->slab_)))
      throw std::runtime_error("warm start between levels of a sharded solve needs both levels slab-decomposed (sipx_set_decomp)");
    // Block by block (x, then l_i and y_i of every set): the coarse block is completed in ONE whole-size temporary of the COARSE
    // level -- an all-gather of the coarse ranks' slabs where that level is slab-decomposed (a collective: every rank makes the same
    // calls); one rank: the block itself -- and resampled from there, padded layout to padded layout (k_resample_padded: entry by entry
    // the copy the reference makes on the chunks in row order, interpolate_y_l.jl:16-94), into the grid points THIS rank stores of
    // the fine block: its planes and the halo planes around them where the fine level holds sparse arrays (round 5: the levels of a
    // slab-decomposed multilevel solve no longer need whole arrays, PARSDMM_multi_level.jl:56-83), all of it otherwise.  A coarse
    // array is 1 / 8 of a fine one: the temporary is the size of a rank's slab of the fine grid on eight GPUs.
    long long nc[3], nf[3];
    for (int a = 0; a < 3; ++a) { nc[a] = c->G_.n[a]; nf[a] = G_.n[a]; }
    const long long Nc = c->G_.N, Nf = G_.N;
    const long long cpad = c->comm_ ? c->chunk_ * c->comm_->world : Nc;
    int nbmax = 1;
    for (auto& st : c->sets_) nbmax = std::max(nbmax, st.nblk_or1());
    T* whole = c->slab_ ? dalloc<T>((size_t)nbmax * cpad, false) : nullptr;
    const int dt = dtype_code();
    const long long f0 = slab_local_ ? std::max<long long>(0, wlo_) : 0, f1 = slab_local_ ? std::min<long long>(Nf, whi_) : Nf;
    // (all blocks of a set's coarse vector side by side: a chunk of the reference's row vector may straddle two of them)
    auto complete = [&](const T* src, int nb) -> const T* {
      if (!c->slab_) return src;             // one rank: the blocks themselves (stride Nc)
      const long long nloc = c->r1_ - c->r0_;
      for (int q = 0; q < nb; ++q) {          // the coarse ranks' planes -> the whole block
        T* w = whole + (long long)q * cpad;
        if (nloc > 0) SIPX_HIP(hipMemcpyAsync(w + c->r0_, src + (long long)q * Nc + c->r0_, nloc * sizeof(T), hipMemcpyDeviceToDevice, stream_));
        c->comm_->allgather(w, (size_t)c->chunk_, dt, stream_);
      }
      return whole;
    };
    const long long cstride = c->slab_ ? cpad : Nc;
    SIPX_HIP(hipStreamSynchronize(c->stream_));
    resample_nn_padded<T>(stream_, nc, nf, nc, nf, f0, f1, complete(c->x_, 1), x_);
    for (int i = 0; i < p_n_; ++i) {
      SetState<T>&f = sets_[i], &g = c->sets_[i];
      if (f.custom || g.custom || f.nblk != g.nblk) throw std::runtime_error("warm start: operator kinds differ between the levels");
      for (int q = 0; q < f.nblk; ++q)
        if (f.dir[q] != g.dir[q]) throw std::runtime_error("warm start: operator kinds differ between the levels");
      if (!f.owned || !g.owned) continue;
      for (int which = 0; which < 2; ++which) {
        const T* src = complete(which ? g.y : g.l, g.nblk_or1());
        T* dst = which ? f.y : f.l;
        if (f.nblk >= 2) {       // TV / D2D / D3D: the reference's D_x-, D_y-, D_z-sized chunks of the row vector (interpolate_y_l.jl:21-30,53-57)
          for (int q = 0; q < f.nblk; ++q)
            resample_nn_rows<T>(stream_, nc, nf, f.nblk, f.dir, q, cstride, f0, f1, src, dst + (long long)q * Nf);
        } else {                 // one block: the grid, minus one along the direction of its difference operator (TD_n, get_TD_operator.jl)
          long long cc[3], cf[3];
          for (int a = 0; a < 3; ++a) { cc[a] = nc[a]; cf[a] = nf[a]; }
          if (f.nblk == 1) { cc[f.dir[0]] -= 1; cf[f.dir[0]] -= 1; }
          resample_nn_padded<T>(stream_, nc, nf, cc, cf, f0, f1, src, dst);
        }
      }
    }
    SIPX_HIP(hipStreamSynchronize(stream_));
    dfree(whole);
  }

  // Slab-decomposed context: completes x and / or the y_i, l_i named on every rank from the ranks' slabs (all-gathers on the
  // engine stream; afterwards the arrays are whole and identical everywhere).  A collective.
  void gather_slabs(bool want_x, const std::vector<char>& want_l, const std::vector<char>& want_y) {
    const int dt = dtype_code();
    if (want_x) comm_->allgather(x_, (size_t)chunk_, dt, stream_);
    for (int i = 0; i < p_n_; ++i)
      for (int which = 0; which < 2; ++which) {
        if (!(which ? want_y[i] : want_l[i])) continue;
        T* arr = which ? sets_[i].y : sets_[i].l;
        const int nb = sets_[i].nblk > 0 ? sets_[i].nblk : 1;
        for (int q = 0; q < nb; ++q) {
          T* blk = arr + (long long)q * G_.N;
          if (r1_ > r0_) SIPX_HIP(hipMemcpyAsync(scr_v_ + r0_, blk + r0_, (r1_ - r0_) * sizeof(T), hipMemcpyDeviceToDevice, stream_));
          comm_->allgather(scr_v_, (size_t)chunk_, dt, stream_);
          SIPX_HIP(hipMemcpyAsync(blk, scr_v_, G_.N * sizeof(T), hipMemcpyDeviceToDevice, stream_));
        }
      }
  }

  // Sparse arrays: nothing whole exists on a rank.  x and the requested y_i, l_i are completed in whole-size TEMPORARIES (the
  // exchange buffer of one N-vector, and the blocks of one set strung together) and go to the host from there.  A collective.
  void download_sparse(void* x, void* const* l, void* const* y) {
    const int dt = dtype_code();
    const long long N = G_.N, Npad = chunk_ * comm_->world, nloc = r1_ - r0_;
    int nbmax = 1;
    for (auto& s : sets_) nbmax = std::max(nbmax, s.nblk_or1());
    T* exch = dalloc<T>((size_t)Npad);
    T* whole = dalloc<T>((size_t)nbmax * N);
    auto complete = [&](const T* src) {      // src: a block of a sparse array (global indexing); afterwards exch[0, N) holds all of it
      if (nloc > 0) SIPX_HIP(hipMemcpyAsync(exch + r0_, src + r0_, nloc * sizeof(T), hipMemcpyDeviceToDevice, stream_));
      comm_->allgather(exch, (size_t)chunk_, dt, stream_);
    };
    if (x) {
      complete(x_);
      SIPX_HIP(hipStreamSynchronize(stream_));
      host_prefault(x, N * sizeof(T));
      SIPX_HIP(hipMemcpy(x, exch, N * sizeof(T), hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < p_n_; ++i)
      for (int which = 0; which < 2; ++which) {
        void* const* dst = which ? y : l;
        if (!(dst && dst[i])) continue;
        const T* arr = which ? sets_[i].y : sets_[i].l;
        for (int q = 0; q < sets_[i].nblk_or1(); ++q) {
          complete(arr + (long long)q * N);
          SIPX_HIP(hipMemcpyAsync(whole + (long long)q * N, exch, N * sizeof(T), hipMemcpyDeviceToDevice, stream_));
        }
        download_rows(sets_[i], whole, (T*)dst[i]);
      }
    SIPX_HIP(hipStreamSynchronize(stream_));
    dfree(exch);
    dfree(whole);
  }

  void download(void* x, void* const* l, void* const* y) override {
    need_final();
    if (slab_local_) { download_sparse(x, l, y); return; }
    if (slab_) {       // every rank holds its planes only: complete x and the requested y, l (a collective: every rank calls it)
      std::vector<char> wl(p_n_, 0), wy(p_n_, 0);
      for (int i = 0; i < p_n_; ++i) { wl[i] = l && l[i]; wy[i] = y && y[i]; }
      gather_slabs(x != nullptr, wl, wy);
    }
    SIPX_HIP(hipStreamSynchronize(stream_));
    if (x) {
      host_prefault(x, Nx_ * sizeof(T));
      SIPX_HIP(hipMemcpy(x, x_, Nx_ * sizeof(T), hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < p_n_; ++i) {
      if (!sets_[i].owned) continue;
      if (l && l[i]) download_rows(sets_[i], sets_[i].l, (T*)l[i]);
      if (y && y[i]) download_rows(sets_[i], sets_[i].y, (T*)y[i]);
    }
  }

  // ------------------------------------------------------------------------------------------
  // Whole solve: src/PARSDMM.jl:63-257 restated (serial path).  begin / step so a caller can advance the
  // solve iteration by iteration (bench warm-up + timed steps); sipx_parsdmm = begin + steps until done.
  void parsdmm_begin(const sipx_options* opt, sipx_log* log) override {
    need_final();
    for (auto& s : sets_)
      if (!s.owned && !comm_) throw std::runtime_error("sipx_parsdmm needs every set local, or a communicator (sipx_set_comm*)");
    Run& R = run_;
    R = Run();
    head_done_ = false;
    for (bool& v : open_valid_) v = false;
    mark_weight_[0] = mark_weight_[1] = 1.0;
    R.log = log;
    R.maxit = opt->maxit;
    const int p = p_n_, pp = pp_n_;
    R.evol_rel_tol = (T)opt->evol_rel_tol; R.feas_tol = (T)opt->feas_tol; R.obj_tol = (T)opt->obj_tol;   // convert_options!
    R.adjust_rho = opt->adjust_rho; R.adjust_gamma = opt->adjust_gamma; R.adjust_feas_rho = opt->adjust_feasibility_rho;
    R.freq = opt->rho_update_frequency;
    if (any_ncvx_) { R.freq = 3; R.adjust_gamma = false; }            // PARSDMM_initialize.jl:107-114
    std::fill(log->timing_ms, log->timing_ms + 7, 0.0);
    log->stopped_feasible = 0;
    for (int i = 0; i < pp; ++i) log->set_feasibility[i] = feas_init_[i];   // :236
    const double maxf = julia_maximum(feas_init_.begin(), feas_init_.end());
    R.active = true;
    if (pp > 0 && maxf < (double)R.feas_tol) {                        // :101-104, PARSDMM.jl:63-82
      {
        const long long c0 = slab_local_ ? std::max<long long>(0, wlo_) : 0, c1 = slab_local_ ? std::min<long long>(G_.N, whi_) : G_.N;
        if (c1 > c0) SIPX_HIP(hipMemcpyAsync(x_ + c0, m_ + c0, (c1 - c0) * sizeof(T), hipMemcpyDeviceToDevice, stream_));
      }
      if (mk_) SIPX_HIP(hipMemsetAsync(x_ + G_.N, 0, G_.N * sizeof(T), stream_));      // x = [m; 0]  PARSDMM.jl:64-69
      SIPX_HIP(hipStreamSynchronize(stream_));
      log->n_iter = 1;
      log->n_feas_rows = 1;
      log->stopped_feasible = 1;
      R.done = true;
      return;
    }
    R.counter = 2;
    R.ind_ref = R.maxit;
    R.tol_ref = 1.0;
    R.rho.resize(p); R.gamma.resize(p); R.rho_new.resize(p); R.rpri.resize(p); R.rdual.resize(p);
    R.feas.resize(std::max(pp, 1));
    for (int i = 0; i < p; ++i) { R.rho[i] = (double)rho_[i]; R.gamma[i] = (double)gamma_[i]; }
    log->n_iter = R.maxit;
    log->n_feas_rows = R.counter;
    if (R.maxit < 1) R.done = true;
  }

  bool parsdmm_step() override {
    Run& R = run_;
    if (!R.active) throw std::runtime_error("sipx_parsdmm_steps: call sipx_parsdmm_begin first");
    if (R.done) return true;
    ObserverGuard og(observer());
    sipx_log* log = R.log;
    const int p = p_n_, pp = pp_n_, maxit = R.maxit;
    std::vector<double>&rho = R.rho, &gamma = R.gamma, &rho_new = R.rho_new, &rpri = R.rpri, &rdual = R.rdual, &feas = R.feas;
    int& counter = R.counter;
    const int i = ++R.i;
    const int par = i & 1;
    resolve_timing(log, par);            // marks recorded two steps ago have long completed
    // Grids of up to 2^23 points on one rank: the marks are recorded on iterations 1-4 and on every seventh after them (coprime
    // with the usual rho_update_frequency 2 / 3 and with the feasibility estimate's 10, so that the sample holds plain,
    // Barzilai-Borwein and feasibility iterations in their proportions), what they measure stands for the iterations since the
    // last such one, and the sums of the y/l update are awaited on a pinned word instead of the record behind them (k_fin_sum).
    // A record costs the stream up to 6 us -- two on a plain iteration, five on one that may change rho: 2048^2 2277 -> 2352 it/s,
    // the buckets then estimates (within 12 % of the every-iteration figures over 35 iterations); at 256^3 the same buys 0.4 % and
    // the buckets stay exact.  SIPX_MARK_STRIDE: 1 = every iteration, k = every k-th, whatever the grid.
    // (sharded: every iteration, unless SIPX_MARK_STRIDE says otherwise -- a rank's share of the headline on eight GPUs through RCCL
    //  with a world of one ran at 2215 it/s with the marks sampled and at 2219 without: its collectives leave gaps the records hide
    //  in; the sums then keep an event of their own, ev_sums_, where the marks are left out)
    // (round 5: every iteration by default at every size -- log.timing is a public field and means the same thing for every
    //  caller; the sampling that round 4 switched on by itself for grids of up to 2^23 points is opt-in: SIPX_MARK_STRIDE=7)
    const int stride = env_knobs().mark_stride > 0 ? env_knobs().mark_stride : 1;
    auto is_timed = [&](int it) { return stride <= 1 || it <= 4 || it % stride == 0; };
    const bool timed = is_timed(i), next_timed = i < maxit && is_timed(i + 1);
    mark_step_[par] = i;
    if (timed) {
      mark_weight_[par] = (double)(i - R.last_timed);
      R.last_timed = i;
    }
    word_sums_ = stride > 1 && !comm_;
    struct WordSumsOff { bool& f; ~WordSumsOff() { f = false; } } word_sums_off{word_sums_};      // (the phase entry points keep the event)
    // Section timing: ONE chain of marks on the engine stream (a record costs the stream about 5 us); the time between two
    // consecutive marks goes to the section named at the later one (-1: a mark that only opens an interval), four marks per
    // step in the common path.  The two host-only sections (stop rule, rho / gamma rules) use the host clock.
    auto mark = [&](int section) {
      if (!timed || nmark_[par] >= MAXMARK) return;
      SIPX_HIP(hipEventRecord(ev_[par * MAXMARK + nmark_[par]], stream_));
      mark_sec_[par][nmark_[par]++] = section;
    };
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    // can the rules at the end of iteration `it` change rho?  If not, the right-hand side of iteration it+1 is known as soon
    // as the y/l kernels of `it` are queued
    auto rho_may_change = [&](int it) {
      for (int k = 0; k < p; ++k)          // the clamp to [1e-2, 1e4] runs every iteration (PARSDMM.jl:226): a rho_ini outside it changes at once
        if ((T)rho[k] != std::max(std::min((T)rho[k], T(1e4)), T(1e-2))) return true;
      return ((R.adjust_rho || R.adjust_gamma) && it % R.freq == 0) || (R.adjust_feas_rho && it % 10 == 0 && it > 10 && pp > 0);
    };
    {
      // (the x-step's section opens where its residual product is queued: here, or in the step before when that was queued ahead)
      if (!head_done_ && timed) open_section(i);
      if (!R.rhs_ready) {
        rhs_compose(rho.data());
        mark(1);
      }
      R.rhs_ready = false;
      int64_t cg_it; double relres; int flag;
      argmin_x(i, &R.tol_ref, &cg_it, &relres, &flag);
      log->cg_it[i - 1] = cg_it;
      log->cg_relres[i - 1] = relres;
      mark(2);
      int flags = 0;
      if (i % 10 == 0) flags |= SIPX_YL_FEAS;
      if (i == 1) flags |= SIPX_YL_FIRST;
      if ((R.adjust_rho || R.adjust_gamma) && i % R.freq == 0) flags |= SIPX_YL_BB;
      defer_sums_ = true;                  // queue the kernels and the reduction of their sums, collect them further down
      const bool rhs_known = i < maxit && !rho_may_change(i);
      fuse_rhs_ = rhs_known;               // rhs_{i+1} may be formed by the sweep that forms y_{i+1}, l_{i+1}
      merge_sums_ = rhs_known && resid_ahead_ && comm_ != nullptr;
      update_y_l(i, flags, rho.data(), gamma.data(), rpri.data(), rdual.data(), feas.data());
      fuse_rhs_ = false;
      defer_sums_ = false;
      merge_sums_ = false;
      mark(3);                             // also the event the host waits on for the sums (unless they come with the pinned word)
      if (timed && nmark_[par] > 0) sums_event_ = ev_[par * MAXMARK + nmark_[par] - 1];
      else if (!word_sums_) {              // (a step without marks whose sums do not come with the pinned word: sharded)
        SIPX_HIP(hipEventRecord(ev_sums_, stream_));
        sums_event_ = ev_sums_;
      }
      // Software pipeline: nothing the GPU is given next may depend on the sums the host is about to read.  When the rules
      // below cannot touch rho, rhs_{i+1} = sum_i A_i'(rho_i y_i + l_i) (and, sharded, its reduce-scatter on the
      // communication stream) is queued now and runs while the host waits for the sums and evaluates the stop rule.  A
      // stop leaves x, y, l as they are: rhs is scratch.
      if (rhs_fused_) {
        R.rhs_ready = true;                  // written by the y/l sweep itself (kernels_multi.hip)
      } else if (rhs_known) {
        rhs_compose(rho.data());
        mark(1);
        R.rhs_ready = true;
      }
      // (round 3 had the product store x_old <- x, which a context without a distance term -- feasibility only -- still had to read
      //  for evol_x after this point: its logged evol_x was zero.  The product no longer touches x_old: x_k stays behind in its own
      //  ring buffer when the x-step moves on, tests/test_gpu_round4.py::test_feasibility_only_logs_evol_x...)
      if (R.rhs_ready && resid_ahead_) {     // ... and so is the residual product of the coming x-step
        if (next_timed) open_section(i + 1);         // (the coming step is a timed one: its x-step section opens here)
        argmin_x_head();
        if (comm_) {                         // (sharded: the sums arrive with the grouped call of the head)
          // (an opening mark: the head's time belongs to the coming step's x-step section, which opened in front of it)
          mark(-1);
          if (timed && nmark_[par] > 0) sums_event_ = ev_[par * MAXMARK + nmark_[par] - 1];
          else {
            SIPX_HIP(hipEventRecord(ev_sums_, stream_));
            sums_event_ = ev_sums_;
          }
        }
      }
      collect_set_sums(rho.data(), rpri.data(), rdual.data(), feas.data());
      auto t_host = clk::now();
      T sd = (T)rdual[0], sp = (T)rpri[0];
      for (int k = 0; k < p; ++k) {
        log->r_pri[(size_t)(i - 1) * p + k] = rpri[k];
        log->r_dual[(size_t)(i - 1) * p + k] = rdual[k];
        if (k > 0) { sd = sd + (T)rdual[k]; sp = sp + (T)rpri[k]; }
        log->rho[(size_t)(i - 1) * p + k] = rho[k];
        log->gamma[(size_t)(i - 1) * p + k] = gamma[k];
      }
      log->r_dual_total[i - 1] = (double)sd;                          // PARSDMM.jl:134
      log->r_pri_total[i - 1] = (double)sp;                           // :138
      if (i % 10 == 0) {                                              // update_y_l.jl:90-105
        for (int k = 0; k < pp; ++k) log->set_feasibility[(size_t)(counter - 1) * pp + k] = feas[k];
        counter += 1;
      }
      log_scalars(&log->obj[i - 1], &log->evol_x[i - 1]);
      log->timing_ms[3] += ms_since(t_host);                          // bookkeeping of the y/l section
      t_host = clk::now();
      // ---- stop_PARSDMM.jl:23-52 ----
      bool stop = false;
      if (i > 6 && pp > 0) {
        const double* row = log->set_feasibility + (size_t)(counter - 2) * pp;
        if (julia_maximum(row, row + pp) < (double)R.feas_tol) {
          double mx = -INFINITY; bool nan = false;
          for (int k = i - 6; k < i; ++k) {
            const T a = (T)log->obj[k], b = (T)log->obj[k - 1];
            const T v = std::fabs((a - b) / b);
            if (std::isnan(v)) nan = true;
            mx = std::max(mx, (double)v);
          }
          if (!nan && mx < (double)R.obj_tol) stop = true;
        }
      }
      if (i > 5 && julia_maximum(log->evol_x + (i - 6), log->evol_x + i) < (double)R.evol_rel_tol) stop = true;
      if (i > 20 && R.adjust_rho) {
        const int lo = std::max(i - 50, 1);
        if (log->r_pri_total[i - 1] > julia_maximum(log->r_pri_total + (lo - 1), log->r_pri_total + (i - 1))) {
          R.adjust_rho = R.adjust_feas_rho = R.adjust_gamma = false;
          R.ind_ref = i;
        }
      }
      if (!R.adjust_rho && i > R.ind_ref + 25) {
        const int lo = std::max(R.ind_ref, std::max(i - 50, 1));
        if (log->r_pri_total[i - 1] > julia_maximum(log->r_pri_total + (lo - 1), log->r_pri_total + (i - 1))) stop = true;
      }
      log->timing_ms[4] += ms_since(t_host);
      if (stop) {
        finish_solve(log, i, counter, rho, gamma);
        return true;
      }
      t_host = clk::now();
      // ---- adjust rho and gamma (PARSDMM.jl:163-227); l_hat / snapshots were fused into update_y_l ----
      rho_new = rho;
      if ((R.adjust_rho || R.adjust_gamma) && i % R.freq == 0)
        adapt_rho_gamma(R.adjust_rho, R.adjust_gamma, rho_new.data(), gamma.data());
      if (R.adjust_feas_rho && i % 10 == 0 && i > 10 && pp > 0) {       // :213-223
        const double* row = log->set_feasibility + (size_t)(counter - 2) * pp;
        int arg = 0;
        bool found_nan = false;
        for (int k = 0; k < pp && !found_nan; ++k) {
          if (std::isnan(row[k])) { arg = k; found_nan = true; }
          else if (row[k] > row[arg]) arg = k;
        }
        rho_new[arg] = (double)(T(2.0) * (T)rho_new[arg]);
      }
      for (int k = 0; k < p; ++k)                                      // :226
        rho_new[k] = (double)std::max(std::min((T)rho_new[k], T(1e4)), T(1e-2));
      log->timing_ms[5] += ms_since(t_host);
      if (R.rhs_ready && rho_new != rho) throw std::runtime_error("internal: rho changed under a right-hand side queued ahead");
      bool changed = false;
      for (int k = 0; k < p; ++k) changed |= rho_new[k] != rho[k];
      if ((i < maxit && !R.rhs_ready) || changed) mark(-1);           // the stream sat idle while the host decided
      if (i < maxit && !R.rhs_ready) {     // rho is final now; queued ahead of the Q update so that, sharded, the
        rhs_compose(rho_new.data());       // reduce-scatter (communication stream) runs beside it
        mark(1);
        R.rhs_ready = true;
      }
      if (changed) {
        q_defer_ = i < maxit;               // (the last iteration's update is applied at once: the context keeps a current Q)
        q_update(rho_new.data(), rho.data());                          // :230-243
        q_defer_ = false;
        if (!q_pending_) mark(6);
      }
      rho = rho_new;
    }
    if (i == maxit) finish_solve(log, maxit, counter, rho, gamma);
    return R.done;
  }

  // common exit of the whole solve (stop rule or maxit): timing resolved, logs truncated, and the context left so that a
  // later solve on it continues from here -- Q holds sum_i rho_i AtA_i for exactly these rho
  void finish_solve(sipx_log* log, int n_iter, int counter, const std::vector<double>& rho, const std::vector<double>& gamma) {
    if (rs_pending_) {                      // a right-hand side queued ahead: let its exchange finish before anyone touches rhs
      SIPX_HIP(hipStreamWaitEvent(stream_, ev_c_[1], 0));
      rs_pending_ = false;
    }
    flush_q_pending();
    resolve_timing(log, 0);
    resolve_timing(log, 1);
    log->n_iter = n_iter;
    log->n_feas_rows = counter;
    for (int k = 0; k < p_n_; ++k) { rho_[k] = (T)rho[k]; gamma_[k] = (T)gamma[k]; }
    run_.rhs_ready = false;
    run_.done = true;
    head_done_ = false;
  }

  void parsdmm(const sipx_options* opt, sipx_log* log) override {
    parsdmm_begin(opt, log);
    while (!parsdmm_step()) {
    }
  }

  // ------------------------------------------------------------------------------------------
  void apply_op(int op, const void* x, void* out, bool adjoint) override {
    SIPX_HIP(hipSetDevice(device_));
    SetState<T> s;
    configure_op(s, op);
    long long H = 4;
    for (int a = 0; a < 3; ++a) H = std::max<long long>(H, G_.st[a]);
    H = (H + 3) / 4 * 4;
    T* dxb = dalloc<T>(G_.N + 2 * H);
    T* dvb = dalloc<T>(s.Mpad + H);
    T *dx = dxb + H, *dv = dvb + H;
    if (!adjoint) {
      SIPX_HIP(hipMemcpy(dx, x, G_.N * sizeof(T), hipMemcpyHostToDevice));
      K<T>::fwd(stream_, G_, s.nblk, s.dir, s.ih, dx, dv);
      download_rows(s, dv, (T*)out);
    } else {
      upload_rows(s, (const T*)x, dv);
      K<T>::adj(stream_, G_, s.nblk, s.dir, s.ih, dv, dx);
      SIPX_HIP(hipStreamSynchronize(stream_));
      SIPX_HIP(hipMemcpy(out, dx, G_.N * sizeof(T), hipMemcpyDeviceToHost));
    }
    dfree(dxb);
    dfree(dvb);
  }

  void project(const sipx_set_desc* d, void* v, int64_t len) override {
    SIPX_HIP(hipSetDevice(device_));
    Grid g1;
    g1.n[0] = len; g1.n[1] = 1; g1.n[2] = 1; g1.N = len; g1.st[0] = 1; g1.st[1] = len; g1.st[2] = len;
    T* dv = dalloc<T>(len);
    T* dc = dalloc<T>(len);
    T *lb = nullptr, *ub = nullptr;
    double* part = dalloc<double>((size_t)(PREP_SLOTS + 2) * NB);
    T* mp = dalloc<T>(2 * NB);
    ProjScalars<T>* ps = dalloc<ProjScalars<T>>(1);
    SIPX_HIP(hipMemcpy(dv, v, len * sizeof(T), hipMemcpyHostToDevice));
    const int prox = d->proj;
    const T plo = (T)d->pmin, phi = (T)d->pmax;
    if (prox == SIPX_PROJ_BOUNDS_VEC && d->mode == SIPX_MODE_WHOLE && d->transform == SIPX_TRANSFORM_NONE) {
      if (!d->lb || !d->ub) throw std::runtime_error("per-element bounds need lb and ub");
      lb = dalloc<T>(len); ub = dalloc<T>(len);
      SIPX_HIP(hipMemcpy(lb, d->lb, len * sizeof(T), hipMemcpyHostToDevice));
      SIPX_HIP(hipMemcpy(ub, d->ub, len * sizeof(T), hipMemcpyHostToDevice));
    }
    if (prox == SIPX_PROJ_L1 && !(d->pmax > 0)) throw std::runtime_error("Radius of L1 ball is negative");
    if (d->mode != SIPX_MODE_WHOLE || d->transform != SIPX_TRANSFORM_NONE || prox == SIPX_PROJ_L1_DFT || prox >= SIPX_PROJ_RANK) {   // incl. SIPX_PROJ_BOUNDS_DFT   // acts on the context grid
      if (len != G_.N) throw std::runtime_error("this projector / application mode needs a vector of the grid size");
      SetState<T> st;
      configure_op(st, SIPX_OP_IDENTITY);
      configure_proj(st, d);
      if (st.ext_kind) {
        st.spec.lb = st.host_lb.empty() ? nullptr : st.host_lb.data();
        st.spec.ub = st.host_ub.empty() ? nullptr : st.host_ub.data();
        st.spec.basis = st.host_basis.empty() ? nullptr : st.host_basis.data();
        ExtProj<T> ext(st.spec, stream_);
        ext.project(dv, false, part, mp, dc);
      } else {                                   // per-fiber bounds, expanded by configure_proj
        lb = dalloc<T>(len); ub = dalloc<T>(len);
        SIPX_HIP(hipMemcpy(lb, st.host_lb.data(), len * sizeof(T), hipMemcpyHostToDevice));
        SIPX_HIP(hipMemcpy(ub, st.host_ub.data(), len * sizeof(T), hipMemcpyHostToDevice));
        proj_apply_grid<T>(stream_, g1, 0, nullptr, len, dv, st.prox, st.plo, st.phi, lb, ub, nullptr);
      }
      SIPX_HIP(hipStreamSynchronize(stream_));
      SIPX_HIP(hipMemcpy(v, dv, len * sizeof(T), hipMemcpyDeviceToHost));
      for (void* q : {(void*)dv, (void*)dc, (void*)part, (void*)mp, (void*)ps, (void*)lb, (void*)ub})
        if (q) dfree(q);
      return;
    }
    const bool two = prox == SIPX_PROJ_L1 || prox == SIPX_PROJ_L2 || prox == SIPX_PROJ_ANNULUS || prox == SIPX_PROJ_CARDINALITY;
    long long* di = prox == SIPX_PROJ_CARDINALITY ? dalloc<long long>(len) : nullptr;
    if (two) {
      K<T>::ps_init(stream_, ps, di);
      K<T>::proj_scalars_arr(stream_, len, dv, prox, plo, phi, ps, part, mp, dc, len);
    }
    proj_apply_grid<T>(stream_, g1, 0, nullptr, len, dv, prox, prox == SIPX_PROJ_BOUNDS_VEC ? T(0) : plo, phi, lb, ub, two ? ps : nullptr);
    SIPX_HIP(hipStreamSynchronize(stream_));
    SIPX_HIP(hipMemcpy(v, dv, len * sizeof(T), hipMemcpyDeviceToHost));
    for (void* q : {(void*)dv, (void*)dc, (void*)lb, (void*)ub, (void*)part, (void*)mp, (void*)ps, (void*)di})
      dfree(q);
  }

  void get_Q(void* Q, int64_t* offsets, int* d) override {
    need_final();
    SIPX_HIP(hipStreamSynchronize(stream_));
    if (d) *d = cds_.d;
    if (offsets) for (int b = 0; b < cds_.d; ++b) offsets[b] = cds_.off[b];
    if (Q && stencil_q_) throw std::runtime_error("stencil Q mode stores no bands (use sipx_apply_Q)");
    if (Q && comm_ && comm_->world > 1) throw std::runtime_error("a sharded context maintains its slab of Q only");
    flush_q_pending();
    if (Q && cds_.sym) {      // the negative bands are not maintained while solving: rebuild them from their partners
      K<T>::mirror_bands(stream_, Nx_, cds_, Q_);
      SIPX_HIP(hipStreamSynchronize(stream_));
    }
    if (Q) SIPX_HIP(hipMemcpy(Q, Q_, (size_t)Nx_ * cds_.d * sizeof(T), hipMemcpyDeviceToHost));
  }

  void apply_Q(const void* x, void* y) override {     // y = Q x through the solver's own kernel (either mode)
    need_final();
    if (comm_ && comm_->world > 1) throw std::runtime_error("a sharded context maintains its slab of Q only");
    SIPX_HIP(hipStreamSynchronize(stream_));
    SIPX_HIP(hipMemcpy(p_, x, Nx_ * sizeof(T), hipMemcpyHostToDevice));
    flush_q_pending();
    if (stencil_q_) K<T>::sq_spmv(stream_, G_, sq_, p_, Ap_);
    else K<T>::spmv(stream_, G_, Nx_, Q_, cds_, p_, Ap_);
    SIPX_HIP(hipStreamSynchronize(stream_));
    SIPX_HIP(hipMemcpy(y, Ap_, Nx_ * sizeof(T), hipMemcpyDeviceToHost));
  }

  double time_spmv(int reps) override {
    need_final();
    flush_q_pending();
    hipEvent_t a, b;
    SIPX_HIP(hipEventCreate(&a));
    SIPX_HIP(hipEventCreate(&b));
    auto one = [&]() {
      if (stencil_q_) K<T>::sq_spmv(stream_, G_, sq_, x_, Ap_);
      else K<T>::spmv(stream_, G_, Nx_, Q_, cds_, x_, Ap_);
    };
    one();   // warm
    SIPX_HIP(hipEventRecord(a, stream_));
    for (int k = 0; k < reps; ++k) one();
    SIPX_HIP(hipEventRecord(b, stream_));
    SIPX_HIP(hipEventSynchronize(b));
    float ms = 0;
    SIPX_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return (double)ms / reps;
  }

  // Per-kernel statistics.  mode 0: off; 1: the product of the CG iteration only (k_cds<MODE=1> / k_sq<MODE=1>: what the
  // bench line's `roofline` is measured on, inside the timed region -- two event records per CG iteration); 2: EVERY kernel
  // the engine launches (two records around each: about 5 us per launch, so this mode is for a window of its own, not for
  // the timed region).  Samples are bracketed on the stream the kernel is launched on.
  void kernel_stats(int enable, int64_t* launches, double* total_ms) override {
    need_final();
    std::vector<KAgg> agg = aggregate_samples();
    const KAgg& a = stencil_q_ ? agg[KID_SQ_DOT] : agg[KID_CDS_DOT];
    if (launches) *launches = a.launches;
    if (total_ms) *total_ms = a.ms;
    start_stats(enable);
  }
  // enable = -1: the counters only (slab_searches, rank_route) -- no synchronisation, the collection and its samples stay as they are
  const char* kernel_stats_json(int enable) override {
    need_final();
    const bool peek = enable < 0;
    std::vector<KAgg> agg = peek ? std::vector<KAgg>(KID_COUNT) : aggregate_samples();
    std::string& o = stats_json_;
    o = "{\"mode\": " + std::to_string(stats_mode_) + ", \"event_pair_overhead_ms\": " + std::to_string(stat_pair_ms_) + ", \"kernels\": [";
    bool first = true;
    char buf[512];
    for (int k = 0; k < KID_COUNT; ++k) {
      if (!agg[k].launches) continue;
      const bool gated = gated_kernel(k);
      std::snprintf(buf, sizeof buf,
                    "%s{\"name\": \"%s\", \"launches\": %lld, \"noop_launches\": %lld, \"total_ms\": %.6f, \"bytes_survey\": %.0f, "
                    "\"bytes_moved\": %.0f, \"gated\": %s, \"inclusive\": %s}",
                    first ? "" : ", ", kernel_name(k), (long long)agg[k].launches, (long long)agg[k].noops, agg[k].ms, agg[k].bytes_survey,
                    agg[k].bytes_moved,
                    gated ? "true" : "false", k == KID_EXT ? "true" : "false");
      o += buf;
      first = false;
    }
    // slab-decomposed solve: threshold searches that went through the speculative exchange since the context was finalised,
    // and how many of them needed their fallback (refinement rounds + full-size exchange)
    o += "], \"slab_searches\": {\"speculative_exchange\": " + std::to_string(spec_searches_) + ", \"fallbacks\": " +
         std::to_string(spec_fallbacks_) + ", \"refinement_rounds\": " + std::to_string(spec_rounds_) + "}";
    // one rank: searches through the batched chain (batched_searches) and how many of them needed their fallback sweeps
    o += std::string(", \"sparse_arrays\": ") + (slab_local_ ? "true" : "false");
    {
      // slab-decomposed long lists: which sets are projected on a materialised v -- by every rank on its own slices, or by an
      // owner rank on the gathered array -- and the fan exchanges (gathers + scatters of N w bytes) that took so far
      std::string loc, fan;
      for (int i = 0; i < p_n_; ++i) {
        if (sets_[i].slab_ext || sets_[i].slab_card || sets_[i].slab_dft) loc += (loc.empty() ? "" : ", ") + std::to_string(i);
        if (sets_[i].fan) fan += (fan.empty() ? "" : ", ") + std::to_string(i);
      }
      if (counting_)
        o += ", \"collectives\": {\"allreduce\": " + std::to_string(counting_->n_allreduce) + ", \"allreduce_with_halo\": " + std::to_string(counting_->n_grouped) +
             ", \"allgather\": " + std::to_string(counting_->n_allgather) + ", \"reduce_scatter\": " + std::to_string(counting_->n_reduce_scatter) +
             ", \"halo\": " + std::to_string(counting_->n_halo) + ", \"scatter_gather\": " + std::to_string(counting_->n_fan) +
             ", \"alltoall\": " + std::to_string(counting_->n_alltoall) + "}";
      o += ", \"slab_loose\": {\"slab_local_sets\": [" + loc + "], \"gathered_sets\": [" + fan + "], \"fan_exchanges\": " + std::to_string(fan_exchanges_) + "}";
    }
    {
      std::string t = selftest_;
      for (auto& ch : t) if (ch == '"' || ch == '\\' || (unsigned char)ch < 32) ch = ' ';
      o += ", \"comm_selftest\": \"" + t + "\"";
    }
    o += ", \"lane_set\": " + std::to_string(lane_set_);       // the set updated on a stream of its own (-1: none), lane_start
    o += ", \"batched_searches\": {\"searches\": " + std::to_string(batch_searches_) + ", \"fallbacks\": " + std::to_string(batch_fallbacks_) + "}";
    // slice-rank / matrix-rank sets: which route their projector took since the context was finalised (ext_proj.hip)
    long long rc[6] = {0, 0, 0, 0, 0, 0};
    for (const auto& st : sets_) {
      if (!st.ext || st.ext_kind != EXT_RANK) continue;
      long long c[6];
      st.ext->route_counts(c);
      for (int q = 0; q < 6; ++q) rc[q] += c[q];
    }
    o += ", \"rank_route\": {\"calls\": " + std::to_string(rc[0]) + ", \"warm_started_subspace\": " + std::to_string(rc[1]) +
         ", \"full_decomposition\": " + std::to_string(rc[2]) + ", \"products_with_gram\": " + std::to_string(rc[3]) +
         ", \"float32_loops\": " + std::to_string(rc[4]) + ", \"float32_gave_up\": " + std::to_string(rc[5]) + "}}";
    if (!peek) start_stats(enable);
    return o.c_str();
  }
  double stat_pair_overhead_ms() const { return stat_pair_ms_; }

  void debug_proj(int set, int which, double* o) override {
    need_final();
    for (int k = 0; k < 16; ++k) o[k] = 0;
    if (set < 0 || set >= p_n_) throw std::runtime_error("set index out of range");
    const ProjScalars<T>* d = (which & 1) ? sets_[set].psf : sets_[set].ps;
    if (!d) return;
    ProjScalars<T> h;
    SIPX_HIP(hipStreamSynchronize(stream_));
    SIPX_HIP(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    o[0] = h.need; o[1] = h.theta; o[2] = h.theta_prev; o[3] = h.hw; o[4] = h.spec_lo; o[5] = h.spec_hi; o[6] = h.lo;
    o[7] = h.hi; o[8] = h.asum; o[9] = h.vmax; o[10] = h.dbg[0]; o[11] = h.dbg[1]; o[12] = h.dbg[2]; o[13] = h.dbg[3];
    o[14] = h.refine;
    o[15] = h.lean + 2 * h.dbg_sampled;
    if (which & 2) { o[4] = h.samp_theta; o[5] = h.samp_lo; o[6] = h.samp_hi; o[7] = h.samp_c; }     // the sampled prediction
  }

  void* stream() override { return (void*)stream_; }
  void* dev_rhs() override { return rhs_; }
  void* dev_x() override { return x_; }
  void get_rhs(void* out) override {
    need_final();
    if (slab_local_ && comm_->world > 1) throw std::runtime_error("a slab-decomposed context holds its planes of rhs only");
    SIPX_HIP(hipStreamSynchronize(stream_));
    SIPX_HIP(hipMemcpy(out, rhs_, Nx_ * sizeof(T), hipMemcpyDeviceToHost));
  }

 private:
  struct KSample { int kid; double bytes_survey, bytes_moved; };
  struct KAgg { long long launches = 0, noops = 0; double ms = 0, bytes_survey = 0, bytes_moved = 0; };
  hipEvent_t stat_event(size_t i) {
    while (stat_ev_.size() <= i) {
      hipEvent_t e;
      SIPX_HIP(hipEventCreate(&e));
      stat_ev_.push_back(e);
    }
    return stat_ev_[i];
  }
  static void obs_begin(void* user, int kid, hipStream_t s, double bytes_survey, double bytes_moved) {
    auto* e = static_cast<Engine<T>*>(user);
    if (e->stats_mode_ == 0 || (e->stats_mode_ == 1 && kid != KID_CDS_DOT && kid != KID_SQ_DOT)) return;
    const size_t i = e->samples_.size();
    e->samples_.push_back(KSample{kid, bytes_survey, bytes_moved});
    e->open_.push_back(i);
    SIPX_HIP(hipEventRecord(e->stat_event(2 * i), s));
  }
  static void obs_end(void* user, int kid, hipStream_t s) {
    auto* e = static_cast<Engine<T>*>(user);
    if (e->stats_mode_ == 0 || (e->stats_mode_ == 1 && kid != KID_CDS_DOT && kid != KID_SQ_DOT)) return;
    if (e->open_.empty()) return;
    const size_t i = e->open_.back();
    e->open_.pop_back();
    if (i < e->samples_.size()) SIPX_HIP(hipEventRecord(e->stat_event(2 * i + 1), s));
  }
  void drop_samples_from(size_t n) {
    if (n < samples_.size()) samples_.resize(n);
    open_.clear();
  }
  void sync_all_streams() {
    SIPX_HIP(hipStreamSynchronize(stream_));
    for (hipStream_t q : pool_) if (q != stream_) SIPX_HIP(hipStreamSynchronize(q));
    if (cstream_) SIPX_HIP(hipStreamSynchronize(cstream_));
  }
  // kernels some of whose launches return at once on a device-side condition
  static bool gated_kernel(int k) {
    return k == KID_PASS_FIRST || k == KID_PASS_LEAN || k == KID_PASS_PROBE || k == KID_PASS_COMPACT || k == KID_SAMPLE ||
           k == KID_DECIDE || k == KID_CDS_FUSED || k == KID_CG_XR || k == KID_CG_P || k == KID_SLOT_SUMS;
  }
  std::vector<KAgg> aggregate_samples() {
    sync_all_streams();
    std::vector<KAgg> agg(KID_COUNT);
    for (size_t i = 0; i < samples_.size(); ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, stat_ev_[2 * i], stat_ev_[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
      KAgg& a = agg[samples_[i].kid];
      // two records with nothing between them are still some microseconds apart on the stream's timeline; that share of
      // every sample is not kernel time (calibrated on the idle stream when the collection is switched on)
      const double t = (double)ms - stat_pair_ms_;
      a.launches += 1;
      a.ms += t > 0 ? t : 0;
      // a gated launch (the pass of an l1 search that the device-side state did not ask for, a CG kernel past convergence) returns at
      // once: a launch that finished sooner than its bytes could cross the fabric at twice the HBM peak did not move them
      const bool noop = gated_kernel(samples_[i].kid) && samples_[i].bytes_moved > 0 && t * 1e-3 < samples_[i].bytes_moved / 16e12;
      a.noops += noop ? 1 : 0;
      a.bytes_survey += noop ? 0.0 : samples_[i].bytes_survey;
      a.bytes_moved += noop ? 0.0 : samples_[i].bytes_moved;
    }
    return agg;
  }
  void start_stats(int mode) {
    samples_.clear();
    open_.clear();
    stats_mode_ = 0;
    if (mode) {
      std::vector<float> gap;
      for (int k = 0; k < 16; ++k) {
        SIPX_HIP(hipEventRecord(stat_event(2 * k), stream_));
        SIPX_HIP(hipEventRecord(stat_event(2 * k + 1), stream_));
      }
      SIPX_HIP(hipStreamSynchronize(stream_));
      for (int k = 0; k < 16; ++k) {
        float ms = 0;
        SIPX_HIP(hipEventElapsedTime(&ms, stat_ev_[2 * k], stat_ev_[2 * k + 1]));
        gap.push_back(ms);
      }
      std::sort(gap.begin(), gap.end());
      stat_pair_ms_ = gap[gap.size() / 2];
    }
    stats_mode_ = mode;
  }
  const LaunchObserver* observer() const { return stats_mode_ ? &obs_ : nullptr; }

  void need_final() const {
    if (!finalized_) throw std::runtime_error("call sipx_finalize first");
    SIPX_HIP(hipSetDevice(device_));
  }

  // projector part of a set descriptor: validation (the reference's error messages) + routing
  void configure_proj(SetState<T>& s, const sipx_set_desc* d) {
    s.prox = d->proj;
    s.ncvx = d->ncvx;
    s.plo = (T)d->pmin;
    s.phi = (T)d->pmax;
    if (d->component < 0 || d->component > 3) throw std::runtime_error("Minkowski component must be 0, 1, 2 or 3");
    s.comp = d->component;
    const int mode = d->mode, dir = d->dir;
    if (mode != SIPX_MODE_WHOLE && mode != SIPX_MODE_FIBER && mode != SIPX_MODE_SLICE)
      throw std::runtime_error("unknown application mode");
    if (mode != SIPX_MODE_WHOLE) {
      if (dir < 0 || dir >= ndim_) throw std::runtime_error("application mode: direction out of range for this grid");
      if (s.nblk > 1) throw std::runtime_error("fiber / slice modes need an operator with one block (identity, D_x, D_y, D_z)");
    }
    auto ext = [&](int kind) {
      if (s.nblk > 1) throw std::runtime_error("this projector needs an operator with one block (identity, D_x, D_y, D_z)");
      s.ext_kind = kind;
      s.spec.kind = kind;
      s.spec.ndim = ndim_;
      s.spec.G = G_;
      for (int a = 0; a < 3; ++a) s.spec.dims[a] = G_.n[a];
      if (s.nblk == 1) s.spec.dims[s.dir[0]] -= 1;           // TD_n of a difference operator (get_TD_operator.jl)
      s.spec.mode = mode;
      s.spec.dir = dir;
      s.spec.pmin = d->pmin;
      s.spec.pmax = d->pmax;
      s.prox = PX_EXT;
      need_ext_ = true;
    };
    if (d->transform == SIPX_TRANSFORM_DCT) {      // x -> C' P(C x): P acts on the DCT coefficients (ext_proj.hip)
      if (d->op != SIPX_OP_IDENTITY || mode != SIPX_MODE_WHOLE)
        throw std::runtime_error("sets behind the DCT act in their own domain: TD_OP must be the identity, mode matrix/tensor");
      if (d->proj != SIPX_PROJ_BOUNDS && d->proj != SIPX_PROJ_BOUNDS_VEC && d->proj != SIPX_PROJ_L1 &&
          d->proj != SIPX_PROJ_CARDINALITY)
        throw std::runtime_error("behind the DCT: bounds, the l1 ball and cardinality are built (l2 / annulus commute with it)");
      if (d->proj == SIPX_PROJ_L1 && !(d->pmax > 0)) throw std::runtime_error("Radius of L1 ball is negative");
      if (d->proj == SIPX_PROJ_BOUNDS_VEC) {
        if (!d->lb || !d->ub) throw std::runtime_error("per-element bounds need lb and ub");
        s.host_lb.assign((const T*)d->lb, (const T*)d->lb + G_.N);
        s.host_ub.assign((const T*)d->ub, (const T*)d->ub + G_.N);
      }
      ext(EXT_DCT);
      s.spec.inner = d->proj;
      return;
    } else if (d->transform != SIPX_TRANSFORM_NONE) {
      throw std::runtime_error("unknown transform");
    }
    switch (d->proj) {
      case SIPX_PROJ_BOUNDS:
        if (mode != SIPX_MODE_WHOLE) throw std::runtime_error("scalar bounds apply to the whole array (use per-fiber vectors)");
        break;
      case SIPX_PROJ_PROX_L1:
        if (mode != SIPX_MODE_WHOLE) throw std::runtime_error("prox_l1 applies to the whole array");
        break;
      case SIPX_PROJ_BOUNDS_VEC:
        if (!d->lb || !d->ub) throw std::runtime_error("per-element bounds need lb and ub");
        if (mode == SIPX_MODE_SLICE)
          throw std::runtime_error("bound constraints per slice of a tensor currently not implemented, yet...");   // project_bounds!.jl:83
        if (mode == SIPX_MODE_FIBER) {      // bounds per fiber: expanded to one bound per row of A_i (reference order)
          long long dims[3] = {G_.n[0], G_.n[1], G_.n[2]};
          if (s.nblk == 1) dims[s.dir[0]] -= 1;
          const T *LB = (const T*)d->lb, *UB = (const T*)d->ub;
          s.host_lb.resize(s.Mtrue);
          s.host_ub.resize(s.Mtrue);
          long long r = 0;
          for (long long k = 0; k < dims[2]; ++k)
            for (long long j = 0; j < dims[1]; ++j)
              for (long long i = 0; i < dims[0]; ++i, ++r) {
                const long long c = dir == 0 ? i : (dir == 1 ? j : k);
                s.host_lb[r] = LB[c];
                s.host_ub[r] = UB[c];
              }
          s.plo = T(1);                       // min(max(x, LB), UB): the fiber methods clip with LB first (project_bounds!.jl:47,65)
        } else {
          s.host_lb.assign((const T*)d->lb, (const T*)d->lb + s.Mtrue);
          s.host_ub.assign((const T*)d->ub, (const T*)d->ub + s.Mtrue);
          s.plo = T(0);
        }
        break;
      case SIPX_PROJ_L1:
        if (mode != SIPX_MODE_WHOLE)
          throw std::runtime_error("l1 and l2 constraints only available for matrix or tensor mode, currently");   // setup_constraints.jl:65-67
        if (!(d->pmax > 0)) throw std::runtime_error("Radius of L1 ball is negative");   // project_l1_Duchi!.jl:22
        s.two_pass = true;
        break;
      case SIPX_PROJ_L2:
        if (mode != SIPX_MODE_WHOLE)
          throw std::runtime_error("l1 and l2 constraints only available for matrix or tensor mode, currently");
        s.two_pass = true;
        break;
      case SIPX_PROJ_ANNULUS:
        if (mode != SIPX_MODE_WHOLE) throw std::runtime_error("annulus constraints apply to the whole array");
        s.two_pass = true;
        break;
      case SIPX_PROJ_CARDINALITY:
        if (mode == SIPX_MODE_WHOLE) { s.two_pass = true; need_idx_ = true; }
        else ext(EXT_CARD_SEG);
        break;
      case SIPX_PROJ_L1_DFT:
        if (d->op != SIPX_OP_IDENTITY || mode != SIPX_MODE_WHOLE)
          throw std::runtime_error("the DFT-l1 projector acts in its own domain: TD_OP must be the identity, mode matrix/tensor");
        if (!(d->pmax > 0)) throw std::runtime_error("Radius of L1 ball is negative");
        ext(EXT_L1_DFT);
        break;
      case SIPX_PROJ_BOUNDS_DFT:
        if (d->op != SIPX_OP_IDENTITY || mode != SIPX_MODE_WHOLE)
          throw std::runtime_error("bounds in the DFT domain act in their own domain: TD_OP must be the identity, mode matrix/tensor");
        if (!d->ub) throw std::runtime_error("bounds in the DFT domain need the mask vector (constraint.max)");
        ext(EXT_DFT_MASK);
        s.host_ub.assign((const T*)d->ub, (const T*)d->ub + G_.N);
        break;
      case SIPX_PROJ_RANK: ext(EXT_RANK); break;
      case SIPX_PROJ_NUCLEAR: ext(EXT_NUCLEAR); break;
      case SIPX_PROJ_HISTOGRAM:
        if (!d->lb || !d->ub) throw std::runtime_error("histogram constraints need sorted lb and ub vectors");
        ext(EXT_HISTOGRAM);
        s.host_lb.assign((const T*)d->lb, (const T*)d->lb + s.Mtrue);     // kept on the host until the ExtProj is built
        s.host_ub.assign((const T*)d->ub, (const T*)d->ub + s.Mtrue);
        break;
      case SIPX_PROJ_SUBSPACE:
        if (d->op != SIPX_OP_IDENTITY) throw std::runtime_error("subspace constraints act on the model itself: TD_OP must be the identity");
        if (!d->basis || d->basis_cols < 1 || d->basis_rows < 1) throw std::runtime_error("subspace constraints need the matrix A");
        ext(EXT_SUBSPACE);
        s.host_basis.assign((const T*)d->basis, (const T*)d->basis + (size_t)d->basis_rows * d->basis_cols);
        s.spec.basis_rows = d->basis_rows;
        s.spec.basis_cols = d->basis_cols;
        s.spec.basis_orth = d->basis_orth;
        break;
      default: throw std::runtime_error("unknown projector kind");
    }
  }

  // SIPX_OP_CSC: validate the SparseMatrixCSC arrays and keep host copies until sipx_finalize
  void configure_custom(SetState<T>& s, const sipx_set_desc* d) const {
    const long long N = G_.N, M = d->csc_rows;
    if (!d->csc_colptr || !d->csc_rowval || !d->csc_nzval || M < 1) throw std::runtime_error("custom operator: CSC arrays missing");
    const long long nnz = d->csc_colptr[N];
    if (d->csc_colptr[0] != 0 || nnz < 0) throw std::runtime_error("custom operator: colptr must start at 0 (0-based arrays)");
    for (long long j = 0; j < N; ++j) {
      if (d->csc_colptr[j + 1] < d->csc_colptr[j]) throw std::runtime_error("custom operator: colptr must be non-decreasing");
      for (long long k = d->csc_colptr[j]; k < d->csc_colptr[j + 1]; ++k) {
        const long long r = d->csc_rowval[k];
        if (r < 0 || r >= M) throw std::runtime_error("custom operator: row index out of range");
        if (k > d->csc_colptr[j] && r <= d->csc_rowval[k - 1]) throw std::runtime_error("custom operator: rows must ascend inside a column");
      }
    }
    s.op = SIPX_OP_CSC;
    s.custom = true;
    s.nblk = 0;                       // the set kernels see an identity over M entries
    s.ident = false;
    s.Mtrue = s.Mpad = M;
    s.h_colptr.assign(d->csc_colptr, d->csc_colptr + N + 1);
    s.h_rowval.assign(d->csc_rowval, d->csc_rowval + nnz);
    s.h_nzval.assign((const T*)d->csc_nzval, (const T*)d->csc_nzval + nnz);
    s.gm.n[0] = M; s.gm.n[1] = 1; s.gm.n[2] = 1; s.gm.N = M; s.gm.st[0] = 1; s.gm.st[1] = M; s.gm.st[2] = M;
  }
  // device copies: CSC as given, and the CSR view (rows ascending, columns ascending inside a row: a counting sort by row
  // of the column-major entries keeps that order) for s = A x
  void upload_custom(SetState<T>& s) {
    const long long N = G_.N, M = s.Mtrue, nnz = (long long)s.h_rowval.size();
    std::vector<long long> rowptr(M + 1, 0), colidx(nnz);
    std::vector<T> rval(nnz);
    for (long long k = 0; k < nnz; ++k) rowptr[s.h_rowval[k] + 1]++;
    for (long long r = 0; r < M; ++r) rowptr[r + 1] += rowptr[r];
    std::vector<long long> next(rowptr.begin(), rowptr.end() - 1);
    for (long long j = 0; j < N; ++j)
      for (long long k = s.h_colptr[j]; k < s.h_colptr[j + 1]; ++k) {
        const long long q = next[s.h_rowval[k]]++;
        colidx[q] = j;
        rval[q] = s.h_nzval[k];
      }
    auto up = [&](auto*& dst, const auto& src) {
      using E = typename std::remove_reference<decltype(src)>::type::value_type;
      dst = dalloc<E>(src.size(), false);
      SIPX_HIP(hipMemcpy(dst, src.data(), src.size() * sizeof(E), hipMemcpyHostToDevice));
    };
    up(s.d_colptr, s.h_colptr); up(s.d_rowval, s.h_rowval); up(s.d_nzval, s.h_nzval);
    up(s.d_rowptr, rowptr); up(s.d_colidx, colidx); up(s.d_rval, rval);
    s.sbuf = dalloc<T>(M);
    s.h_colptr.clear(); s.h_rowval.clear(); s.h_nzval.clear();
  }

  void configure_op(SetState<T>& s, int op) const {
    s.op = op;
    const int zdir = ndim_ == 2 ? 1 : 2;
    switch (op) {
      case SIPX_OP_IDENTITY: s.nblk = 0; break;
      case SIPX_OP_DX: s.nblk = 1; s.dir[0] = 0; break;
      case SIPX_OP_DY:
        if (ndim_ == 2) throw std::runtime_error("D_y needs a 3-D grid");
        s.nblk = 1; s.dir[0] = 1; break;
      case SIPX_OP_DZ: s.nblk = 1; s.dir[0] = zdir; break;
      case SIPX_OP_TV:                                   // vcat(D_z[,D_y],D_x): get_discrete_Grad.jl:31-33,69-72
        if (ndim_ == 2) { s.nblk = 2; s.dir[0] = 1; s.dir[1] = 0; }
        else { s.nblk = 3; s.dir[0] = 2; s.dir[1] = 1; s.dir[2] = 0; }
        break;
      default: throw std::runtime_error("unknown operator kind");
    }
    s.ident = s.nblk == 0;
    s.Mtrue = 0;
    for (int q = 0; q < s.nblk; ++q) {
      const int a = s.dir[q];
      if (G_.n[a] < 2) throw std::runtime_error("difference operator along a dimension of size 1");
      s.ih[q] = ih_[a];
      s.blk_rows[q] = G_.N / G_.n[a] * (G_.n[a] - 1);
      s.Mtrue += s.blk_rows[q];
    }
    if (s.ident) s.Mtrue = G_.N;
    s.Mpad = s.ident ? G_.N : (long long)s.nblk * G_.N;
  }

  std::vector<long long> default_ata_offsets(const SetState<T>& s) const {   // mat2CDS: ascending (mat2CDS.jl:9-13)
    std::vector<long long> o = {0};
    for (int q = 0; q < s.nblk; ++q) {
      o.push_back(G_.st[s.dir[q]]);
      o.push_back(-G_.st[s.dir[q]]);
    }
    std::sort(o.begin(), o.end());
    o.erase(std::unique(o.begin(), o.end()), o.end());
    if (s.comp == 3) {     // [B B; B B]: every diagonal of B also shifted by -N and +N (mat2CDS of the 2N x 2N block matrix)
      std::vector<long long> all;
      for (long long sh : {-G_.N, 0ll, G_.N})
        for (long long v : o) all.push_back(v + sh);
      std::sort(all.begin(), all.end());
      all.erase(std::unique(all.begin(), all.end()), all.end());
      return all;
    }
    return o;
  }

  int q_col(long long off) const {
    for (int b = 0; b < cds_.d; ++b)
      if (cds_.off[b] == off) return b;
    throw std::runtime_error("attempted to update a diagonal in A in CDS storage that does not exist. A and B need "
                             "to have the same nonzero diagonals");   // CDS_scaled_add!.jl:18-20
  }

  void plan_Q_offsets() {  // PARSDMM_initialize.jl:216-221: first-seen offsets over a zero-padded table
    std::vector<long long> seen;
    auto see = [&](long long o) {
      if (std::find(seen.begin(), seen.end(), o) == seen.end()) seen.push_back(o);
    };
    for (auto& s : sets_) {
      if (s.ata_off.size() > 999) throw std::runtime_error("more than 999 bands in one set");
      for (long long o : s.ata_off) see(o);
      see(0);
    }
    if ((int)seen.size() > MAXD) throw std::runtime_error("Q has more bands than this build supports");
    cds_.d = (int)seen.size();
    for (int b = 0; b < cds_.d; ++b) cds_.off[b] = seen[b];
    // symmetric read of Q (CdsArgs::sym): every negative band needs its positive partner, at a band index >= 1 so that
    // the shifted address never leaves the allocation; explicit AtA bands must be symmetric bit for bit
    bool ok = !cds_full_;
    for (int b = 0; b < cds_.d && ok; ++b) {
      cds_.partner[b] = b;
      if (cds_.off[b] >= 0) continue;
      int pb = -1;
      for (int c = 0; c < cds_.d; ++c)
        if (cds_.off[c] == -cds_.off[b]) pb = c;
      if (pb < 1) ok = false;
      else cds_.partner[b] = pb;
    }
    for (auto& s : sets_) ok = ok && (s.host_ata.empty() || explicit_bands_symmetric(s));
    cds_.sym = ok ? 1 : 0;
    // the z-marching product (k_cds_march): the 7-band matrix of a 3-D grid in one of the two band orders it is compiled for
    cds_.march = 0;
    if (cds_.sym && cds_.d == 7 && ndim_ == 3 && !mk_) {
      const long long s1 = G_.st[1], s2 = G_.st[2];
      const long long o1[7] = {0, -1, 1, -s1, s1, -s2, s2}, o2[7] = {0, -s2, -s1, -1, 1, s1, s2};
      bool m1 = true, m2 = true;
      for (int b = 0; b < 7; ++b) { m1 = m1 && cds_.off[b] == o1[b]; m2 = m2 && cds_.off[b] == o2[b]; }
      cds_.march = m1 ? 1 : (m2 ? 2 : 0);
      const long long want[4] = {0, 1, s1, s2};
      for (int q = 0; q < 4; ++q)
        for (int b = 0; b < 7; ++b)
          if (cds_.off[b] == want[q]) cds_.mb[q] = b;
      for (int a = 0; a < 3; ++a) cds_.gn[a] = G_.n[a];
    }
  }

  bool explicit_bands_symmetric(const SetState<T>& s) const {
    const long long N = G_.N;
    for (size_t b = 0; b < s.ata_off.size(); ++b) {
      const long long o = s.ata_off[b];
      if (o >= 0) continue;
      long long pb = -1;
      for (size_t c = 0; c < s.ata_off.size(); ++c)
        if (s.ata_off[c] == -o) pb = (long long)c;
      if (pb < 0) return false;
      const T* lo = s.host_ata.data() + b * (size_t)N;
      const T* up = s.host_ata.data() + (size_t)pb * (size_t)N;
      for (long long r = -o; r < N; ++r)
        if (std::memcmp(&lo[r], &up[r + o], sizeof(T)) != 0) return false;
    }
    return true;
  }

  // stencil mode: the whole of Q is four scalars, recomputed from the current rho (no update history)
  void stencil_weights(const double* rho) {
    double w0 = 0, w[3] = {0, 0, 0};
    sq_.mask = 0;
    for (int i = 0; i < p_n_; ++i) {
      const SetState<T>& s = sets_[i];
      const double r = (double)(T)rho[i];
      if (s.nblk == 0) w0 += r;
      for (int q = 0; q < s.nblk; ++q) {
        w[s.dir[q]] += r * (double)(T)(s.ih[q] * s.ih[q]);
        sq_.mask |= 1 << s.dir[q];
      }
    }
    sq_.w0 = (T)w0;
    for (int d = 0; d < 3; ++d) sq_.w[d] = (T)w[d];
  }

  void assemble_Q() {      // PARSDMM_initialize.jl:222-230
    if (stencil_q_) {
      for (auto& s : sets_)
        if (s.ata) throw std::runtime_error("stencil Q mode needs descriptor-generated AtA for every set (pass ata_R = NULL)");
      std::vector<double> r(rho_.begin(), rho_.end());
      stencil_weights(r.data());
      return;
    }
    // (sparse arrays: the rows of every band that the rank's part of the x-step reads)
    if (Q_) {                       // sipx_reset: the bands are there, zero them and add the sets up again
      dzero(Q_, stream_);
    } else if (slab_local_) {
      std::vector<std::pair<size_t, size_t>> rg;
      long long maxoff = 0;
      for (int b = 0; b < cds_.d; ++b) maxoff = std::max<long long>(maxoff, std::llabs(cds_.off[b]));
      for (int b = 0; b < cds_.d; ++b) {
        // rows [r0 - maxoff, r1) of every band: the symmetric read takes band +o at row r - o -- for the first rows of the grid
        // that is the TAIL of the band stored in front (masked, but loaded), so the range is not clipped at the band's first row
        const long long lo = std::max<long long>(0, (long long)b * Nx_ + (r1_ > r0_ ? r0_ : 0) - maxoff - 64);
        const long long hi = std::min<long long>((long long)Nx_ * cds_.d, (long long)b * Nx_ + std::max(qr1_, qr0_) + 64);
        if (hi > lo) rg.push_back({(size_t)lo * sizeof(T), (size_t)hi * sizeof(T)});
      }
      Q_ = (T*)sparse_alloc_bytes((size_t)Nx_ * cds_.d * sizeof(T), rg, device_);
    } else {
      Q_ = dalloc<T>((size_t)Nx_ * cds_.d);
    }
    if (mk_) {
      std::vector<T> al(rho_.begin(), rho_.end());
      mk_q_update(al);
      return;
    }
    QArgs<T> a;
    a.nsets = 0;
    for (int i = 0; i < p_n_; ++i) push_qset(a, sets_[i], rho_[i]);   // Q = 0 + rho_1 AtA_1 + rho_2 AtA_2 + ...
    K<T>::q_update(stream_, G_, qr0_, qr1_, cds_, a, Q_);
  }

  // Minkowski mode: Q[:, b] += alpha_i AtA_i[:, b] for the sets with alpha_i != 0, batches of MAX_SETS in set order
  void mk_q_update(const std::vector<T>& alpha) {
    MkArgs<T> a;
    a.nsets = 0;
    for (int i = 0; i < p_n_; ++i) {
      if (alpha[i] == T(0)) continue;
      const SetState<T>& s = sets_[i];
      for (long long o : s.ata_off) (void)q_col(o);
      MkSet<T>& m = a.s[a.nsets++];
      m.alpha = alpha[i];
      m.nblk = s.nblk;
      m.comp = s.comp;
      for (int k = 0; k < 3; ++k) { m.dir[k] = s.dir[k]; m.ih[k] = s.ih[k]; }
      if (a.nsets == MAX_SETS) {
        K<T>::q_update_mk(stream_, G_, cds_, a, Q_);
        a.nsets = 0;
      }
    }
    K<T>::q_update_mk(stream_, G_, cds_, a, Q_);
  }

  void push_qset(QArgs<T>& a, const SetState<T>& s, T alpha) {
    if (s.ata_off.size() > 9) throw std::runtime_error("more than 9 bands in one set's AtA are not supported");
    for (long long o : s.ata_off) (void)q_col(o);     // CDS_scaled_add!.jl:18-20: the diagonal must exist in Q
    if (a.nsets == MAX_SETS) {                         // flush a full batch, keep the order
      K<T>::q_update(stream_, G_, qr0_, qr1_, cds_, a, Q_);
      a.nsets = 0;
    }
    QSet<T>& q = a.s[a.nsets++];
    q.alpha = alpha;
    q.ata = s.ata;
    q.nblk = s.nblk;
    for (int k = 0; k < 3; ++k) { q.dir[k] = s.dir[k]; q.ih[k] = s.ih[k]; }
    q.nband = (int)s.ata_off.size();
    for (int k = 0; k < q.nband; ++k) q.off[k] = s.ata_off[k];
  }

  SetArgs<T> set_args(const SetState<T>& s, T rho, T gamma, int flags) const {
    SetArgs<T> a;
    a.y = s.y; a.l = s.l; a.dy = s.dy; a.lh0 = s.lh0; a.s0 = s.s0;
    a.y0 = s.snap == 0 ? s.y : s.y0;          // the snapshot pair (the zero-filled other pair before the first snapshot)
    a.l0 = s.snap == 0 ? s.l : s.l0;
    a.yo = s.y; a.lo = s.l;
    a.v = scr_v_;
    a.x = x_; a.m = m_; a.xold = xold_; a.lb = s.lb; a.ub = s.ub;
    a.nblk = s.nblk;
    for (int q = 0; q < 3; ++q) { a.dir[q] = s.dir[q]; a.ih[q] = s.ih[q]; }
    a.rho = rho;
    a.rho1 = T(1) / rho;       // rho1 = TF(1.0) ./ rho   update_y_l.jl:33-34
    a.gamma = gamma;
    a.prox = s.prox;
    a.plo = s.plo; a.phi = s.phi;
    a.ps = s.ps;
    a.flags = flags;
    a.vsrc = 0;
    return a;
  }

  // The same for a set that all ranks project together (identity operator: s = src): this rank's slab of slices only, the
  // partial sums of the ranks add up in the all-reduce of the packed per-set sums.
  void dist_feasibility(SetState<T>& s, const T* src, double* dst, int set = 0) {
    const long long nloc = r1_ - r0_;
    if (nloc > 0) {
      SIPX_HIP(hipMemcpyAsync(loose_v_ + r0_, src + r0_, nloc * sizeof(T), hipMemcpyDeviceToDevice, stream_));
      SIPX_HIP(hipMemcpyAsync(loose_w_ + r0_, src + r0_, nloc * sizeof(T), hipMemcpyDeviceToDevice, stream_));
      if (!s.slab_dft) s.ext->project(loose_v_ + r0_, true, part_tmp_, maxpart_, scr_c_);
    }
    // (the slab-decomposed transform is a collective: ranks without planes take part)
    if (s.slab_dft) s.ddft->project(loose_v_ + r0_, true, comm_.get(), &hooks_, part_tmp_, maxpart_, scr_c_, scr_c_len_, (int*)hovf_ + set);
    ext_dist2<T>(stream_, nloc, loose_v_ + r0_, loose_w_ + r0_, dst);
  }

  // A gathered set of a slab-decomposed solve (SetState::fan).  fan_collect: every rank materialises v (v_is_s = 0) or s = A x
  // (1) on its planes -- rows a difference operator does not have stay zero -- and the owner gathers the whole array;
  // fan_project_owner: P in place on the owner (a library-backed projector, or the cardinality search on the array: the k-th
  // magnitude and the tie rule by lowest padded index are those of the on-the-fly search, the zeros of the missing rows are the
  // smallest magnitudes there are); fan_return: every rank receives its planes of P(v), and the last plane of the rank below,
  // which the update recomputes for the adjoint stencils (Gyl_).  Engine stream, set order: the same on every rank.
  // In an update the gathered sets come FIRST (fan_begin: collect, the owner projects on the fan stream, which waits for the
  // gather), then every rank's own sets on the engine stream, and only then the scatters (fan_return waits for the fan stream): an
  // owner's whole-array projection runs beside its share of the slab-local sets instead of in front of everybody's.
  void fan_collect(SetState<T>& s, const SetArgs<T>& a, int v_is_s, T* buf) {
    if (r1_ > r0_) SIPX_HIP(hipMemsetAsync(buf + r0_, 0, (size_t)(r1_ - r0_) * sizeof(T), stream_));
    SetArgs<T> b = a;
    b.v = buf;
    K<T>::store_v(stream_, Gr_, b, v_is_s, buf);
    comm_->gather(buf, (size_t)chunk_, dtype_code(), s.fan_owner, stream_);
    ++fan_exchanges_;
  }
  void fan_project_owner(SetState<T>& s, bool feas, T* buf, hipStream_t q, double* ptmp, T* mpart, T* cbuf) {
    ObsScope obs(KID_EXT, q, 0.0);
    if (s.ext_kind) {
      s.ext->set_stream(q);
      s.ext->project(buf, feas, ptmp, mpart, cbuf);
    } else {
      ProjScalars<T>* ps = feas ? s.psf : s.ps;
      K<T>::proj_scalars_arr(q, G_.N, buf, s.prox, s.plo, s.phi, ps, ptmp, mpart, cbuf, s.Mtrue);
      proj_apply_grid<T>(q, G_, s.nblk, s.dir, G_.N, buf, s.prox, s.plo, s.phi, (const T*)nullptr, (const T*)nullptr, ps);
    }
  }
  void fan_begin(SetState<T>& s, const SetArgs<T>& a) {
    fan_collect(s, a, 0, s.fanv);
    if (comm_->rank != s.fan_owner) return;
    // (the all-kernel statistics window times one kernel at a time on the engine stream: in turn there)
    hipStream_t q = (fan_st_ && stats_mode_ != 2 && fan_overlap_) ? fan_st_ : stream_;
    if (q != stream_) {
      SIPX_HIP(hipEventRecord(fan_fork_, stream_));
      SIPX_HIP(hipStreamWaitEvent(q, fan_fork_, 0));
    }
    // (in turn on the engine stream the fan stream's scratch is idle: its compaction buffer is whole-size, the engine's may not be)
    fan_project_owner(s, false, s.fanv, q, fan_ptmp_, fan_mpart_, fan_c_);
    if (q != stream_) SIPX_HIP(hipEventRecord(s.fan_ev, q));
    s.fan_on_side = q != stream_;
  }
  void fan_return(SetState<T>& s) {
    const int dt = dtype_code();
    T* buf = s.fanv;
    if (s.fan_on_side) { SIPX_HIP(hipStreamWaitEvent(stream_, s.fan_ev, 0)); s.fan_on_side = false; }
    comm_->scatter(buf, (size_t)chunk_, dt, s.fan_owner, stream_);
    ++fan_exchanges_;
    if (!s.ident && r1_ > r0_ && (prev_ >= 0 || next_ >= 0))
      comm_->halo_exchange(buf + r0_, buf + r0_ - plane_, prev_, buf + r1_ - plane_, buf + r1_, next_, (size_t)plane_, dt, stream_);
  }
  void fan_feasibility(SetState<T>& s, const SetArgs<T>& a, double* dst) {
    fan_collect(s, a, 1, loose_v_);
    if (comm_->rank != s.fan_owner) return;
    SIPX_HIP(hipMemcpyAsync(loose_w_, loose_v_, (size_t)s.Mpad * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    fan_project_owner(s, true, loose_v_, stream_, part_tmp_, maxpart_, fan_c_);      // (fan_c_: whole-size, this rank owns a gathered set)
    ext_dist2<T>(stream_, s.Mpad, loose_v_, loose_w_, dst);
  }

  // ||P(s) - s||^2, ||s||^2 for a library-backed projector: s = A x materialised twice, one copy projected
  void ext_feasibility(SetState<T>& s, const SetArgs<T>& a, double* dst) {
    K<T>::store_v(stream_, G_, a, 1, scr_v_);
    SIPX_HIP(hipMemcpyAsync(scr_w_, scr_v_, s.Mpad * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    s.ext->project(scr_v_, true, part_tmp_, maxpart_, scr_c_);
    ext_dist2<T>(stream_, s.Mpad, scr_v_, scr_w_, dst);
  }

  T feas_value(double fe, double ss) const {
    return (T)std::sqrt(fe) / ((T)std::sqrt(ss) + T(100) * std::numeric_limits<T>::epsilon());
  }

  // Import / export between the reference's row order (host) and the padded layout (device): one contiguous PCIe
  // copy plus a device gather / scatter per operator block (the pads keep the zeros they were allocated with).
  void upload_rows(const SetState<T>& s, const T* rows, T* dev) const {
    if (s.ident || s.custom) {
      SIPX_HIP(hipMemcpy(dev, rows, (size_t)s.Mtrue * sizeof(T), hipMemcpyHostToDevice));
      return;
    }
    T* tmp = dalloc<T>(s.Mtrue, false);
    SIPX_HIP(hipMemcpy(tmp, rows, (size_t)s.Mtrue * sizeof(T), hipMemcpyHostToDevice));
    long long r0 = 0;
    for (int q = 0; q < s.nblk; ++q) {
      K<T>::rows_unpack(stream_, G_, s.dir[q], s.blk_rows[q], tmp + r0, dev + (long long)q * G_.N);
      r0 += s.blk_rows[q];
    }
    SIPX_HIP(hipStreamSynchronize(stream_));
    dfree(tmp);
  }
  // (sparse arrays: through a whole-size temporary, of which the rank's share is kept)
  void upload_rows_ranged(const SetState<T>& s, const T* rows, T* dev) const {
    if (!slab_local_) { upload_rows(s, rows, dev); return; }
    T* full = dalloc<T>((size_t)s.Mpad);
    upload_rows(s, rows, full);
    const long long c0 = std::max<long long>(0, wlo_), c1 = std::min<long long>(G_.N, whi_);
    for (int q = 0; q < s.nblk_or1() && c1 > c0; ++q)
      SIPX_HIP(hipMemcpy(dev + (long long)q * G_.N + c0, full + (long long)q * G_.N + c0, (c1 - c0) * sizeof(T), hipMemcpyDeviceToDevice));
    SIPX_HIP(hipDeviceSynchronize());
    dfree(full);
  }
  void download_rows(const SetState<T>& s, const T* dev, T* rows) const {
    SIPX_HIP(hipStreamSynchronize(stream_));
    if (s.ident || s.custom) {
      host_prefault(rows, (size_t)s.Mtrue * sizeof(T));
      SIPX_HIP(hipMemcpy(rows, dev, (size_t)s.Mtrue * sizeof(T), hipMemcpyDeviceToHost));
      return;
    }
    T* tmp = dalloc<T>(s.Mtrue, false);
    long long r0 = 0;
    for (int q = 0; q < s.nblk; ++q) {
      K<T>::rows_pack(stream_, G_, s.dir[q], s.blk_rows[q], dev + (long long)q * G_.N, tmp + r0);
      r0 += s.blk_rows[q];
    }
    SIPX_HIP(hipStreamSynchronize(stream_));
    host_prefault(rows, (size_t)s.Mtrue * sizeof(T));
    SIPX_HIP(hipMemcpy(rows, tmp, (size_t)s.Mtrue * sizeof(T), hipMemcpyDeviceToHost));
    dfree(tmp);
  }

  // the event a timed step's first interval starts from (round 4: the first mark of a step used to open nothing when the residual
  // product had been queued ahead, and the x-step of every such iteration went uncounted -- "argmin x" read 0.21 of 0.5 ms)
  // (four slots, by step: the opening of step i + 1 is recorded during step i, before the marks of step i - 1 are resolved)
  void open_section(int step) {
    const int k = step & 3;
    if (!open_ev_[k]) SIPX_HIP(hipEventCreate(&open_ev_[k]));
    SIPX_HIP(hipEventRecord(open_ev_[k], stream_));
    open_valid_[k] = true;
  }
  void resolve_timing(sipx_log* log, int par) {
    const int n = nmark_[par];
    const int slot = mark_step_[par] & 3;
    if (!n) { open_valid_[slot] = false; return; }
    nmark_[par] = 0;
    // timing_ms: [0] initialization (host side) [1] rhs [2] argmin x [3] y/l update [4] stop rule [5] rho / gamma rules [6] Q update
    SIPX_HIP(hipEventSynchronize(ev_[par * MAXMARK + n - 1]));
    hipEvent_t prev = open_valid_[slot] ? open_ev_[slot] : nullptr;
    open_valid_[slot] = false;
    const double w = mark_weight_[par] > 0 ? mark_weight_[par] : 1.0;      // the iterations this timed one stands for
    for (int k = 0; k < n; ++k) {
      hipEvent_t e = ev_[par * MAXMARK + k];
      const int sec = mark_sec_[par][k];
      if (sec >= 0 && prev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, prev, e) == hipSuccess) log->timing_ms[sec] += w * ms;
        else (void)hipGetLastError();
      }
      prev = e;
    }
  }

  int dtype_code() const { return sizeof(T) == 8 ? SIPX_F64 : SIPX_F32; }

  // partials of the per-set slots -> hres_ (pinned).  Sharded: summed over the ranks on the way (sets a rank does not own
  // contribute the zeros their slots were allocated with), so every rank reads the sums of every set.
  void reduce_set_sums(int nslots) {
    if (comm_ && merge_sums_) {            // their all-reduce rides with the residual sums of the next x-step (argmin_x_head)
      K<T>::fin_sum(stream_, part_sets_, nslots, dres_, nullptr);
      merged_nslots_ = nslots;
    } else if (comm_) {
      K<T>::fin_sum(stream_, part_sets_, nslots, dres_, nullptr);
      comm_->allreduce_sum(dres_, (size_t)nslots, SIPX_F64, stream_);
      K<T>::copy_f64(stream_, dres_, hres_, nslots);
    } else if (word_sums_) {
      // (one rank, whole-solve loop: the host spins on a pinned word the last workgroup publishes, see k_fin_sum)
      K<T>::fin_sum(stream_, part_sets_, nslots, nullptr, hres_, sums_ticket_, (unsigned long long*)sums_word_, ++sums_seq_);
      sums_by_word_ = true;
      return;
    } else {
      K<T>::fin_sum(stream_, part_sets_, nslots, nullptr, hres_);
    }
    sums_by_word_ = false;
  }

  // Verdict of CG iteration `iter` of solve `seq`: spins on the ticket word, which the device publishes as soon as the
  // verdict is known (and after the state mirror, so that is complete too).  Every enqueued iteration publishes one; should
  // the stream nevertheless run dry without it (a faulted kernel), the mirror decides.
  bool wait_ticket(unsigned seq, int iter, const CgState<T>* mirror) {
    // ticket = (seq << 32) | (iter << 1) | done.  A ticket of a LATER iteration of this solve (small grids queue one
    // iteration ahead) means that this one did not converge: past `done` no kernel publishes anything.
    auto verdict = [&](unsigned long long t, bool& done) {
      if ((unsigned)(t >> 32) != seq) return false;
      const unsigned ti = (unsigned)(t & 0xffffffffull) >> 1;
      if (ti < (unsigned)iter) return false;
      done = ti == (unsigned)iter ? (t & 1ull) != 0 : false;
      return true;
    };
    bool done = false;
    for (unsigned spins = 1;; ++spins) {
      if (verdict(__atomic_load_n(ticket_, __ATOMIC_ACQUIRE), done)) return done;
      if ((spins & 0xfff) == 0) {
        const hipError_t q = hipStreamQuery(stream_);
        if (q == hipSuccess) {
          if (verdict(__atomic_load_n(ticket_, __ATOMIC_ACQUIRE), done)) return done;
          return mirror->done != 0;
        }
        if (q != hipErrorNotReady) SIPX_HIP(q);
        (void)hipGetLastError();        // hipErrorNotReady is not an error: keep it out of the launch checks
      }
    }
  }

  // the sequence number k_fin_sum publishes behind the sums (pinned; see there)
  void wait_word(volatile unsigned long long* word, unsigned long long seq) {
    for (unsigned spins = 1;; ++spins) {
      if (__atomic_load_n((unsigned long long*)word, __ATOMIC_ACQUIRE) == seq) return;
      if ((spins & 0x3ffff) == 0) {
        const hipError_t q = hipStreamQuery(stream_);
        if (q == hipSuccess) {
          if (__atomic_load_n((unsigned long long*)word, __ATOMIC_ACQUIRE) == seq) return;
          throw std::runtime_error("internal: the stream ran dry without the sums of the y/l update (a kernel faulted?)");
        }
        if (q != hipErrorNotReady) SIPX_HIP(q);
        (void)hipGetLastError();
      }
    }
  }

  void dump_ps(int set, const double* reg) {       // SIPX_SPEC_DEBUG: the state of a search as the device sees it (synchronises)
    ProjScalars<T> h;
    double r[PREP_SLOTS + 1];
    SIPX_HIP(hipStreamSynchronize(stream_));
    SIPX_HIP(hipMemcpy(&h, sets_[set].ps, sizeof(h), hipMemcpyDeviceToHost));
    SIPX_HIP(hipMemcpy(r, reg, sizeof(r), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "[sipx spec]     set %d: need %d spec_ok %d refine %d lean %d ovf %d bracket (%.9g, %.9g] spec (%.9g, %.9g] n_compact %llu t0 %.9g t7 %.9g\n"
                 "[sipx spec]       S", set, h.need, h.spec_ok, h.refine, h.lean, h.spec_overflow, h.lo, h.hi, h.spec_lo, h.spec_hi,
                 (unsigned long long)h.n_compact, h.t[0], h.t[L1_K - 1]);
    for (int k = 0; k < L1_K; ++k) std::fprintf(stderr, " %.6g", r[3 + k]);
    std::fprintf(stderr, " C");
    for (int k = 0; k < L1_K; ++k) std::fprintf(stderr, " %.0f", r[3 + L1_K + k]);
    std::fprintf(stderr, " asum %.6g ovf %.0f\n", r[0], r[PREP_SLOTS]);
    std::fprintf(stderr, "[sipx spec]       theta %.9g theta_prev %.9g hw %.4g sampled %d samp_theta %.9g samp (%.9g, %.9g) samp_c %.0f want_sample %d rescaled %d\n",
                 (double)h.theta, h.theta_prev, h.hw, h.sampled, h.samp_theta, h.samp_lo, h.samp_hi, h.samp_c, h.want_sample, h.rescaled);
  }

  // the word k_spec_decide publishes for a set: (seq << 2) | verdict bits
  unsigned wait_verdict(volatile unsigned* word, unsigned seq) {
    for (unsigned spins = 1;; ++spins) {
      const unsigned w = __atomic_load_n((unsigned*)word, __ATOMIC_ACQUIRE);
      if ((w >> 2) == seq) return w & 3u;
      if ((spins & 0x3ffff) == 0) {          // (a stream query costs the runtime tens of microseconds: rarely)
        const hipError_t q = hipStreamQuery(stream_);
        if (q == hipSuccess) {
          const unsigned w2 = __atomic_load_n((unsigned*)word, __ATOMIC_ACQUIRE);
          if ((w2 >> 2) == seq) return w2 & 3u;
          throw std::runtime_error("internal: the stream ran dry without the verdict of a threshold search (a kernel faulted?)");
        }
        if (q != hipErrorNotReady) SIPX_HIP(q);
        (void)hipGetLastError();
      }
    }
  }

  void free_set(SetState<T>& s) {
    if (s.st) (void)hipStreamSynchronize(s.st);
    if (s.ev) (void)hipEventDestroy(s.ev);
    if (s.fan_ev) (void)hipEventDestroy(s.fan_ev);
    dfree(s.fanv);
    for (void* p : {(void*)s.ptmp, (void*)s.mpart, (void*)s.cbuf, (void*)s.d_colptr, (void*)s.d_rowval, (void*)s.d_rowptr,
                    (void*)s.d_colidx, (void*)s.d_nzval, (void*)s.d_rval, (void*)s.sbuf})
      dfree(p);
    for (void* p : s.halo_allocs) dfree(p);
    for (void* p : {(void*)s.lh0, (void*)s.s0, (void*)s.lb, (void*)s.ub, (void*)s.ata,
                    (void*)s.ps, (void*)s.psf})
      dfree(p);
  }

  struct Run {     // state of one whole solve (sipx_parsdmm_begin / _steps)
    bool active = false, done = false;
    sipx_log* log = nullptr;
    int maxit = 0, freq = 2, counter = 2, ind_ref = 0, i = 0, last_timed = 0;
    T evol_rel_tol = 0, feas_tol = 0, obj_tol = 0;
    bool adjust_rho = true, adjust_gamma = true, adjust_feas_rho = true;
    double tol_ref = 1.0;
    bool rhs_ready = false;        // the right-hand side of the coming iteration is already queued
    std::vector<double> rho, gamma, rho_new, rpri, rdual, feas;
  };
  enum { MAXMARK = 12 };
  Run run_;
  int device_ = 0, ndim_ = 2;
  hipStream_t stream_ = nullptr;
  Grid G_;
  T ih_[3];
  std::vector<SetState<T>> sets_;
  std::vector<int32_t> owned_;
  bool finalized_ = false, feasibility_only_ = false, any_ncvx_ = false;
  int p_n_ = 0, pp_n_ = 0;
  std::vector<T> rho_, gamma_;
  std::vector<double> feas_init_;
  T *x_base_ = nullptr, *p_base_ = nullptr, *m_base_ = nullptr, *r_base_ = nullptr, *p2_base_ = nullptr, *p2_ = nullptr;
  bool cg_fused_ = false;
  long long halo_ = 0;
  T *x_ = nullptr, *xold_ = nullptr, *rhs_ = nullptr, *m_ = nullptr, *r_ = nullptr, *p_ = nullptr, *Ap_ = nullptr;
  T *Q_ = nullptr, *scr_v_ = nullptr, *scr_c_ = nullptr, *maxpart_ = nullptr;
  long long* scr_i_ = nullptr;
  long long scr_c_len_ = 0;
  T* scr_w_ = nullptr;
  T *loose_v_ = nullptr, *loose_w_ = nullptr;    // global-indexed vectors of the materialised sets of a slab-decomposed list (finalize)
  bool loose_owned_ = false, loose_whole_ = false;
  bool need_idx_ = false, need_ext_ = false;
  CdsArgs cds_;
  bool cds_full_ = false;         // SIPX_CDS_FULL=1: read all d bands of Q (no symmetric partner reads)
  bool set_streams_ = true;       // SIPX_SERIAL_SETS=1 keeps every set on the engine stream (A/B measurements)
  hipEvent_t ev_fork_ = nullptr, ev_fork2_ = nullptr;
  std::vector<hipStream_t> pool_;   // streams the sets are dealt onto, round robin
  bool set_streams_forced_ = false, search_streams_ = false;
  int search_next_ = 0;
  int n_set_streams_ = 2, pool_next_ = 0;   // measured: 2 beats 1 by 1-4 %, 3+ lose again at 512^3 (streaming passes collide)
  bool mk_ = false;               // Minkowski mode: unknowns [u; v]
  long long Nx_ = 0;              // number of unknowns (N, or 2N in Minkowski mode)
  T *w_base_ = nullptr, *w_ = nullptr;
  StencilQ<T> sq_{};
  bool stencil_q_ = false;
  double *part_cg_ = nullptr, *part_tmp_ = nullptr, *part_sets_ = nullptr;
  CgState<T>*cg_dev_ = nullptr, *cg_host_ = nullptr;
  double* hres_ = nullptr;
  volatile unsigned long long* sums_word_ = nullptr;   // pinned: sequence number of the latest reduction of the set sums (k_fin_sum)
  unsigned* sums_ticket_ = nullptr;
  unsigned long long sums_seq_ = 0;
  bool sums_by_word_ = false, word_sums_ = false;      // word_sums_: set by the whole-solve loop for the duration of a step (one rank)
  volatile int* hlean_ = nullptr;     // per set: the coming l1 search wants a sampled prediction (written by k_l1_solve)
  volatile int* hovf_ = nullptr;      // per set: the slab-decomposed search overflowed its exchange segments (k_gather_unpack)
  volatile unsigned* hverd_ = nullptr;   // per set: verdict of the speculative exchange of a slab-decomposed search (k_spec_decide)
  unsigned spec_seq_ = 0;
  long long spec_searches_ = 0, spec_fallbacks_ = 0, spec_rounds_ = 0;     // searches through the speculative exchange / of those, fallbacks / refinement rounds (all-reduces) of the fallbacks
  bool spec_exchange_ = true;         // SIPX_SPEC_EXCHANGE=0: every search through (all-reduce, ..., all-gather), as before
  // the lane of the slice-rank / nuclear-norm set (sipx_finalize, lane_start)
  int lane_set_ = -1;
  hipStream_t lane_st_ = nullptr;
  hipEvent_t lane_fork_ = nullptr, lane_ev_ = nullptr;
  T* lane_v_ = nullptr;
  std::thread lane_thr_;
  std::exception_ptr lane_err_;
  long long lane_sample_ = -1;        // index of the statistics sample that books the lane's interval (-1: none open)
  SetArgs<T> lane_args_;
  bool sweep_partial_ = false, has_loose_ = false;   // the sweep takes a subset of the sets (in_sweep); some owned set keeps its per-set kernels
  bool pass_multi_ = false;           // SIPX_PASS_MULTI=1: full first passes / fallback passes of the batched searches in one sweep per group (measured slower)
  bool sweep_plain_ = false;          // the sweep takes the plain iterations of this context: every set carries a third y / l pair
  bool feas_sample_ = true;           // SIPX_FEAS_SAMPLE=0: the feasibility searches of the batched chain start from their own last theta (A/B switch)
  bool search_batch_ = false;         // one rank + sweep: the searches of all sets as one chain of launches (batched_searches; SIPX_SEARCH_BATCH=0: per-set chains on the set streams)
  long long batch_searches_ = 0, batch_fallbacks_ = 0;
  bool spec_batch_ = true;            // SIPX_SPEC_BATCH=0: the small steps of the exchange as one kernel per set on the set streams
  bool slab_lean_multi_ = true;       // SIPX_SLAB_LEAN_MULTI=0: one lean first pass per set in the batched exchange
  T* fbuf_ = nullptr;                 // fast segments: world x two-pass sets x (fcap + header)
  long long l1_sample_runs_ = 0;
  bool l1_sample_ = true;             // SIPX_L1_SAMPLE=0: no sampled prediction of theta (A/B switch)
  bool yl_multi_ = true;              // SIPX_YL_MULTI=0: never take the one-sweep y/l update (A/B switch)
  bool x0_mode_ = false;              // s_0 = A x_0 recomputed from a snapshot of x (see finalize)
  // x lives in a RING of three buffers (round 4): the x-step writes x_{k+1} = x_k + alpha_1 p_1 into a buffer that holds neither
  // x_k nor the Barzilai-Borwein snapshot x_0 and goes on in place there, so x_k stays behind untouched as x_old (the reference's
  // copy, PARSDMM.jl:128, is never made -- the residual product used to write it: 1 N w per iteration) and the snapshot of the
  // x0 mode is simply the buffer that held x on the last BB / first iteration (1 N w per BB iteration written before).
  T *xr_base_[3] = {nullptr, nullptr, nullptr}, *xr_[3] = {nullptr, nullptr, nullptr};
  int x_cur_ = 0, x_snap_ = -1;
  bool fuse_rhs_ = false;             // the whole-solve loop: rho cannot change before the next iteration, so the sweep may write its rhs
  bool rhs_fused_ = false;            // ... and did
  std::vector<hipEvent_t> ev_;
  hipEvent_t open_ev_[4] = {nullptr, nullptr, nullptr, nullptr};
  bool open_valid_[4] = {false, false, false, false};
  int mark_step_[2] = {0, 0};          // the step whose marks sit in a parity's slots
  double mark_weight_[2] = {1.0, 1.0};
  int nmark_[2] = {0, 0};
  int mark_sec_[2][MAXMARK];
  hipEvent_t sums_event_ = nullptr;
  // sharded solve: communicator, this rank's slab [r0_, r1_) of the x-step, rows of Q it maintains
  std::unique_ptr<Comm> comm_;
  CountingComm* counting_ = nullptr;   // comm_ itself, with its call counters
  long long plane_ = 0, chunk_ = 0, r0_ = 0, r1_ = 0, qr0_ = 0, qr1_ = 0;
  int prev_ = -1, next_ = -1;
  hipStream_t cstream_ = nullptr;           // communication stream: the reduce-scatter of rhs runs beside the engine stream
  hipEvent_t ev_c_[2] = {nullptr, nullptr};
  bool rs_pending_ = false;
  double* dres_ = nullptr;                  // device copy of the reduced per-set sums (all-reduce buffer)
  hipEvent_t ev_sums_ = nullptr, ev_cgb_ = nullptr;
  bool sums_pending_ = false, defer_sums_ = false;
  // SIPX_Q_FUSED=1 (measured, NOT the default): a Q update decided at the end of an iteration of the whole-solve loop is not
  // applied at once; the residual product that opens the next x-step applies it on the fly (K::resid_qupdate, z-marching matrices
  // on one rank) into the second copy of Q, and the two copies swap roles; whoever else reads Q first flushes it through
  // k_q_update.  Bit-identical (tested), 8 N w instead of 12 for update + product -- and slower: regenerating the band values costs
  // the 512-thread march kernel more than the traffic saves (512^3: 115 -> 109 it/s), and at 256^3 the separate kernels meet
  // in the Infinity Cache (k_q_update leaves the four bands there: 46 us for 8 N w; 774 -> 732 it/s).
  bool q_fused_ = false, q_pending_ = false, q_defer_ = false;
  QArgs<T> q_pending_args_;
  T* Q2_ = nullptr;
  long long dev_bytes_ = 0;           // device bytes this context allocated (dalloc + the library-backed projectors' own buffers)
  bool slab_dist_logs_ = false;       // slab-decomposed and a distance term among the sets: obj / evol_x sums come from its y/l update
  bool lean_multi_ = false;           // k_lean_multi for the lean first passes of the l1 searches (finalize: above 2^24 grid points; SIPX_LEAN_MULTI=0/1)
  bool head_done_ = false;            // the residual product of the coming x-step is queued already (argmin_x_head)
  bool merge_sums_ = false;           // sharded whole-solve loop: the coming reduction of the set sums leaves its all-reduce to argmin_x_head
  int merged_nslots_ = 0;
  bool resid_ahead_ = true;           // SIPX_RESID_AHEAD=0: the residual product waits for the host's stop rule, as before
  int sums_flags_ = 0;
  volatile unsigned long long* ticket_ = nullptr;   // pinned: verdict of the latest CG iteration (publish_ticket)
  unsigned cg_seq_ = 0;
  // slab decomposition of the whole iteration (sipx_set_decomp): the grids the set kernels are launched on, the collectives
  // of the threshold searches, the exchange buffer of their gathered magnitudes
  bool slab_req_ = false, slab_ = false;
  hipStream_t fan_st_ = nullptr;      // the stream a gathered set's owner projects on (beside the engine stream)
  hipEvent_t fan_fork_ = nullptr;
  double* fan_ptmp_ = nullptr;
  T *fan_mpart_ = nullptr, *fan_c_ = nullptr;
  long long fan_exchanges_ = 0;       // gathers + scatters of the gathered sets so far (stats)
  bool fan_overlap_ = true;           // SIPX_FAN_OVERLAP=0: the owner projects in turn on the engine stream (A/B switch)
  bool slab_loose_ = false;           // slab-decomposed with sets projected on a materialised v (SetState::slab_ext, fan)
  // slab-decomposed with SPARSE arrays: every N-sized array of the context is backed by memory for the rank's planes (and the
  // halo planes around them) only -- see SparseBlock.  [wlo_, whi_): the grid points whose entries exist on this rank.
  bool slab_full_req_ = false, slab_local_ = false;
  // verdict of the communicator self-test of sipx_finalize ("none" without a communicator)
  std::string selftest_ = "none", mapped_why_;
  double* agree_buf_ = nullptr;
  bool selftest_mapped_failed_ = false, selftest_alltoall_failed_ = false;
  std::string a2a_why_;
  long long wlo_ = 0, whi_ = 0;
  Grid Gr_, Gyl_;
  ChainHooks hooks_;
  T* gbuf_ = nullptr;
  double *stage_ = nullptr, *sstage_ = nullptr;
  const ChainHooks* hooks() const { return slab_ ? &hooks_ : nullptr; }
  double obj_ss_ = 0, evo_ss_ = 0, xx_ss_ = 0;
  bool have_log_sums_ = false;
  hipEvent_t cg_ev_[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> stat_ev_;       // events 2 i, 2 i + 1 bracket sample i
  std::vector<KSample> samples_;
  std::vector<size_t> open_;              // samples whose closing record is still to come (launch scopes may nest)
  int stats_mode_ = 0;
  LaunchObserver obs_{this, &Engine<T>::obs_begin, &Engine<T>::obs_end};
  std::string stats_json_;
  double stat_pair_ms_ = 0;       // elapsed time between two adjacent event records (median of 16 on the idle stream)
};

EngineBase* make_engine(int dtype, int ndim, const int64_t* n, const double* h, int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    throw std::runtime_error("libsipx: no HIP device visible -- this engine has no CPU fallback");
  if (device < 0 || device >= count) throw std::runtime_error("libsipx: device index out of range");
  if (dtype == SIPX_F32) return new Engine<float>(ndim, n, h, device);
  if (dtype == SIPX_F64) return new Engine<double>(ndim, n, h, device);
  throw std::runtime_error("dtype must be SIPX_F32 or SIPX_F64");
}

template <typename T>
static void cds_spmv_T(int64_t N, int d, const void* R, const int64_t* off, const void* x, void* y) {
  if (d < 1 || d > MAXD) throw std::runtime_error("cds_spmv: band count out of range");
  long long H = 4;
  for (int b = 0; b < d; ++b) H = std::max<long long>(H, std::llabs((long long)off[b]));
  H = (H + 3) / 4 * 4;
  T* dR = dalloc<T>((size_t)N * d, false);
  T* dxb = dalloc<T>(N + 2 * H);          // zero halo on both sides
  T* dx = dxb + H;
  T* dy = dalloc<T>(N);
  SIPX_HIP(hipMemcpy(dR, R, (size_t)N * d * sizeof(T), hipMemcpyHostToDevice));
  SIPX_HIP(hipMemcpy(dx, x, N * sizeof(T), hipMemcpyHostToDevice));
  CdsArgs a;
  a.d = d;
  for (int b = 0; b < d; ++b) a.off[b] = off[b];
  Grid g;
  g.n[0] = N; g.n[1] = 1; g.n[2] = 1; g.N = N; g.st[0] = 1; g.st[1] = N; g.st[2] = N;
  K<T>::spmv(nullptr, g, N, dR, a, dx, dy);
  SIPX_HIP(hipDeviceSynchronize());
  SIPX_HIP(hipMemcpy(y, dy, N * sizeof(T), hipMemcpyDeviceToHost));
  dfree(dR); dfree(dxb); dfree(dy);
}

template <typename T>
static void resample_T(int ndim, const int64_t* nc, const int64_t* nf, const void* in, void* out) {
  long long c[3] = {1, 1, 1}, f[3] = {1, 1, 1};
  for (int q = 0; q < ndim && q < 3; ++q) { c[q] = nc[q]; f[q] = nf[q]; }
  const long long Nc = c[0] * c[1] * c[2], Nf = f[0] * f[1] * f[2];
  if (Nc < 1 || Nf < 1) throw std::runtime_error("resample: empty array");
  T* di = dalloc<T>(Nc, false);
  T* dout = dalloc<T>(Nf, false);
  SIPX_HIP(hipMemcpy(di, in, Nc * sizeof(T), hipMemcpyHostToDevice));
  resample_nn<T>(nullptr, c, f, di, dout);
  SIPX_HIP(hipDeviceSynchronize());
  SIPX_HIP(hipMemcpy(out, dout, Nf * sizeof(T), hipMemcpyDeviceToHost));
  dfree(di); dfree(dout);
}
void resample_nn_host(int dtype, int ndim, const int64_t* nc, const int64_t* nf, const void* in, void* out, int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    throw std::runtime_error("libsipx: no HIP device visible -- this engine has no CPU fallback");
  if (ndim < 1 || ndim > 3) throw std::runtime_error("resample: ndim must be 1..3");
  SIPX_HIP(hipSetDevice(device));
  if (dtype == SIPX_F32) resample_T<float>(ndim, nc, nf, in, out);
  else if (dtype == SIPX_F64) resample_T<double>(ndim, nc, nf, in, out);
  else throw std::runtime_error("dtype must be SIPX_F32 or SIPX_F64");
}

template <typename T>
static void prox_l2s_T(int64_t n, void* x, double rho, const void* m) {
  T* dx = dalloc<T>(n, false);
  T* dm = dalloc<T>(n, false);
  SIPX_HIP(hipMemcpy(dx, x, n * sizeof(T), hipMemcpyHostToDevice));
  SIPX_HIP(hipMemcpy(dm, m, n * sizeof(T), hipMemcpyHostToDevice));
  prox_l2s_dev<T>(nullptr, n, dx, dm, (T)rho);
  SIPX_HIP(hipDeviceSynchronize());
  SIPX_HIP(hipMemcpy(x, dx, n * sizeof(T), hipMemcpyDeviceToHost));
  dfree(dx); dfree(dm);
}
void prox_l2s_host(int dtype, int64_t n, void* x, double rho, const void* m, int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    throw std::runtime_error("libsipx: no HIP device visible -- this engine has no CPU fallback");
  if (n < 1) return;
  SIPX_HIP(hipSetDevice(device));
  if (dtype == SIPX_F32) prox_l2s_T<float>(n, x, rho, m);
  else if (dtype == SIPX_F64) prox_l2s_T<double>(n, x, rho, m);
  else throw std::runtime_error("dtype must be SIPX_F32 or SIPX_F64");
}

void cds_spmv_host(int dtype, int64_t N, int d, const void* R, const int64_t* off, const void* x, void* y, int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    throw std::runtime_error("libsipx: no HIP device visible -- this engine has no CPU fallback");
  SIPX_HIP(hipSetDevice(device));
  if (dtype == SIPX_F32) cds_spmv_T<float>(N, d, R, off, x, y);
  else if (dtype == SIPX_F64) cds_spmv_T<double>(N, d, R, off, x, y);
  else throw std::runtime_error("dtype must be SIPX_F32 or SIPX_F64");
}

}  // namespace sipx
