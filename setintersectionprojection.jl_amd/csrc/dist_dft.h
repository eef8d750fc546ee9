// The l1-ball projector behind the unitary DFT on a grid that is decomposed by z-slab (round 5): x -> F' P_l1(F x) with the
// transform itself distributed -- what the reference's parallel mode runs on one worker (get_projector.jl:27-33 with
// project_l1_Duchi!.jl:21-52 on the coefficients; test/test_PARSDMM_parallel.jl:69-121) spread over the ranks of a slab-decomposed
// solve, so that BASELINE config 4's list needs no owner rank and no N-vector exchange:
//   2-D real-to-complex transforms of the rank's own planes (hipFFT, batched)  ->  ONE all-to-all that transposes the half spectrum
//   from z-slabs to slabs of k1 rows  ->  1-D transforms along z  ->  the l1 threshold search over the magnitudes of ALL ranks
//   (the slab collectives of every other search: all-reduced probe sums, all-gathered bracket)  ->  shrinkage  ->  the way back.
// The model is real: the half spectrum k0 = 0 .. n0/2 is transformed and the magnitudes of the coefficients whose conjugates are
// not stored count twice, as in the one-GPU projector (ext_proj.hip, k_cabs_half).
#pragma once
#include <hip/hip_runtime.h>

#include "comm.h"
#include "sipx_common.h"

namespace sipx {

template <typename T>
struct DistDftImpl;

template <typename T>
class DistDft {
 public:
  // n: the grid; this rank holds the planes [z0, z1) of it, the exchange layout gives every rank zchunk planes; radius: of the l1 ball
  DistDft(const long long n[3], long long z0, long long z1, long long zchunk, int world, int rank, double radius, hipStream_t stream);
  ~DistDft();
  DistDft(const DistDft&) = delete;
  // v: the rank's planes of the real array (n0 * n1 * (z1 - z0) entries), P(v) in place -- untouched where v lies inside the ball.
  // A collective: every rank calls it (ranks without planes included).  feas: the warm-start state of the feasibility estimate.
  void project(T* v, bool feas, Comm* comm, const ChainHooks* hooks, double* partials, T* maxpart, T* compact, long long compact_len,
               int* host_ovf);
  void set_stream(hipStream_t s);
  void reset();
  long long device_bytes() const;

 private:
  DistDftImpl<T>* impl_;
};

}  // namespace sipx
