// Projectors that act on a materialised vector v (one block of the padded layout): library transforms
// (hipFFT / rocSOLVER / rocBLAS / hipCUB sort) and the per-fiber / per-slice selections; see ext_proj.hip.
#pragma once
#include "sipx_common.h"

namespace sipx {

enum { EXT_L1_DFT = 1, EXT_RANK = 2, EXT_NUCLEAR = 3, EXT_CARD_SEG = 4, EXT_HISTOGRAM = 5, EXT_SUBSPACE = 6, EXT_DFT_MASK = 7, EXT_DCT = 8 };

// What an ExtProj acts on: the valid extents `dims` (TD_n of the operator) inside the padded grid G (strides G.st),
// split into segments by the application mode (whole array, fibers along `dir`, slices orthogonal to `dir`).
struct ExtSpec {
  int kind = 0;
  int ndim = 3;
  Grid G;
  long long dims[3] = {1, 1, 1};
  int mode = 0, dir = 2;          // SIPX_MODE_*, 0-based direction
  double pmin = 0, pmax = 0;
  int inner = 0;                  // EXT_DCT: the SIPX_PROJ_* kind applied to the transform coefficients
  const void* lb = nullptr;       // EXT_HISTOGRAM: host TF[prod(dims)], ascending
  const void* ub = nullptr;
  const void* basis = nullptr;    // EXT_SUBSPACE: host TF[basis_rows x basis_cols], column-major
  long long basis_rows = 0;
  int basis_cols = 0, basis_orth = 0;
};

template <typename T>
struct ExtImpl;

template <typename T>
class ExtProj {
 public:
  ExtProj(const ExtSpec& spec, hipStream_t stream);
  ~ExtProj();
  ExtProj(const ExtProj&) = delete;
  // v <- P(v) in place (padded layout, G.N entries); feas selects the warm-start state of the feasibility estimate
  void project(T* v, bool feas, double* partials, T* maxpart, T* compact);
  // rank projector: calls since construction, calls served by the warm-started subspace route, calls that decomposed fully,
  // products with the Gram matrices the subspace route spent, calls whose filters ran in Float32 on the deflated matrices, and
  // those of them that went back to Float64 (all zero for the other kinds)
  void route_counts(long long out[6]) const;
  // the stream of the calls to come (the engine runs the slice-rank set of a long list on a lane of its own)
  void set_stream(hipStream_t s);
  // forget every warm start and counter: the projector behaves like a newly built one (sipx_reset)
  void reset();

 private:
  ExtImpl<T>* impl_;
};

// sum (projected - original)^2 and sum original^2 into partial slots dst[0..NB), dst[NB..2NB)
template <typename T>
void ext_dist2(hipStream_t s, long long N, const T* projected, const T* original, double* dst);

}  // namespace sipx
