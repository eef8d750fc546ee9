// Projectors backed by a library transform (hipFFT / rocSOLVER); see ext_proj.hip.
#pragma once
#include "sipx_common.h"

namespace sipx {

enum { EXT_L1_DFT = 1, EXT_RANK = 2 };

template <typename T>
struct ExtImpl;

template <typename T>
class ExtProj {
 public:
  ExtProj(int kind, const Grid& G, int ndim, hipStream_t stream, double pmax, int slice_dir);
  ~ExtProj();
  ExtProj(const ExtProj&) = delete;
  // v <- P(v) in place (N reals); feas selects the warm-start state of the feasibility estimate
  void project(T* v, bool feas, double* partials, T* maxpart, T* compact);

 private:
  ExtImpl<T>* impl_;
};

// sum (projected - original)^2 and sum original^2 into partial slots dst[0..NB), dst[NB..2NB)
template <typename T>
void ext_dist2(hipStream_t s, long long N, const T* projected, const T* original, double* dst);

}  // namespace sipx
