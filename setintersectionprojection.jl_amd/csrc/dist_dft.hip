// The slab-decomposed l1-DFT projector (dist_dft.h).  gfx950 only; hipFFT for the transforms, the engine's communicator for the
// transposition, the slab search machinery of kernels_proj.hip for the threshold.
#include "dist_dft.h"

#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "engine.h"

namespace sipx {

namespace {

template <typename T>
struct Cx {
  T re, im;
};

void fftc(hipfftResult r, const char* what) {
  if (r != HIPFFT_SUCCESS) throw std::runtime_error(std::string("hipFFT (slab-decomposed DFT): ") + what + " failed, code " + std::to_string((int)r));
}

// A[p][k1][k0] (the rank's planes, all k1 rows) -> S[d][p][k1 - d c1][k0]: the rows that rank d = k1 / c1 will transform along z.
// (planes p >= pz and rows beyond n1 of a destination's block are never read on the other side)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dft_pack(long long pz, long long n1, long long nh0, long long zc, long long c1,
                                                    const Cx<T>* __restrict__ A, Cx<T>* __restrict__ S) {
  const long long tot = pz * n1 * nh0;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < tot; e += (long long)gridDim.x * BLOCK) {
    const long long k0 = e % nh0, t = e / nh0, k1 = t % n1, p = t / n1;
    const long long d = k1 / c1, j = k1 - d * c1;
    S[((d * zc + p) * c1 + j) * nh0 + k0] = A[e];
  }
}
// ... and back: S[d][p][j][k0], received from rank d (its rows, this rank's planes) -> A[p][d c1 + j][k0]
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dft_unpack(long long pz, long long n1, long long nh0, long long zc, long long c1,
                                                      const Cx<T>* __restrict__ S, Cx<T>* __restrict__ A) {
  const long long tot = pz * n1 * nh0;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < tot; e += (long long)gridDim.x * BLOCK) {
    const long long k0 = e % nh0, t = e / nh0, k1 = t % n1, p = t / n1;
    const long long d = k1 / c1, j = k1 - d * c1;
    A[e] = S[((d * zc + p) * c1 + j) * nh0 + k0];
  }
}
// Magnitudes of the coefficients this rank holds after the transposition -- R[z][j][k0], z < n2, j < m1 of the c1 rows of a block --
// strung together without the unused rows: mag[(z m1 + j) nh0 + k0]; those of k0 = 1 .. ndup, whose conjugates are not stored, a
// second time behind them (ext_proj.hip, k_cabs_half): the search sees this rank's share of all N coefficients.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dft_abs(long long n2, long long m1, long long c1, long long nh0, long long ndup,
                                                   const Cx<T>* __restrict__ R, T* __restrict__ mag) {
  const long long Nh = n2 * m1 * nh0;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < Nh; e += (long long)gridDim.x * BLOCK) {
    const long long k0 = e % nh0, row = e / nh0, j = row % m1, z = row / m1;
    const Cx<T> c = R[(z * c1 + j) * nh0 + k0];
    const T m = (T)hypot((double)c.re, (double)c.im);
    mag[e] = m;
    if (k0 >= 1 && k0 <= ndup) mag[Nh + row * ndup + (k0 - 1)] = m;
  }
}
// z <- sign(z) max(|z| - theta, 0)   (project_l1_Duchi!.jl:49 on complex input; ext_proj.hip, k_csoft)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dft_soft(long long n2, long long m1, long long c1, long long nh0, Cx<T>* __restrict__ R,
                                                    const T* __restrict__ mag, const ProjScalars<T>* __restrict__ ps) {
  if (!ps->need) return;
  const T th = ps->theta;
  const long long Nh = n2 * m1 * nh0;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < Nh; e += (long long)gridDim.x * BLOCK) {
    const long long k0 = e % nh0, row = e / nh0, j = row % m1, z = row / m1;
    const T a = mag[e];
    T t = a - th;
    t = t > T(0) ? t : T(0);
    const T f = a > T(0) ? t / a : T(0);
    Cx<T>& c = R[(z * c1 + j) * nh0 + k0];
    c.re = c.re * f;
    c.im = c.im * f;
  }
}
// v <- w / N unless v already lies inside the ball (F'F = I: the round trip would only add rounding noise; ext_proj.hip, k_unpack)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_dft_store(long long n, const T* __restrict__ w, T* __restrict__ v, T scale,
                                                     const ProjScalars<T>* __restrict__ ps) {
  if (!ps->need) return;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < n; e += (long long)gridDim.x * BLOCK) v[e] = w[e] * scale;
}

}  // namespace

template <typename T>
struct DistDftImpl {
  long long n[3] = {1, 1, 1}, N = 0;
  long long z0 = 0, z1 = 0, pz = 0, zc = 0;
  long long nh0 = 0, ndup = 0, c1 = 0, m1 = 0;
  int world = 1, rank = 0;
  long long chunk = 0;              // complex entries a rank sends to (and receives from) each rank
  hipStream_t stream = nullptr;
  hipfftHandle p_fwd2 = 0, p_inv2 = 0, p_z = 0;
  bool have2 = false, havez = false;
  Cx<T>*A = nullptr, *S = nullptr, *R = nullptr, *X = nullptr;
  T *mag = nullptr, *wr = nullptr;
  // scratch of the threshold search, this projector's own (partial slots, per-block extrema, the compaction buffer, the exchange
  // segments of the bracket): nothing it leaves behind can be read by another set's search
  double* sp = nullptr;
  T *sm = nullptr, *sc = nullptr, *sg = nullptr;
  long long sc_len = 0, sg_len = 0;
  ProjScalars<T>*ps = nullptr, *psf = nullptr;
  T radius_raw = 0;
  long long bytes = 0;
  template <typename U>
  U* alloc(long long count) {
    U* p = nullptr;
    if (count <= 0) count = 1;
    SIPX_HIP(hipMalloc((void**)&p, (size_t)count * sizeof(U)));
    // (on the projector's own stream: a fill on the null stream is not ordered against a non-blocking stream -- the initialisation
    //  kernel of the search state that follows would race with it, and a rank whose state came out all zero takes other decisions
    //  than the ranks it shares every collective with)
    SIPX_HIP(hipMemsetAsync(p, 0, (size_t)count * sizeof(U), stream));
    bytes += count * (long long)sizeof(U);
    if (long long* t = alloc_tally()) *t += count * (long long)sizeof(U);
    return p;
  }
};

template <typename T>
static void dist_dft_release(DistDftImpl<T>* impl) {
  DistDftImpl<T>& I = *impl;
  if (I.p_fwd2) (void)hipfftDestroy(I.p_fwd2);
  if (I.p_inv2) (void)hipfftDestroy(I.p_inv2);
  if (I.p_z) (void)hipfftDestroy(I.p_z);
  for (void* p : {(void*)I.A, (void*)I.S, (void*)I.R, (void*)I.X, (void*)I.mag, (void*)I.wr, (void*)I.ps, (void*)I.psf, (void*)I.sp, (void*)I.sm,
                  (void*)I.sc, (void*)I.sg})
    if (p) (void)hipFree(p);
  delete impl;
}
template <typename T>
DistDft<T>::DistDft(const long long n[3], long long z0, long long z1, long long zchunk, int world, int rank, double radius, hipStream_t stream)
    : impl_(new DistDftImpl<T>()) {
  DistDftImpl<T>& I = *impl_;
  try {
  for (int a = 0; a < 3; ++a) I.n[a] = n[a];
  I.N = n[0] * n[1] * n[2];
  I.z0 = z0; I.z1 = z1 > z0 ? z1 : z0; I.pz = I.z1 - I.z0; I.zc = zchunk;
  I.world = world; I.rank = rank; I.stream = stream;
  if (!(radius > 0)) throw std::runtime_error("Radius of L1 ball is negative");
  if (n[0] < 2 || n[2] < 1 || zchunk < 1 || zchunk * world < n[2]) throw std::runtime_error("slab-decomposed DFT: grid / slab layout out of range");
  I.nh0 = n[0] / 2 + 1;
  I.ndup = n[0] - I.nh0;
  I.c1 = (n[1] + world - 1) / world;
  const long long a1 = std::min<long long>(n[1], (long long)rank * I.c1), b1 = std::min<long long>(n[1], (long long)(rank + 1) * I.c1);
  I.m1 = b1 - a1;
  I.chunk = I.zc * I.c1 * I.nh0;
  I.stream = stream;
  I.A = I.template alloc<Cx<T>>(I.pz * n[1] * I.nh0);
  I.S = I.template alloc<Cx<T>>(I.chunk * world);
  I.R = I.template alloc<Cx<T>>(I.chunk * world);
  I.X = I.template alloc<Cx<T>>(I.chunk * world);
  I.mag = I.template alloc<T>(n[2] * I.c1 * n[0]);
  I.wr = I.template alloc<T>(I.pz * n[1] * n[0]);
  I.ps = I.template alloc<ProjScalars<T>>(1);
  I.psf = I.template alloc<ProjScalars<T>>(1);
  K<T>::ps_init(stream, I.ps, nullptr);
  K<T>::ps_init(stream, I.psf, nullptr);
  SIPX_HIP(hipStreamSynchronize(stream));
  I.radius_raw = (T)(radius * sqrt((double)I.N));        // ||F_unitary v||_1 <= b  <=>  ||FFT v||_1 <= b sqrt(N)
  const bool dbl = sizeof(T) == 8;
  if (I.pz > 0) {
    int nn[2] = {(int)n[1], (int)n[0]};
    fftc(hipfftPlanMany(&I.p_fwd2, 2, nn, nullptr, 1, 0, nullptr, 1, 0, dbl ? HIPFFT_D2Z : HIPFFT_R2C, (int)I.pz), "plan (2-D, real to complex)");
    fftc(hipfftPlanMany(&I.p_inv2, 2, nn, nullptr, 1, 0, nullptr, 1, 0, dbl ? HIPFFT_Z2D : HIPFFT_C2R, (int)I.pz), "plan (2-D, complex to real)");
    fftc(hipfftSetStream(I.p_fwd2, stream), "set stream");
    fftc(hipfftSetStream(I.p_inv2, stream), "set stream");
    I.have2 = true;
  }
  if (I.m1 > 0) {
    int nz[1] = {(int)n[2]};
    int emb[1] = {(int)n[2]};
    const int stride = (int)(I.c1 * I.nh0);
    fftc(hipfftPlanMany(&I.p_z, 1, nz, emb, stride, 1, emb, stride, 1, dbl ? HIPFFT_Z2Z : HIPFFT_C2C, (int)(I.m1 * I.nh0)), "plan (1-D along z)");
    fftc(hipfftSetStream(I.p_z, stream), "set stream");
    I.havez = true;
  }
  } catch (...) {          // (an allocation or a plan failed: nothing of a half-built projector stays behind)
    dist_dft_release(impl_);
    impl_ = nullptr;
    throw;
  }
}

template <typename T>
DistDft<T>::~DistDft() { dist_dft_release(impl_); }

template <typename T>
void DistDft<T>::set_stream(hipStream_t s) {
  DistDftImpl<T>& I = *impl_;
  I.stream = s;
  if (I.have2) { fftc(hipfftSetStream(I.p_fwd2, s), "set stream"); fftc(hipfftSetStream(I.p_inv2, s), "set stream"); }
  if (I.havez) fftc(hipfftSetStream(I.p_z, s), "set stream");
}

template <typename T>
void DistDft<T>::reset() {
  DistDftImpl<T>& I = *impl_;
  K<T>::ps_init(I.stream, I.ps, nullptr);
  K<T>::ps_init(I.stream, I.psf, nullptr);
}

template <typename T>
long long DistDft<T>::device_bytes() const { return impl_->bytes; }

template <typename T>
void DistDft<T>::project(T* v, bool feas, Comm* comm, const ChainHooks* hooks, double* partials, T* maxpart, T* compact,
                         long long compact_len, int* host_ovf) {
  DistDftImpl<T>& I = *impl_;
  hipStream_t s = I.stream;
  const bool dbl = sizeof(T) == 8;
  const int dt = dbl ? SIPX_F64 : SIPX_F32;
  ProjScalars<T>* ps = feas ? I.psf : I.ps;
  const long long nloc = I.pz * I.n[1] * I.n[0];
  // ---- forward: planes -> rows
  if (I.pz > 0) {
    if (dbl) fftc(hipfftExecD2Z(I.p_fwd2, (hipfftDoubleReal*)v, (hipfftDoubleComplex*)I.A), "forward (2-D)");
    else fftc(hipfftExecR2C(I.p_fwd2, (hipfftReal*)v, (hipfftComplex*)I.A), "forward (2-D)");
    hipLaunchKernelGGL((k_dft_pack<T>), dim3(fit_grid(I.pz * I.n[1] * I.nh0, NB)), dim3(BLOCK), 0, s, I.pz, I.n[1], I.nh0, I.zc, I.c1, I.A, I.S);
  }
  comm->alltoall(I.S, I.R, I.X, (size_t)(2 * I.chunk), dt, s);
  if (I.m1 > 0) {
    if (dbl) fftc(hipfftExecZ2Z(I.p_z, (hipfftDoubleComplex*)I.R, (hipfftDoubleComplex*)I.R, HIPFFT_FORWARD), "forward (along z)");
    else fftc(hipfftExecC2C(I.p_z, (hipfftComplex*)I.R, (hipfftComplex*)I.R, HIPFFT_FORWARD), "forward (along z)");
    hipLaunchKernelGGL((k_dft_abs<T>), dim3(fit_grid(I.n[2] * I.m1 * I.nh0, NB)), dim3(BLOCK), 0, s, I.n[2], I.m1, I.c1, I.nh0, I.ndup, I.R, I.mag);
  }
  // ---- the threshold: this rank's share of the N magnitudes, the sums and the bracket through the slab collectives
  (void)partials; (void)maxpart; (void)compact; (void)compact_len;
  if (!I.sp) {
    I.sp = I.template alloc<double>((long long)(PREP_SLOTS + 2) * NB);
    I.sm = I.template alloc<T>(2 * NB);
    I.sc_len = std::max<long long>(I.n[2] * I.c1 * I.n[0], hooks->gcap + 64) + 64;
    I.sc = I.template alloc<T>(I.sc_len);
    I.sg_len = (long long)I.world * (hooks->gcap + GATHER_HDR);
    I.sg = I.template alloc<T>(I.sg_len);
  }
  ChainHooks hk = *hooks;
  hk.gbuf = I.sg;
  K<T>::proj_scalars_arr_slab(s, I.n[2] * I.m1 * I.n[0], I.mag, PX_L1, T(0), I.radius_raw, ps, I.sp, I.sm, I.sc, I.N, &hk, I.sc_len, host_ovf);
  static const bool dbg = getenv("SIPX_DFT_DEBUG") != nullptr;      // the state the search ended in, per rank, on stderr (synchronises)
  if (dbg) {
    ProjScalars<T> h;
    SIPX_HIP(hipStreamSynchronize(s));
    SIPX_HIP(hipMemcpy(&h, ps, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[sipx dft] rank %d feas %d: need %d theta %.9g asum %.9g vmax %.9g bracket (%.9g, %.9g] spec (%.9g, %.9g] spec_ok %d ovf %d gather_ovf %d n_compact %llu refine %d rounds %d\n",
            I.rank, (int)feas, h.need, (double)h.theta, h.asum, (double)h.vmax, h.lo, h.hi, h.spec_lo, h.spec_hi, h.spec_ok, h.spec_overflow, h.gather_overflow,
            (unsigned long long)h.n_compact, h.refine, h.rounds_used);
  }
  // ---- shrinkage and the way back (every rank makes the same calls whether or not v lies inside the ball: the collectives match;
  //      inside the ball nothing is stored at the end)
  if (I.m1 > 0) {
    hipLaunchKernelGGL((k_dft_soft<T>), dim3(fit_grid(I.n[2] * I.m1 * I.nh0, NB)), dim3(BLOCK), 0, s, I.n[2], I.m1, I.c1, I.nh0, I.R, I.mag, ps);
    if (dbl) fftc(hipfftExecZ2Z(I.p_z, (hipfftDoubleComplex*)I.R, (hipfftDoubleComplex*)I.R, HIPFFT_BACKWARD), "inverse (along z)");
    else fftc(hipfftExecC2C(I.p_z, (hipfftComplex*)I.R, (hipfftComplex*)I.R, HIPFFT_BACKWARD), "inverse (along z)");
  }
  comm->alltoall(I.R, I.S, I.X, (size_t)(2 * I.chunk), dt, s);
  if (I.pz > 0) {
    hipLaunchKernelGGL((k_dft_unpack<T>), dim3(fit_grid(I.pz * I.n[1] * I.nh0, NB)), dim3(BLOCK), 0, s, I.pz, I.n[1], I.nh0, I.zc, I.c1, I.S, I.A);
    if (dbl) fftc(hipfftExecZ2D(I.p_inv2, (hipfftDoubleComplex*)I.A, (hipfftDoubleReal*)I.wr), "inverse (2-D)");
    else fftc(hipfftExecC2R(I.p_inv2, (hipfftComplex*)I.A, (hipfftReal*)I.wr), "inverse (2-D)");
    hipLaunchKernelGGL((k_dft_store<T>), dim3(fit_grid(nloc, NB)), dim3(BLOCK), 0, s, nloc, I.wr, v, (T)(1.0 / (double)I.N), ps);
  }
  SIPX_HIP(hipGetLastError());
}

template class DistDft<float>;
template class DistDft<double>;

}  // namespace sipx
