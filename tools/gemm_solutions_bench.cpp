// Which of rocBLAS's kernels is the fastest for the products of the slice-rank projector?  rocblas_gemm_strided_batched_ex lets the
// caller name a solution; this lists every solution that can run the shape, times each, and prints the default beside them.
// Shapes: G V (k x b x k), the filter product -- the default kernel (MT64x32x32) cuts the b = 56 columns into two tiles and reads
// every Gram matrix twice (profiles/r05_c4_512_pmc.json: 2.58 GB per launch against 1.3 GB) -- and the skinny products of a
// Rayleigh-Ritz step.   hipcc --offload-arch=gfx950 -O2 -o scratch/gemm_solutions_bench tools/gemm_solutions_bench.cpp -lrocblas
#define ROCBLAS_BETA_FEATURES_API 1
#define ROCBLAS_NO_DEPRECATED_WARNINGS 1
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { fprintf(stderr, "error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)
struct Shape { const char* name; rocblas_operation ta, tb; int m, n, k, lda, ldb, ldc; long long sa, sb, sc; };
int main(int argc, char** argv) {
  const int k = argc > 1 ? atoi(argv[1]) : 512, b = argc > 2 ? atoi(argv[2]) : 56, batch = argc > 3 ? atoi(argv[3]) : 512;
  rocblas_handle h; CK(rocblas_create_handle(&h));
  double *G, *Y, *Z, *H;
  CK(hipMalloc(&G, sizeof(double) * (size_t)k * k * batch)); CK(hipMalloc(&Y, sizeof(double) * (size_t)k * 64 * batch));
  CK(hipMalloc(&Z, sizeof(double) * (size_t)k * 64 * batch)); CK(hipMalloc(&H, sizeof(double) * (size_t)64 * 64 * batch));
  {
    std::vector<double> hg((size_t)k * k * batch), hy((size_t)k * 64 * batch), hh((size_t)64 * 64 * batch);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
    for (auto& v : hg) v = rnd();
    for (auto& v : hy) v = rnd();
    for (auto& v : hh) v = rnd();
    CK(hipMemcpy(G, hg.data(), hg.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(Y, hy.data(), hy.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(H, hh.data(), hh.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  const auto N_ = rocblas_operation_none, T_ = rocblas_operation_transpose;
  const long long sG = (long long)k * k, sY = (long long)k * b, sH = (long long)b * b;
  std::vector<Shape> shapes = {
      {"G V  (k x b x k)", N_, N_, k, b, k, k, k, k, sG, sY, sY},
      {"V'G  (b x k x k: the transposed form)", T_, N_, b, k, k, k, k, b, sY, sG, sY},
      {"Y'Z  (b x b x k)", T_, N_, b, b, k, k, k, b, sY, sY, sH},
      {"Y R  (k x b x b)", N_, N_, k, b, b, k, b, k, sY, sH, sY},
  };
  const double one = 1.0, zero = 0.0;
  for (const Shape& s : shapes) {
    const double *A = s.ta == N_ && s.m == k && s.k == k ? G : Y, *B = (s.k == k && s.n == k) ? G : (s.k == b ? H : Y);
    double* C = (s.m == b && s.n == b) ? H : Z;
    auto run = [&](int sol) {
      return rocblas_gemm_strided_batched_ex(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, rocblas_datatype_f64_r, s.lda, s.sa, B, rocblas_datatype_f64_r, s.ldb,
                                             s.sb, &zero, C, rocblas_datatype_f64_r, s.ldc, s.sc, C, rocblas_datatype_f64_r, s.ldc, s.sc, batch,
                                             rocblas_datatype_f64_r, sol ? rocblas_gemm_algo_solution_index : rocblas_gemm_algo_standard, sol, 0);
    };
    auto time = [&](int sol) {
      for (int i = 0; i < 2; ++i) if (run(sol) != rocblas_status_success) return -1.0;
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      const int reps = 10;
      for (int i = 0; i < reps; ++i) run(sol);
      CK(hipDeviceSynchronize());
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    rocblas_int ns = 0;
    CK(rocblas_gemm_strided_batched_ex_get_solutions(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, rocblas_datatype_f64_r, s.lda, s.sa, B, rocblas_datatype_f64_r, s.ldb,
                                                     s.sb, &zero, C, rocblas_datatype_f64_r, s.ldc, s.sc, C, rocblas_datatype_f64_r, s.ldc, s.sc, batch,
                                                     rocblas_datatype_f64_r, rocblas_gemm_algo_solution_index, 0, nullptr, &ns));
    std::vector<rocblas_int> sols(ns);
    CK(rocblas_gemm_strided_batched_ex_get_solutions(h, s.ta, s.tb, s.m, s.n, s.k, &one, A, rocblas_datatype_f64_r, s.lda, s.sa, B, rocblas_datatype_f64_r, s.ldb,
                                                     s.sb, &zero, C, rocblas_datatype_f64_r, s.ldc, s.sc, C, rocblas_datatype_f64_r, s.ldc, s.sc, batch,
                                                     rocblas_datatype_f64_r, rocblas_gemm_algo_solution_index, 0, sols.data(), &ns));
    const double flop = 2.0 * s.m * s.n * s.k * batch;
    const double t_def = time(0);
    std::vector<std::pair<double, int>> res;
    for (int sol : sols) { const double t = time(sol); if (t > 0) res.push_back({t, sol}); }
    std::sort(res.begin(), res.end());
    // are the fastest solutions bit-identical to the default one?  (same MFMA instruction, no split of k: the same order of additions)
    const size_t nc = (size_t)s.sc * batch;
    std::vector<double> ref(nc), out(nc);
    CK(run(0)); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), C, nc * sizeof(double), hipMemcpyDeviceToHost));
    printf("%s, batch %d: %d solutions; default %.3f ms (%.1f TFLOP/s)\n", s.name, batch, ns, t_def, flop / t_def * 1e-9);
    for (size_t i = 0; i < res.size() && i < 6; ++i) {
      CK(hipMemset(C, 0, nc * sizeof(double)));
      CK(run(res[i].second)); CK(hipDeviceSynchronize());
      CK(hipMemcpy(out.data(), C, nc * sizeof(double), hipMemcpyDeviceToHost));
      size_t diff = 0; double maxd = 0;
      for (size_t q = 0; q < nc; ++q) if (out[q] != ref[q]) { ++diff; maxd = std::max(maxd, std::abs(out[q] - ref[q])); }
      printf("    solution %6d  %.3f ms  (%.1f TFLOP/s)  %s (%zu of %zu entries differ, max %.3g)\n", res[i].second, res[i].first, flop / res[i].first * 1e-9,
             diff ? "DIFFERS from the default" : "same bits as the default", diff, nc, maxd);
    }
    fflush(stdout);
  }
  return 0;
}
