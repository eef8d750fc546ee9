// Timing of the candidate routes to the per-slice rank-r projection: gesvdj (current), Gram + syevd / syevdx / syevdj.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s -> %d line %d\n", #x, (int)e_, __LINE__); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256, batch = argc > 2 ? atoi(argv[2]) : 256, r = 32;
  rocblas_handle h;
  CK(rocblas_create_handle(&h));
  const size_t sA = (size_t)n * n;
  std::vector<double> host(sA * batch);
  unsigned s = 1;
  for (auto& v : host) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / 16777216.0 - 0.5; }
  double *A, *G, *W, *E, *U, *V, *S, *Z;
  rocblas_int *info, *nev;
  CK(hipMalloc(&A, sA * batch * 8)); CK(hipMalloc(&G, sA * batch * 8)); CK(hipMalloc(&Z, sA * batch * 8));
  CK(hipMalloc(&U, sA * batch * 8)); CK(hipMalloc(&V, sA * batch * 8));
  CK(hipMalloc(&W, (size_t)n * batch * 8)); CK(hipMalloc(&E, (size_t)n * batch * 8)); CK(hipMalloc(&S, (size_t)n * batch * 8));
  CK(hipMalloc(&info, batch * 8)); CK(hipMalloc(&nev, batch * 4));
  const double one = 1, zero = 0;
  auto reset = [&]() { CK(hipMemcpy(A, host.data(), sA * batch * 8, hipMemcpyHostToDevice)); };
  auto gram = [&]() { CK(rocblas_dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, n, n, n, &one, A, n, sA, A, n, sA, &zero, G, n, sA, batch)); };
  for (int rep = 0; rep < 2; ++rep) {
    reset(); CK(hipDeviceSynchronize());
    double t0 = now();
    CK(rocsolver_dgesvdj_strided_batched(h, rocblas_svect_singular, rocblas_svect_singular, n, n, A, n, sA, 0.0, E, 100, info + batch, S, n, U, n, sA, V, n, sA, info, batch));
    CK(hipDeviceSynchronize());
    printf("gesvdj            n=%d batch=%d: %.1f ms\n", n, batch, (now() - t0) * 1e3);
    reset(); CK(hipDeviceSynchronize());
    t0 = now(); gram(); CK(hipDeviceSynchronize());
    double tg = now() - t0;
    t0 = now();
    CK(rocsolver_dsyevd_strided_batched(h, rocblas_evect_original, rocblas_fill_upper, n, G, n, sA, W, n, E, n, info, batch));
    CK(hipDeviceSynchronize());
    printf("gram %.1f ms + syevd   : %.1f ms\n", tg * 1e3, (now() - t0) * 1e3);
    gram(); CK(hipDeviceSynchronize());
    t0 = now();
    CK(rocsolver_dsyevdx_strided_batched(h, rocblas_evect_original, rocblas_erange_index, rocblas_fill_upper, n, G, n, sA, 0.0, 0.0, n - r + 1, n, nev, W, n, Z, n, sA, info, batch));
    CK(hipDeviceSynchronize());
    printf("syevdx (top %d)        : %.1f ms\n", r, (now() - t0) * 1e3);
    gram(); CK(hipDeviceSynchronize());
    t0 = now();
    CK(rocsolver_dsyevdj_strided_batched(h, rocblas_evect_original, rocblas_fill_upper, n, G, n, sA, W, n, info, batch));
    CK(hipDeviceSynchronize());
    printf("syevdj                 : %.1f ms\n", (now() - t0) * 1e3);
    gram(); CK(hipDeviceSynchronize());
    t0 = now();
    CK(rocsolver_dsyevj_strided_batched(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper, n, G, n, sA, 0.0, E, 100, info + batch, W, n, info, batch));
    CK(hipDeviceSynchronize());
    printf("syevj                  : %.1f ms\n", (now() - t0) * 1e3);
    {   // bisection + inverse iteration for the top r only
      rocblas_int* ifail; CK(hipMalloc(&ifail, (size_t)n * batch * 4));
      gram(); CK(hipDeviceSynchronize());
      t0 = now();
      CK(rocsolver_dsyevx_strided_batched(h, rocblas_evect_original, rocblas_erange_index, rocblas_fill_upper, n, G, n, sA, 0.0, 0.0,
                                          n - r + 1, n, 0.0, nev, W, n, Z, n, sA, ifail, n, info, batch));
      CK(hipDeviceSynchronize());
      printf("syevx (top %d)         : %.1f ms\n", r, (now() - t0) * 1e3);
      CK(hipFree(ifail));
    }
    {   // one step of block subspace iteration on G with b = r + 16 vectors: Y = G X, Cholesky-QR of Y, Rayleigh-Ritz
      const int b = r + 16;
      const size_t sX = (size_t)n * b, sH = (size_t)b * b;
      double *X, *Y, *H; CK(hipMalloc(&X, sX * batch * 8)); CK(hipMalloc(&Y, sX * batch * 8)); CK(hipMalloc(&H, sH * batch * 8));
      CK(hipMemcpy(X, host.data(), sX * batch * 8, hipMemcpyHostToDevice));
      gram(); CK(hipDeviceSynchronize());
      for (int it = 0; it < 3; ++it) {
        t0 = now();
        CK(rocblas_dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, n, b, n, &one, G, n, sA, X, n, sX, &zero, Y, n, sX, batch));
        CK(hipDeviceSynchronize()); const double t1 = now();
        CK(rocblas_dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, b, b, n, &one, Y, n, sX, Y, n, sX, &zero, H, b, sH, batch));
        CK(rocsolver_dpotrf_strided_batched(h, rocblas_fill_upper, b, H, b, sH, info, batch));
        CK(rocblas_dtrsm_strided_batched(h, rocblas_side_right, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, n, b, &one, H, b, sH, Y, n, sX, batch));
        CK(hipDeviceSynchronize()); const double t2 = now();
        // Rayleigh-Ritz: H = Q' G Q, eig(H), X = Q Z
        CK(rocblas_dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, n, b, n, &one, G, n, sA, Y, n, sX, &zero, X, n, sX, batch));
        CK(rocblas_dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, b, b, n, &one, Y, n, sX, X, n, sX, &zero, H, b, sH, batch));
        CK(rocsolver_dsyevd_strided_batched(h, rocblas_evect_original, rocblas_fill_upper, b, H, b, sH, W, n, E, n, info, batch));
        CK(rocblas_dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, n, b, b, &one, Y, n, sX, H, b, sH, &zero, X, n, sX, batch));
        CK(hipDeviceSynchronize()); const double t3 = now();
        printf("subspace step b=%d: G X %.2f ms, Cholesky-QR %.2f ms, Rayleigh-Ritz %.2f ms\n", b, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
      }
      CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(H));
    }
    fflush(stdout);
  }
  return 0;
}
