// Does the batched f64 eigen-decomposition behind the slice-rank projector overlap with itself?  rocsolver_dsyevd_strided_batched on
// `batch` Gram matrices of n x n: one call, against the batch split over k streams (one rocBLAS handle each) running concurrently.
// (round 2 measured 208 / 56 / 33 ms for 512 / 128 / 64 slices of 512 x 512: less than linear in the batch)
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s -> %d line %d\n", #x, (int)e_, __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512, batch = argc > 2 ? atoi(argv[2]) : 512;
  const size_t sA = (size_t)n * n;
  std::vector<double> host(sA * batch);
  unsigned s = 1;
  for (auto& v : host) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / 16777216.0 - 0.5; }
  double *A, *G, *W, *E;
  rocblas_int* info;
  CK(hipMalloc(&A, sA * batch * 8)); CK(hipMalloc(&G, sA * batch * 8));
  CK(hipMalloc(&W, (size_t)n * batch * 8)); CK(hipMalloc(&E, (size_t)n * batch * 8)); CK(hipMalloc(&info, batch * 8));
  CK(hipMemcpy(A, host.data(), sA * batch * 8, hipMemcpyHostToDevice));
  const int KMAX = 8;
  rocblas_handle h[KMAX];
  hipStream_t st[KMAX];
  for (int k = 0; k < KMAX; ++k) { CK(rocblas_create_handle(&h[k])); CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking)); CK(rocblas_set_stream(h[k], st[k])); }
  const double one = 1, zero = 0;
  auto gram = [&]() { CK(rocblas_dgemm_strided_batched(h[0], rocblas_operation_transpose, rocblas_operation_none, n, n, n, &one, A, n, sA, A, n, sA, &zero, G, n, sA, batch)); CK(hipDeviceSynchronize()); };
  for (int rep = 0; rep < 2; ++rep) {
    for (int k : {1, 2, 4, 8}) {
      gram();
      const int per = batch / k;
      double t0 = now();
      for (int j = 0; j < k; ++j)
        CK(rocsolver_dsyevd_strided_batched(h[j], rocblas_evect_original, rocblas_fill_upper, n, G + (size_t)j * per * sA, n, sA, W + (size_t)j * per * n, n,
                                            E + (size_t)j * per * n, n, info + j * per, per));
      CK(hipDeviceSynchronize());
      printf("syevd n=%d batch=%d over %d stream(s), issued from one thread: %.1f ms\n", n, batch, k, (now() - t0) * 1e3);
      if (k > 1) {      // the calls issued from k host threads (rocSOLVER's host-side loop over the columns serialises one thread's calls)
        gram();
        t0 = now();
        std::vector<std::thread> th;
        for (int j = 0; j < k; ++j)
          th.emplace_back([&, j]() {
            CK(hipSetDevice(0));
            CK(rocsolver_dsyevd_strided_batched(h[j], rocblas_evect_original, rocblas_fill_upper, n, G + (size_t)j * per * sA, n, sA, W + (size_t)j * per * n, n,
                                                E + (size_t)j * per * n, n, info + j * per, per));
            CK(hipStreamSynchronize(st[j]));
          });
        for (auto& t : th) t.join();
        printf("syevd n=%d batch=%d over %d stream(s), one host thread per stream: %.1f ms\n", n, batch, k, (now() - t0) * 1e3);
      }
    }
  }
  return 0;
}
