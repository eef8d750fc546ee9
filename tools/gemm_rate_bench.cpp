// What do the products of the slice-rank projector cost in Float32 against Float64?  512 matrices of k x k times a block of b
// columns (rocBLAS strided batched GEMM), and the skinny products of a Rayleigh-Ritz step.  usage: gemm_rate_bench [k=512] [b=56] [batch=512]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { fprintf(stderr, "error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)
template <typename T> rocblas_status gemm(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const T* A, int lda, long long sa, const T* B, int ldb, long long sb, T* C, int ldc, long long sc, int batch);
template <> rocblas_status gemm<float>(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const float* A, int lda, long long sa, const float* B, int ldb, long long sb, float* C, int ldc, long long sc, int batch) {
  const float one = 1, zero = 0; return rocblas_sgemm_strided_batched(h, ta, tb, m, n, k, &one, A, lda, sa, B, ldb, sb, &zero, C, ldc, sc, batch); }
template <> rocblas_status gemm<double>(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double* A, int lda, long long sa, const double* B, int ldb, long long sb, double* C, int ldc, long long sc, int batch) {
  const double one = 1, zero = 0; return rocblas_dgemm_strided_batched(h, ta, tb, m, n, k, &one, A, lda, sa, B, ldb, sb, &zero, C, ldc, sc, batch); }
template <typename T> void run(const char* name, int k, int b, int batch) {
  rocblas_handle h; CK(rocblas_create_handle(&h));
  T *G, *Y, *Z, *H;
  CK(hipMalloc(&G, sizeof(T) * (size_t)k * k * batch)); CK(hipMalloc(&Y, sizeof(T) * (size_t)k * b * batch)); CK(hipMalloc(&Z, sizeof(T) * (size_t)k * b * batch));
  CK(hipMalloc(&H, sizeof(T) * (size_t)b * b * batch));
  CK(hipMemset(G, 0, sizeof(T) * (size_t)k * k * batch)); CK(hipMemset(Y, 0, sizeof(T) * (size_t)k * b * batch));
  auto N_ = rocblas_operation_none, T_ = rocblas_operation_transpose;
  auto time = [&](const char* what, double flop, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 20;
    for (int i = 0; i < reps; ++i) fn();
    CK(hipDeviceSynchronize());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%-8s %-28s %8.3f ms  %7.1f TFLOP/s\n", name, what, ms, flop / ms * 1e-9);
  };
  time("G Y (k x b x k)", 2.0 * k * k * b * batch, [&] { CK(gemm<T>(h, N_, N_, k, b, k, G, k, (long long)k * k, Y, k, (long long)k * b, Z, k, (long long)k * b, batch)); });
  time("Y'Z (b x b x k)", 2.0 * b * b * k * batch, [&] { CK(gemm<T>(h, T_, N_, b, b, k, Y, k, (long long)k * b, Z, k, (long long)k * b, H, b, (long long)b * b, batch)); });
  time("Y R (k x b x b)", 2.0 * k * b * b * batch, [&] { CK(gemm<T>(h, N_, N_, k, b, b, Y, k, (long long)k * b, H, b, (long long)b * b, Z, k, (long long)k * b, batch)); });
  time("X'X (k x k x k) Gram", 2.0 * k * k * k * batch, [&] { CK(gemm<T>(h, T_, N_, k, k, k, G, k, (long long)k * k, G, k, (long long)k * k, G + 0, k, (long long)k * k, 1)); CK(gemm<T>(h, T_, N_, k, k, k, G, k, (long long)k * k, G, k, (long long)k * k, Z, k, 0, 1)); });
  CK(hipFree(G)); CK(hipFree(Y)); CK(hipFree(Z)); CK(hipFree(H));
  rocblas_destroy_handle(h);
}
int main(int argc, char** argv) {
  const int k = argc > 1 ? atoi(argv[1]) : 512, b = argc > 2 ? atoi(argv[2]) : 56, batch = argc > 3 ? atoi(argv[3]) : 512;
  printf("k = %d, b = %d, batch = %d\n", k, b, batch);
  run<double>("Float64", k, b, batch);
  run<float>("Float32", k, b, batch);
  return 0;
}
