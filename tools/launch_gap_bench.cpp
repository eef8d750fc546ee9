// What does a chain of DEPENDENT small kernels cost per link on this GPU -- launched one by one on a stream, and as the nodes of
// a captured HIP graph?  (review item 6: the gaps of the 2048^2 iteration are 6 us behind every kernel of the CG loop.)
// Each kernel touches `n` floats (n = 0: an empty kernel; n = 2^22: the size of a C2 vector); times by HIP events over the chain.
//   hipcc --offload-arch=gfx950 -O2 -o scratch/launch_gap_bench tools/launch_gap_bench.cpp && scratch/launch_gap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_touch(float* a, long long n, float s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) a[i] = a[i] * s + 1.0f;
}
int main() {
  const int CHAIN = 2000;
  float* a;
  CK(hipMalloc(&a, sizeof(float) << 22));
  CK(hipMemset(a, 0, sizeof(float) << 22));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (long long n : {0ll, 1ll << 16, 1ll << 22}) {
    const int grid = n == 0 ? 1 : (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    for (int rep = 0; rep < 2; ++rep) {               // (second repetition is the measurement)
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < CHAIN; ++i) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, a, n, 0.5f);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) std::printf("n = %8lld: stream launches  %7.2f us per kernel\n", n, ms * 1e3 / CHAIN);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < CHAIN; ++i) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, a, n, 0.5f);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) std::printf("n = %8lld: graph of %d nodes %7.2f us per kernel\n", n, CHAIN, ms * 1e3 / CHAIN);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
