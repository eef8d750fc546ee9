// What does THIS box sustain for a streaming kernel that reads NR arrays and writes NW arrays (16 B per lane and array, grid-stride)?
// The y/l sweep of the headline list reads 13 and writes 11 N-vectors per plane; a copy is 1 + 1.  Figures in TB/s of bytes moved.
// build: hipcc -O3 --offload-arch=gfx950 tools/stream_mix_bench.hip -o scratch/stream_mix_bench ; run: scratch/stream_mix_bench [log2 elements per array, default 27]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Ptrs { float4* p[24]; };
template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void k_mix(Ptrs a, long long nvec) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      float4 v;
      if (NT) { v.x = __builtin_nontemporal_load(&a.p[r][i].x); v.y = __builtin_nontemporal_load(&a.p[r][i].y); v.z = __builtin_nontemporal_load(&a.p[r][i].z); v.w = __builtin_nontemporal_load(&a.p[r][i].w); }
      else v = a.p[r][i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      float4 o = make_float4(acc.x + w, acc.y, acc.z, acc.w);
      if (NT) { __builtin_nontemporal_store(o.x, &a.p[NR + w][i].x); __builtin_nontemporal_store(o.y, &a.p[NR + w][i].y); __builtin_nontemporal_store(o.z, &a.p[NR + w][i].z); __builtin_nontemporal_store(o.w, &a.p[NR + w][i].w); }
      else a.p[NR + w][i] = o;
    }
    if (NW == 0 && acc.x == 12345.678f) a.p[0][i] = acc;      // keep the loads alive
  }
}
template <int NR, int NW, bool NT>
static void run(const Ptrs& a, long long nvec, int wg_per_cu, const char* tag) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL((k_mix<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, a, nvec);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, a, nvec);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double bytes = (double)(NR + NW) * nvec * 16.0;
  printf("%-22s reads %2d writes %2d  %s  %2d wg/CU  %8.3f ms  %6.2f TB/s\n", tag, NR, NW, NT ? "nt " : "def", wg_per_cu, best, bytes / best / 1e9);
}
int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 27;
  const long long n = 1ll << lg, nvec = n / 4;
  Ptrs a;
  for (int k = 0; k < 24; ++k) { CK(hipMalloc(&a.p[k], n * sizeof(float))); CK(hipMemset(a.p[k], 0, n * sizeof(float))); }
  printf("elements per array 2^%d (%.0f MiB)\n", lg, n * 4.0 / 1048576.0);
  for (int wg : {4, 8, 16}) {
    run<1, 1, false>(a, nvec, wg, "copy");
    run<1, 1, true>(a, nvec, wg, "copy");
    run<6, 0, false>(a, nvec, wg, "read only");
    run<0, 6, true>(a, nvec, wg, "write only");
    run<4, 2, false>(a, nvec, wg, "cg x/r update (4+2)");
    run<13, 11, false>(a, nvec, wg, "sweep mix (13+11)");
    run<13, 11, true>(a, nvec, wg, "sweep mix (13+11)");
    run<7, 0, false>(a, nvec, wg, "lean pass (7+0)");
    run<11, 1, false>(a, nvec, wg, "rhs (10+1)");
  }
  return 0;
}
