// pair_probe.hip -- x and v stream fast together when they lie in different 32 GiB regions of physical memory
// (window_probe.hip).  With separate hipMalloc blocks: x first, then candidate blocks for v, a spacer of S GiB allocated
// before each further candidate.  How many candidates until the pair is fast?
// usage: pair_probe <spacer GiB> <candidates>     Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/pair_probe profiles/pair_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BLOCK = 512;
__global__ __launch_bounds__(BLOCK) void stream(double2* __restrict__ a, double2* __restrict__ b, long long n2_env, long long chunk2) {
  const int env = blockIdx.y, blk = blockIdx.x;
  const long long begin = (long long)blk * chunk2;
  const long long end = begin + chunk2 < n2_env ? begin + chunk2 : n2_env;
  double2* ae = a + (size_t)env * n2_env; double2* be = b + (size_t)env * n2_env;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = ae[i], w = be[i];
    u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0;
    ae[i] = u; be[i] = w;
  }
}
float run(double2* a, double2* b) {
  const int envs = 64, nblk = 123, reps = 4; const long long n2_env = 500000;
  const long long chunk2 = ((n2_env + nblk - 1) / nblk + BLOCK - 1) / BLOCK * BLOCK;
  dim3 grid((unsigned)((n2_env + chunk2 - 1) / chunk2), envs);
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stream, grid, dim3(BLOCK), 0, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream, grid, dim3(BLOCK), 0, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
}
int main(int argc, char** argv) {
  const double S = argc > 1 ? atof(argv[1]) : 4.0;
  const int K = argc > 2 ? atoi(argv[2]) : 12;
  const size_t arr = (size_t)64 * 500000 * 16;
  void* x; CHK(hipMalloc(&x, arr)); CHK(hipMemset(x, 0, arr));
  std::vector<void*> keep;
  printf("spacer %.1f GiB; pair time with candidate k (us):", S);
  for (int k = 0; k < K; ++k) {
    if (k && S > 0) { void* sp; if (hipMalloc(&sp, (size_t)(S * 1024) << 20) != hipSuccess) { printf(" (spacer failed)"); break; } keep.push_back(sp); }
    void* v; CHK(hipMalloc(&v, arr)); CHK(hipMemset(v, 0, arr)); keep.push_back(v);
    printf(" %.0f", run((double2*)x, (double2*)v)); fflush(stdout);
  }
  printf("\n");
  return 0;
}
