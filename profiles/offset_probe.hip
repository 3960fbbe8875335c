// offset_probe.hip -- one 3 GB allocation; x at its start, v at x + 512,000,000 B + gap.  Is there a gap between the two
// arrays for which the in-place stream of both (grid 123 x 64 x 512) runs faster?  (STREAM-style array-offset tuning.)
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/offset_probe profiles/offset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BLOCK = 512;
template <int ONE>
__global__ __launch_bounds__(BLOCK) void stream(double2* __restrict__ a, double2* __restrict__ b, long long n2_env, long long chunk2) {
  const int env = blockIdx.y, blk = blockIdx.x;
  const long long begin = (long long)blk * chunk2;
  const long long end = begin + chunk2 < n2_env ? begin + chunk2 : n2_env;
  double2* ae = a + (size_t)env * n2_env;
  double2* be = b + (size_t)env * n2_env;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = ae[i];
    if (ONE) { u.x += 1.0; u.y += 1.0; ae[i] = u; }
    else { double2 w = be[i]; u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0; ae[i] = u; be[i] = w; }
  }
}
template <int ONE>
float run(double2* a, double2* b, int envs, long long n2_env) {
  const int nblk = 123, reps = 5;
  const long long chunk2 = ((n2_env + nblk - 1) / nblk + BLOCK - 1) / BLOCK * BLOCK;
  dim3 grid((unsigned)((n2_env + chunk2 - 1) / chunk2), envs);
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stream<ONE>, grid, dim3(BLOCK), 0, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream<ONE>, grid, dim3(BLOCK), 0, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
}
int main() {
  const int envs = 64; const long long n2_env = 500000; const size_t arr = (size_t)envs * n2_env * 16;
  char* base; CHK(hipMalloc((void**)&base, (size_t)3 << 30)); CHK(hipMemset(base, 0, (size_t)3 << 30));
  printf("one array, in place (half the bytes): %.1f us; at +1 GiB: %.1f us\n", run<1>((double2*)base, nullptr, envs, n2_env),
         run<1>((double2*)(base + ((size_t)1 << 30)), nullptr, envs, n2_env));
  const long long gaps[] = {0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 2097152 + 4096,
                            4194304, 12345 * 16, 24000000, 24000000 + 8192, 88 * 1048576 - 512000000 % 1048576, 536870912 - 512000000, 1073741824 - 512000000};
  for (int round = 0; round < 2; ++round)
    for (long long g : gaps) {
      printf("gap %11lld B (v - x = %11lld = %8.3f MiB): %.1f us\n", g, (long long)arr + g, (double)(arr + g) / 1048576.0,
             run<0>((double2*)base, (double2*)(base + arr + g), envs, n2_env));
      fflush(stdout);
    }
  return 0;
}
