// window_probe.hip -- ONE big hipMalloc block; the config-2 stream (x at w, v at w + 512,000,000 B) is timed in windows that
// slide through it in steps of S MiB.  Are there places where a window is fast, and how sharp are their edges?
// usage: window_probe <block GiB> <step MiB>      Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/window_probe profiles/window_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
  const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 56) << 30;
  const size_t step = (size_t)(argc > 2 ? atoi(argv[2]) : 256) << 20;
  const size_t arr = (size_t)64 * 500000 * 16, win = 2 * arr;
  char* base; CHK(hipMalloc((void**)&base, total)); CHK(hipMemset(base, 0, total));
  printf("block of %zu GiB at %p, windows of %.3f GB every %zu MiB (us per pass):\n", total >> 30, (void*)base, win / 1e9, step >> 20);
  int col = 0;
  for (size_t w = 0; w + win <= total; w += step) {
    printf(" %5.0f", run((double2*)(base + w), (double2*)(base + w + arr)));
    if (++col % 16 == 0) printf("   <- up to %.2f GiB\n", (double)(w + step) / (1u << 30));
    fflush(stdout);
  }
  printf("\n");
  // x in one place, v a distance D further on (both 512 MB arrays well inside slow territory when D = 0.477 GiB)
  const double Ds[] = {0.4768, 1, 2, 4, 8, 12, 16, 20, 24, 28, 30, 31, 32, 33, 34, 36, 40, 48, 56, 64, 72, 80, 96};
  for (double w_gib : {1.0, 5.0, 17.0}) {
    printf("x at %.0f GiB, v at x + D:", w_gib);
    for (double D : Ds) {
      const size_t w = (size_t)(w_gib * 1024) << 20, d = (size_t)(D * 1024.0 * 1024.0) * 1024;
      if (w + d + arr > total) break;
      printf("  D=%g: %.0f", D, run((double2*)(base + w), (double2*)(base + w + d)));
    }
    printf("\n");
  }
  return 0;
}
