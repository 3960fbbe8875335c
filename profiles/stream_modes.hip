// stream_modes.hip -- which streaming pattern does this box's HBM like best?  Same bytes as one sweep of BASELINE config 2
// (64 environments x 1e6 particles x 32 B read + written), launch geometry of the sweeps (grid (nblk, envs) x 512, contiguous
// chunk per workgroup, 16 B per lane and access):
//   0  in place, two arrays (x[], v[]): the sweeps' pattern
//   1  in place, one interleaved array (x,v per particle)
//   2  out of place, two arrays -> two other arrays (ping-pong buffers)
//   3  out of place, interleaved -> interleaved
//   4  read only (both arrays, sum kept in a register)    5  write only (both arrays)
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/stream_modes profiles/stream_modes.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BLOCK = 512;

template <int MODE>
__global__ __launch_bounds__(BLOCK) void stream(double2* __restrict__ a, double2* __restrict__ b, double2* __restrict__ c,
                                                double2* __restrict__ d, long long n2_env, long long chunk2, double* sink) {
  const int env = blockIdx.y, blk = blockIdx.x;
  const long long begin = (long long)blk * chunk2;
  const long long end = begin + chunk2 < n2_env ? begin + chunk2 : n2_env;
  double acc = 0;
  if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5) {
    double2* ae = a + (size_t)env * n2_env; double2* be = b + (size_t)env * n2_env;
    double2* ce = (MODE == 2 ? c : a) + (size_t)env * n2_env; double2* de = (MODE == 2 ? d : b) + (size_t)env * n2_env;
    for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
      double2 u{1.0, 2.0}, w{3.0, 4.0};
      if (MODE != 5) { u = ae[i]; w = be[i]; }
      u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0;
      if (MODE == 4) acc += u.x + w.y; else { ce[i] = u; de[i] = w; }
    }
  } else {   // interleaved: element pair (2i, 2i+1) of one array twice as long
    double2* ae = a + (size_t)env * 2 * n2_env;
    double2* ce = (MODE == 3 ? c : a) + (size_t)env * 2 * n2_env;
    for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
      double2 u = ae[2 * i], w = ae[2 * i + 1];
      u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0;
      ce[2 * i] = u; ce[2 * i + 1] = w;
    }
  }
  if (MODE == 4 && acc == 1.2345e300) *sink = acc;
}

template <int MODE>
float run(int envs, int nblk, long long n2_env, int reps, double2* a, double2* b, double2* c, double2* d, double* sink) {
  const long long chunk2 = ((n2_env + nblk - 1) / nblk + BLOCK - 1) / BLOCK * BLOCK;
  dim3 grid(nblk, envs);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(stream<MODE>, grid, dim3(BLOCK), 0, 0, a, b, c, d, n2_env, chunk2, sink);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream<MODE>, grid, dim3(BLOCK), 0, 0, a, b, c, d, n2_env, chunk2, sink);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  int envs = 64, nblk = 123, reps = 20;
  long long N = 1000000;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--envs")) envs = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--nblk")) nblk = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--particles")) N = atoll(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
  }
  const long long n2_env = N / 2;
  const size_t bytes = (size_t)envs * n2_env * 16;   // one array
  double2 *a, *b, *c, *d; double* sink;
  CHK(hipMalloc(&a, 2 * bytes)); b = a + bytes / 16;           // a|b contiguous so that the interleaved modes can use 2x length
  CHK(hipMalloc(&c, 2 * bytes)); d = c + bytes / 16;
  CHK(hipMalloc(&sink, 8));
  CHK(hipMemset(a, 0, 2 * bytes)); CHK(hipMemset(c, 0, 2 * bytes));
  const double gb = 4.0 * bytes / 1e9;                         // read + written per launch (modes 4, 5: half)
  printf("envs=%d nblk=%d N=%lld: %.3f GB read+written per launch\n", envs, nblk, N, gb);
  const char* names[6] = {"in place, 2 arrays", "in place, interleaved", "out of place, 2 arrays", "out of place, interleaved", "read only", "write only"};
  for (int round = 0; round < 3; ++round) {
    float t[6];
    t[0] = run<0>(envs, nblk, n2_env, reps, a, b, c, d, sink);
    t[1] = run<1>(envs, nblk, n2_env, reps, a, b, c, d, sink);
    t[2] = run<2>(envs, nblk, n2_env, reps, a, b, c, d, sink);
    t[3] = run<3>(envs, nblk, n2_env, reps, a, b, c, d, sink);
    t[4] = run<4>(envs, nblk, n2_env, reps, a, b, c, d, sink);
    t[5] = run<5>(envs, nblk, n2_env, reps, a, b, c, d, sink);
    for (int m = 0; m < 6; ++m)
      printf("round %d  [%d %-26s] %8.1f us  %6.2f TB/s\n", round, m, names[m], t[m], (m >= 4 ? 0.5 : 1.0) * gb / t[m] * 1e3);
  }
  return 0;
}
