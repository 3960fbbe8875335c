// placement_probe.hip -- does the speed of an in-place stream depend on which allocation it runs in?
// K allocations of the config-2 particle state (x and v of 64 x 1e6 float64 in one hipMalloc, as the library makes it),
// all alive at once; the same read-modify-write stream (grid 123 x 64 x 512, 16 B per lane) is timed on each, three
// rounds.  Then the same with the allocation size rounded up to a multiple of 1 GiB, and with x and v allocated apart.
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/placement_probe profiles/placement_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BLOCK = 512;

__global__ __launch_bounds__(BLOCK) void stream(double2* __restrict__ a, double2* __restrict__ b, long long n2_env, long long chunk2) {
  const int env = blockIdx.y, blk = blockIdx.x;
  const long long begin = (long long)blk * chunk2;
  const long long end = begin + chunk2 < n2_env ? begin + chunk2 : n2_env;
  double2* ae = a + (size_t)env * n2_env;
  double2* be = b + (size_t)env * n2_env;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = ae[i], w = be[i];
    u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0;
    ae[i] = u; be[i] = w;
  }
}

float run(double2* a, double2* b, int envs, long long n2_env, int reps) {
  const int nblk = 123;
  const long long chunk2 = ((n2_env + nblk - 1) / nblk + BLOCK - 1) / BLOCK * BLOCK;
  dim3 grid((unsigned)((n2_env + chunk2 - 1) / chunk2), envs);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stream, grid, dim3(BLOCK), 0, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream, grid, dim3(BLOCK), 0, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 8;
  const int envs = argc > 3 ? atoi(argv[3]) : 64;          // 12: the state fits the 256 MB Infinity Cache
  const int only = argc > 2 ? atoi(argv[2]) : -1;      // run just this allocation style (fresh process per style)
  const long long N = 1000000, n2_env = N / 2;
  const size_t arr = (size_t)envs * n2_env * 16;          // one array: 512,000,000 B
  struct Mode { const char* name; size_t alloc; bool apart; size_t gap; };
  const Mode modes[] = {
      {"x|v in one hipMalloc of 1,024,000,000 B (the library's layout)", 2 * arr, false, 0},
      {"the same, size rounded up to 1 GiB", (size_t)1 << 30, false, 0},
      {"x and v in two hipMallocs", arr, true, 0},
      {"one hipMalloc, v starts 2 MiB-aligned after x", 2 * arr + (4u << 20), false, ((arr + (2u << 20) - 1) / (2u << 20)) * (2u << 20) - arr},
      {"one hipMalloc, size rounded up to a multiple of 2 MiB", (2 * arr + (2u << 20) - 1) / (2u << 20) * (2u << 20), false, 0},
      {"one hipMalloc of 1,024,000,000 B, a 12,345,678 B allocation in front of each", 2 * arr, false, 0},
      {"hipExtMallocWithFlags(hipDeviceMallocContiguous), 1,024,000,000 B", 2 * arr, false, 0},
      {"hipExtMallocWithFlags(hipDeviceMallocUncached), 1,024,000,000 B", 2 * arr, false, 0},
  };
  int mi = -1;
  for (const Mode& m : modes) {
    ++mi;
    if (only >= 0 && mi != only) continue;
    std::vector<double2*> A, B;
    std::vector<void*> owned;
    for (int k = 0; k < K; ++k) {
      if (mi == 5) { void* junk; CHK(hipMalloc(&junk, 12345678)); owned.push_back(junk); }
      void* p;
      if (mi == 6) CHK(hipExtMallocWithFlags(&p, m.alloc, hipDeviceMallocContiguous));
      else if (mi == 7) CHK(hipExtMallocWithFlags(&p, m.alloc, hipDeviceMallocUncached));
      else CHK(hipMalloc(&p, m.alloc));
      CHK(hipMemset(p, 0, m.alloc)); owned.push_back(p);
      double2* a = static_cast<double2*>(p);
      double2* b;
      if (m.apart) { void* q; CHK(hipMalloc(&q, m.alloc)); CHK(hipMemset(q, 0, m.alloc)); owned.push_back(q); b = static_cast<double2*>(q); }
      else b = reinterpret_cast<double2*>(reinterpret_cast<char*>(p) + arr + m.gap);
      A.push_back(a); B.push_back(b);
    }
    printf("%s\n", m.name);
    for (int round = 0; round < 3; ++round) {
      printf("  round %d:", round);
      for (int k = 0; k < K; ++k) printf(" %6.1f", run(A[k], B[k], envs, n2_env, envs < 32 ? 20 : 5));
      printf(" us\n");
    }
    for (void* p : owned) CHK(hipFree(p));
  }
  return 0;
}
