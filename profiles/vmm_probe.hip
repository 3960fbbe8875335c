// vmm_probe.hip -- physically contiguous blocks stream SLOWER than scattered ones (placement_probe.hip: every
// hipDeviceMallocContiguous block 389 us, default blocks 337-391 us).  Can the scatter be made on purpose?  The config-2
// particle state (1,024,000,000 B, rounded up to the granularity) is built from physical chunks of G bytes (hipMemCreate)
// mapped into one virtual range (hipMemAddressReserve / hipMemMap) in a chosen order:
//   identity      chunk i of the range = i-th chunk created
//   permuted      a fixed pseudo-random permutation of the same chunks
//   strided S     S times as many chunks are created, the range takes every S-th one (the others stay allocated)
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/vmm_probe profiles/vmm_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
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

struct Mapped {
  void* va = nullptr; size_t size = 0; size_t G = 0;
  std::vector<hipMemGenericAllocationHandle_t> all;
};

Mapped build(size_t bytes, size_t G, int mode, int stride, int dev) {
  Mapped m;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  const size_t n = (bytes + G - 1) / G;
  m.size = n * G; m.G = G;
  const size_t total = n * (size_t)stride;
  m.all.resize(total);
  for (size_t i = 0; i < total; ++i) CHK(hipMemCreate(&m.all[i], G, &prop, 0));
  CHK(hipMemAddressReserve(&m.va, m.size, 0, nullptr, 0));
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i * stride;
  if (mode == 1) {   // fixed pseudo-random permutation (LCG Fisher-Yates)
    unsigned long long s = 88172645463325252ull;
    for (size_t i = n - 1; i > 0; --i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      std::swap(order[i], order[(s >> 33) % (i + 1)]);
    }
  }
  for (size_t i = 0; i < n; ++i) CHK(hipMemMap(static_cast<char*>(m.va) + i * G, G, 0, m.all[order[i]], 0));
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CHK(hipMemSetAccess(m.va, m.size, &acc, 1));
  CHK(hipMemset(m.va, 0, m.size));
  return m;
}

void destroy(Mapped& m) {
  CHK(hipDeviceSynchronize());
  for (size_t off = 0; off < m.size; off += m.G) CHK(hipMemUnmap(static_cast<char*>(m.va) + off, m.G));
  for (auto h : m.all) CHK(hipMemRelease(h));
  CHK(hipMemAddressFree(m.va, m.size));
}

int main(int argc, char** argv) {
  const bool together = argc > 1;
  const int envs = 64; const long long n2_env = 500000; const size_t arr = (size_t)envs * n2_env * 16;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  CHK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CHK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
  void* plain; CHK(hipMalloc(&plain, 2 * arr)); CHK(hipMemset(plain, 0, 2 * arr));
  printf("plain hipMalloc block: %.1f us\n", run((double2*)plain, (double2*)((char*)plain + arr), envs, n2_env, 5));
  struct Case { const char* name; size_t G; int mode, stride; };
  const Case cases[] = {
      {"2 MiB chunks, identity", (size_t)2 << 20, 0, 1}, {"2 MiB chunks, permuted", (size_t)2 << 20, 1, 1},
      {"2 MiB chunks, every 4th of 4x as many", (size_t)2 << 20, 0, 4}, {"2 MiB chunks, every 4th, permuted", (size_t)2 << 20, 1, 4},
      {"64 MiB chunks, identity", (size_t)64 << 20, 0, 1}, {"64 MiB chunks, permuted", (size_t)64 << 20, 1, 1},
      {"64 MiB chunks, every 4th", (size_t)64 << 20, 0, 4},
      {"one 1 GiB chunk", (size_t)1 << 30, 0, 1},
  };
  // the cache regime (12 environments, 192 MB of particles inside the 256 MB Infinity Cache)
  {
    const size_t arr12 = (size_t)12 * n2_env * 16;
    Mapped m = build(2 * arr12, (size_t)2 << 20, 0, 1, 0);
    printf("12 envs: plain %.1f %.1f us, mapped (2 MiB chunks) %.1f %.1f us\n", run((double2*)plain, (double2*)((char*)plain + arr12), 12, n2_env, 20),
           run((double2*)plain, (double2*)((char*)plain + arr12), 12, n2_env, 20),
           run((double2*)m.va, (double2*)((char*)m.va + arr12), 12, n2_env, 20), run((double2*)m.va, (double2*)((char*)m.va + arr12), 12, n2_env, 20));
    destroy(m);
  }
  // several mapped ranges and several plain blocks alive at once
  if (together) {
    std::vector<Mapped> ms; std::vector<void*> ps;
    for (int k = 0; k < 6; ++k) {
      ms.push_back(build(2 * arr, k % 2 ? (size_t)2 << 20 : (size_t)1 << 30, 0, 1, 0));
      void* q; CHK(hipMalloc(&q, 2 * arr)); CHK(hipMemset(q, 0, 2 * arr)); ps.push_back(q);
    }
    printf("six mapped ranges (1 GiB chunk / 2 MiB chunks alternating):");
    for (auto& m : ms) printf(" %.1f", run((double2*)m.va, (double2*)((char*)m.va + arr), envs, n2_env, 5));
    printf(" us\nsix plain hipMalloc blocks made in between:           ");
    for (void* q : ps) printf(" %.1f", run((double2*)q, (double2*)((char*)q + arr), envs, n2_env, 5));
    printf(" us\n");
    for (auto& m : ms) destroy(m);
    for (void* q : ps) CHK(hipFree(q));
  }
  for (const Case& c : cases) {
    if (c.G % grec != 0 && c.G % gmin != 0) { printf("%s: granularity not allowed\n", c.name); continue; }
    Mapped m = build(2 * arr, c.G, c.mode, c.stride, 0);
    double2* a = static_cast<double2*>(m.va);
    double2* b = reinterpret_cast<double2*>(static_cast<char*>(m.va) + arr);
    printf("%-40s: %.1f  %.1f  %.1f us\n", c.name, run(a, b, envs, n2_env, 5), run(a, b, envs, n2_env, 5), run(a, b, envs, n2_env, 5));
    fflush(stdout);
    destroy(m);
  }
  return 0;
}
