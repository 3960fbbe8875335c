// vmm_layers.hip -- the map of block speeds over 240 GB (placement_probe 240) shows slow blocks almost everywhere, fast ones
// where memory has been recycled and single fast blocks about every 24 GB = one die layer of the eight 12-high HBM3E stacks
// (8 x 3 GB): a block inside one layer has that layer's banks only, a block across layers more of them.  Test: build the
// config-2 state from chunks of G bytes taken from L regions that are F bytes apart in allocation order (filler handles
// are created in between and never mapped), chunk c from region c mod L.
// usage: vmm_layers L F_GiB G_MiB      Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/vmm_layers profiles/vmm_layers.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
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
float run(double2* a, double2* b, int envs, long long n2_env) {
  const int nblk = 123, reps = 5;
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
  const int L = argc > 1 ? atoi(argv[1]) : 12;
  const double F_gib = argc > 2 ? atof(argv[2]) : 23.0;
  const size_t G = (size_t)(argc > 3 ? atof(argv[3]) * 1024 : 2048) << 10;      // MiB, fractions allowed
  const int envs = 64; const long long n2_env = 500000; const size_t arr = (size_t)envs * n2_env * 16;
  void* plain; CHK(hipMalloc(&plain, 2 * arr)); CHK(hipMemset(plain, 0, 2 * arr));
  printf("L=%d regions, %.1f GiB apart, chunks of %zu KiB.  plain hipMalloc block: %.1f us\n", L, F_gib, G >> 10,
         run((double2*)plain, (double2*)((char*)plain + arr), envs, n2_env));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  const size_t n = (2 * arr + G - 1) / G, per = (n + L - 1) / L;
  std::vector<std::vector<hipMemGenericAllocationHandle_t>> h(L);
  std::vector<hipMemGenericAllocationHandle_t> fillers;
  const size_t F = (size_t)(F_gib * 1024.0) << 20;
  for (int l = 0; l < L; ++l) {
    h[l].resize(per);
    for (size_t i = 0; i < per; ++i) CHK(hipMemCreate(&h[l][i], G, &prop, 0));
    if (l + 1 < L && F > 0) {
      hipMemGenericAllocationHandle_t f;
      hipError_t e = hipMemCreate(&f, F, &prop, 0);
      if (e != hipSuccess) { printf("filler %d failed: %s\n", l, hipGetErrorString(e)); (void)hipGetLastError(); break; }
      fillers.push_back(f);
    }
  }
  void* va; CHK(hipMemAddressReserve(&va, n * G, 0, nullptr, 0));
  for (size_t c = 0; c < n; ++c) CHK(hipMemMap(static_cast<char*>(va) + c * G, G, 0, h[c % L][c / L], 0));
  hipMemAccessDesc acc{}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  CHK(hipMemSetAccess(va, n * G, &acc, 1));
  for (auto f : fillers) CHK(hipMemRelease(f));            // the fillers only had to exist while the regions were taken
  CHK(hipMemset(va, 0, n * G));
  double2* a = static_cast<double2*>(va); double2* b = reinterpret_cast<double2*>(static_cast<char*>(va) + arr);
  printf("layered range: %.1f  %.1f  %.1f us\n", run(a, b, envs, n2_env), run(a, b, envs, n2_env), run(a, b, envs, n2_env));
  CHK(hipDeviceSynchronize());
  for (size_t c = 0; c < n; ++c) CHK(hipMemUnmap(static_cast<char*>(va) + c * G, G));
  for (auto& v : h) for (auto x : v) CHK(hipMemRelease(x));
  CHK(hipMemAddressFree(va, n * G));
  return 0;
}
