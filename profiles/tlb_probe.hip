// tlb_probe.hip -- is address translation what separates slow blocks from fast ones?  For each of K default hipMalloc blocks
// and K physically contiguous ones (always slow): (a) the in-place stream of placement_probe.hip, (b) random 64-byte reads over
// the whole block (one page-table walk per access if translations do not stick), (c) the same confined to a 16 MB window
// that moves every 4096 accesses.
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/tlb_probe profiles/tlb_probe.hip
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

// every lane reads `iters` 16-byte words at hashed positions (one 64-byte line per 4 lanes); window = bytes the addresses span
__global__ __launch_bounds__(BLOCK) void gather(const double2* __restrict__ a, unsigned long long nwords, unsigned long long window_words,
                                                int iters, double* sink) {
  unsigned long long s = (blockIdx.x * (unsigned long long)BLOCK + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  double acc = 0;
  const unsigned long long nwin = nwords / window_words;
  for (int it = 0; it < iters; ++it) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const unsigned long long win = window_words == nwords ? 0 : ((blockIdx.x + it / 64) % nwin) * window_words;
    const unsigned long long idx = win + ((s >> 20) % window_words & ~3ull) + (threadIdx.x & 3);
    acc += a[idx].x;
  }
  if (acc == 1.2345e300) *sink = acc;
}

float time_stream(double2* a, double2* b) {
  const int envs = 64, nblk = 123; const long long n2_env = 500000;
  const long long chunk2 = ((n2_env + nblk - 1) / nblk + BLOCK - 1) / BLOCK * BLOCK;
  dim3 grid((unsigned)((n2_env + chunk2 - 1) / chunk2), envs);
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stream, grid, dim3(BLOCK), 0, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(stream, grid, dim3(BLOCK), 0, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / 5 * 1e3f;
}
float time_gather(const double2* a, unsigned long long nwords, unsigned long long window_words, double* sink) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(gather, dim3(2048), dim3(BLOCK), 0, 0, a, nwords, window_words, 256, sink);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gather, dim3(2048), dim3(BLOCK), 0, 0, a, nwords, window_words, 256, sink);
  CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / 3 * 1e3f;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 6;
  const size_t arr = (size_t)64 * 500000 * 16, bytes = 2 * arr;
  const unsigned long long nwords = bytes / 16;
  double* sink; CHK(hipMalloc(&sink, 8));
  for (int kind = 0; kind < 2; ++kind) {
    std::vector<void*> blocks;
    for (int k = 0; k < K; ++k) {
      void* p;
      if (kind) CHK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous)); else CHK(hipMalloc(&p, bytes));
      CHK(hipMemset(p, 0, bytes)); blocks.push_back(p);
    }
    printf("%s\n", kind ? "physically contiguous blocks" : "default hipMalloc blocks");
    for (void* p : blocks) {
      double2* a = static_cast<double2*>(p);
      printf("  stream %6.1f us | random 16-B reads over the block %7.1f us | within moving 16 MB windows %7.1f us | within 2 MB windows %7.1f us\n",
             time_stream(a, reinterpret_cast<double2*>(static_cast<char*>(p) + arr)), time_gather(a, nwords, nwords, sink),
             time_gather(a, nwords, (16ull << 20) / 16, sink), time_gather(a, nwords, (2ull << 20) / 16, sink));
    }
    for (void* p : blocks) CHK(hipFree(p));
  }
  return 0;
}
