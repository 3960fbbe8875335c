// atomics_probe.hip -- what does it cost a sweep to hand its per-workgroup mesh to the next kernel?
// A streaming copy in the sweeps' launch geometry (grid (nblk, envs) x 512 threads, 16 B per lane of two
// arrays read and written in place) followed by one of these epilogues per workgroup:
//   0  nothing
//   1  one slab row [env][blk][Ng] of plain 8-B stores            (round 1's hand-off)
//   2  Ng no-return 64-bit integer atomic adds into [env][Ng]     (candidate: order-independent sums)
//   3  the same with returning atomics
//   4  Ng no-return fp64 atomic adds into [env][Ng]
// `--nocopy` drops the streaming loop, so that the epilogue alone is timed.
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/atomics_probe profiles/atomics_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BLOCK = 512;

template <int MODE>
__global__ __launch_bounds__(BLOCK) void probe(double2* __restrict__ a, double2* __restrict__ b, long long n2_env,
                                               long long chunk2, int Ng, int flushes, double* __restrict__ slab,
                                               unsigned long long* __restrict__ acc, double* __restrict__ accd,
                                               int copy, int interleave) {
  extern __shared__ unsigned long long mesh[];
  const int env = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  for (int c = threadIdx.x; c < Ng; c += BLOCK) mesh[c] = 0ull;
  __syncthreads();
  unsigned long long local = 0;
  if (copy) {
    double2* ae = a + (size_t)env * n2_env;
    double2* be = b + (size_t)env * n2_env;
    // contiguous: workgroup blk owns [blk chunk2, (blk+1) chunk2); interleaved: tiles blk, blk + nblk, blk + 2 nblk, ...
    long long begin = interleave ? (long long)blk * BLOCK : (long long)blk * chunk2;
    long long end = interleave ? n2_env : (begin + chunk2 < n2_env ? begin + chunk2 : n2_env);
    const long long stride = interleave ? (long long)nblk * BLOCK : BLOCK;
    for (long long i = begin + threadIdx.x; i < end; i += stride) {
      double2 u = ae[i], w = be[i];
      u.x += 1.0; u.y += 1.0; w.x += 1.0; w.y += 1.0;
      local += (unsigned long long)(long long)u.x;
      ae[i] = u;
      be[i] = w;
    }
  }
  atomicAdd(&mesh[(threadIdx.x * 7) % Ng], local | 1ull);
  __syncthreads();
  for (int f = 0; f < flushes; ++f) {
    if (MODE == 1) {
      double* row = slab + ((size_t)f * gridDim.y * nblk + (size_t)env * nblk + blk) * Ng;
      for (int c = threadIdx.x; c < Ng; c += BLOCK) row[c] = (double)mesh[c];
    } else if (MODE == 2) {
      unsigned long long* row = acc + ((size_t)f * gridDim.y + env) * Ng;
      for (int c = threadIdx.x; c < Ng; c += BLOCK) atomicAdd(&row[c], mesh[c]);
    } else if (MODE == 3) {
      unsigned long long* row = acc + ((size_t)f * gridDim.y + env) * Ng;
      unsigned long long r = 0;
      for (int c = threadIdx.x; c < Ng; c += BLOCK) r += atomicAdd(&row[c], mesh[c]);
      if (r == 0x7fffffffffffffffull) mesh[0] = r;
    } else if (MODE == 4) {
      double* row = accd + ((size_t)f * gridDim.y + env) * Ng;
      for (int c = threadIdx.x; c < Ng; c += BLOCK) atomicAdd(&row[c], (double)mesh[c]);
    }
  }
}

int g_interleave = 0;

template <int MODE>
float run(int envs, int nblk, long long n2_env, int Ng, int flushes, int copy, int reps, double2* a, double2* b,
          double* slab, unsigned long long* acc, double* accd) {
  const long long chunk2 = (n2_env + nblk - 1) / nblk;
  dim3 grid(nblk, envs);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w)
    hipLaunchKernelGGL(probe<MODE>, grid, dim3(BLOCK), Ng * 8, 0, a, b, n2_env, chunk2, Ng, flushes, slab, acc, accd, copy, g_interleave);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(probe<MODE>, grid, dim3(BLOCK), Ng * 8, 0, a, b, n2_env, chunk2, Ng, flushes, slab, acc, accd, copy, g_interleave);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  CHK(hipGetLastError());
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms / reps * 1e3f;   // us per launch
}

int main(int argc, char** argv) {
  int envs = 64, nblk = 128, Ng = 256, reps = 20;
  long long N = 1000000;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--envs")) envs = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--nblk")) nblk = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--mesh")) Ng = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--particles")) N = atoll(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--interleave")) g_interleave = 1;
  }
  const long long n2_env = N / 2;                     // double2 elements per env and array
  double2 *a, *b;
  double *slab, *accd;
  unsigned long long* acc;
  CHK(hipMalloc(&a, (size_t)envs * n2_env * 16));
  CHK(hipMalloc(&b, (size_t)envs * n2_env * 16));
  CHK(hipMemset(a, 0, (size_t)envs * n2_env * 16));
  CHK(hipMemset(b, 0, (size_t)envs * n2_env * 16));
  CHK(hipMalloc(&slab, (size_t)2 * envs * nblk * Ng * 8));
  CHK(hipMalloc(&acc, (size_t)2 * envs * Ng * 8));
  CHK(hipMalloc(&accd, (size_t)2 * envs * Ng * 8));
  CHK(hipMemset(acc, 0, (size_t)2 * envs * Ng * 8));
  CHK(hipMemset(accd, 0, (size_t)2 * envs * Ng * 8));
  printf("envs=%d nblk=%d Ng=%d N=%lld  (%.1f MB streamed r+w per launch, %d workgroups, %.2f MB per flush)\n", envs, nblk,
         Ng, N, 4.0 * envs * n2_env * 16 / 1e6, envs * nblk, (double)envs * nblk * Ng * 8 / 1e6);
  const char* names[5] = {"none", "slab row stores", "u64 atomics (no return)", "u64 atomics (returning)", "f64 atomics (no return)"};
  for (int copy = 1; copy >= 0; --copy)
    for (int flushes = 1; flushes <= 2; ++flushes)
      for (int round = 0; round < 2; ++round) {
        float t[5];
        t[0] = run<0>(envs, nblk, n2_env, Ng, flushes, copy, reps, a, b, slab, acc, accd);
        t[1] = run<1>(envs, nblk, n2_env, Ng, flushes, copy, reps, a, b, slab, acc, accd);
        t[2] = run<2>(envs, nblk, n2_env, Ng, flushes, copy, reps, a, b, slab, acc, accd);
        t[3] = run<3>(envs, nblk, n2_env, Ng, flushes, copy, reps, a, b, slab, acc, accd);
        t[4] = run<4>(envs, nblk, n2_env, Ng, flushes, copy, reps, a, b, slab, acc, accd);
        printf("copy=%d flushes=%d round=%d:", copy, flushes, round);
        for (int m = 0; m < 5; ++m) printf("  [%d %s] %.1f us", m, names[m], t[m]);
        printf("\n");
      }
  return 0;
}
