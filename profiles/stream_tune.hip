// stream_tune.hip -- how fast can an in-place read-modify-write stream of two arrays go on this box, and with what
// launch shape?  Same bytes as one sweep (envs x N x 32 B read and written).  Variables: threads per workgroup, tiles in
// flight per lane (U loads issued before the first store), workgroups per environment, workgroups per CU (capped by a
// dynamic LDS allocation), and whether a workgroup walks its chunk tile by tile or all lanes take strided tiles.
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/stream_tune profiles/stream_tune.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void stream(double2* __restrict__ a, double2* __restrict__ b, long long n2_env, long long chunk2) {
  extern __shared__ double pad[];
  const int env = blockIdx.y, blk = blockIdx.x;
  const long long begin = (long long)blk * chunk2;
  const long long end = begin + chunk2 < n2_env ? begin + chunk2 : n2_env;
  double2* ae = a + (size_t)env * n2_env;
  double2* be = b + (size_t)env * n2_env;
  if (pad && threadIdx.x == 100000) pad[0] = 0;
  long long i = begin + threadIdx.x;
  for (; i + (U - 1) * BLOCK < end; i += U * BLOCK) {
    double2 u[U], w[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { u[k] = ae[i + k * BLOCK]; w[k] = be[i + k * BLOCK]; }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      u[k].x += w[k].x; u[k].y += w[k].y; w[k].x += 1.0; w[k].y += 1.0;
      ae[i + k * BLOCK] = u[k]; be[i + k * BLOCK] = w[k];
    }
  }
  for (; i < end; i += BLOCK) {
    double2 u = ae[i], w = be[i];
    u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0;
    ae[i] = u; be[i] = w;
  }
}

template <int BLOCK, int U>
float run(int envs, int nblk, long long n2_env, int reps, int lds, double2* a, double2* b) {
  const long long chunk2 = ((n2_env + nblk - 1) / nblk + BLOCK - 1) / BLOCK * BLOCK;
  nblk = (int)((n2_env + chunk2 - 1) / chunk2);
  dim3 grid(nblk, envs);
  CHK(hipFuncSetAttribute((const void*)stream<BLOCK, U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream<BLOCK, U>), grid, dim3(BLOCK), lds, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream<BLOCK, U>), grid, dim3(BLOCK), lds, 0, a, b, n2_env, chunk2);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
}

static long long n2_env_for_pair(long long N) { return N / 2; }

int main(int argc, char** argv) {
  int envs = 64, reps = 10, contiguous = 0, pair = 0;
  long long N = 1000000;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--envs")) envs = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--particles")) N = atoll(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--pair")) pair = 1;               // x and v in two different 32 GiB regions of one 100 GiB block (the fast kind)
    else if (!strcmp(argv[i], "--contiguous")) contiguous = 1;   // one physically contiguous block (always of the slow kind)
  }
  const long long n2_env = N / 2;
  const size_t bytes = (size_t)envs * n2_env * 16;
  double2 *a, *b;
  if (pair) {
    char* base; CHK(hipMalloc((void**)&base, (size_t)100 << 30));
    a = reinterpret_cast<double2*>(base + ((size_t)1 << 30));
    float best = 1e30f;
    for (int gib : {33, 49, 65, 81, 97}) {
      double2* cand = reinterpret_cast<double2*>(base + ((size_t)gib << 30));
      CHK(hipMemset(a, 0, bytes)); CHK(hipMemset(cand, 0, bytes));
      const float t = run<512, 1>(envs, 123, n2_env_for_pair(N), reps, 0, a, cand);
      printf("v at %d GiB: %.1f us\n", gib, t);
      if (t < best) { best = t; b = cand; }
    }
  } else if (contiguous) {
    void* p; CHK(hipExtMallocWithFlags(&p, 2 * bytes, hipDeviceMallocContiguous));
    a = static_cast<double2*>(p); b = a + bytes / 16;
  } else {
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes));
  }
  CHK(hipMemset(a, 0, bytes)); CHK(hipMemset(b, 0, bytes));
  const double gb = 4.0 * bytes / 1e9;
  printf("envs=%d N=%lld: %.3f GB read+written per launch\n", envs, N, gb);
  const int nblks[] = {16, 32, 62, 123, 245, 489};
  const int ldss[] = {0, 40 * 1024, 80 * 1024, 160 * 1024};     // workgroups per CU: by waves / 4 / 2 / 1
  for (int lds : ldss)
    for (int nblk : nblks) {
      float t[9];
      t[0] = run<256, 1>(envs, nblk, n2_env, reps, lds, a, b);
      t[1] = run<256, 2>(envs, nblk, n2_env, reps, lds, a, b);
      t[2] = run<256, 4>(envs, nblk, n2_env, reps, lds, a, b);
      t[3] = run<512, 1>(envs, nblk, n2_env, reps, lds, a, b);
      t[4] = run<512, 2>(envs, nblk, n2_env, reps, lds, a, b);
      t[5] = run<512, 4>(envs, nblk, n2_env, reps, lds, a, b);
      t[6] = run<1024, 1>(envs, nblk, n2_env, reps, lds, a, b);
      t[7] = run<1024, 2>(envs, nblk, n2_env, reps, lds, a, b);
      t[8] = run<1024, 4>(envs, nblk, n2_env, reps, lds, a, b);
      printf("lds=%3dK nblk=%3d | 256thr U1/2/4 %6.1f %6.1f %6.1f | 512thr %6.1f %6.1f %6.1f | 1024thr %6.1f %6.1f %6.1f us   best %.2f TB/s\n",
             lds / 1024, nblk, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8],
             [&] { float m = t[0]; for (float x : t) m = x < m ? x : m; return gb / m * 1e3; }());
      fflush(stdout);
    }
  return 0;
}
