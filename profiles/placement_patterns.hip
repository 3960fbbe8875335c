// placement_patterns.hip -- K allocations of the config-2 particle state alive at once (their speed differs by up to 16 %
// for the same kernel: placement_probe.hip); which way of handing the particles to workgroups is fast on ALL of them?
//   P0  contiguous chunks of 8 tiles (64 KiB per array and workgroup): the sweeps' geometry
//   P1  contiguous chunks of 9 tiles      P2  of 7 tiles      P3  of 8 tiles + 1/2 tile skew: not possible (tiles) -> 11 tiles
//   P4  tiles dealt cyclically to the workgroups of an environment
//   P5  chunks of 8 tiles, workgroup b starts at tile (b mod 8) of its chunk and wraps around
//   P6  chunks of 8 tiles, grid transposed (consecutive workgroup ids = consecutive environments)
// Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/placement_patterns profiles/placement_patterns.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BLOCK = 512;

// mode 0: contiguous chunk, 1: cyclic tiles, 2: contiguous with rotated start, 3: contiguous, transposed grid
template <int MODE>
__global__ __launch_bounds__(BLOCK) void stream(double2* __restrict__ a, double2* __restrict__ b, long long n2_env, int tiles_per_chunk, int nblk) {
  int env = MODE == 3 ? blockIdx.x : blockIdx.y, blk = MODE == 3 ? blockIdx.y : blockIdx.x;
  if (MODE == 4) {   // XCD-aware: workgroup ids go round-robin to the 8 XCDs; give each XCD one contiguous eighth of the work
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, W = gridDim.x * gridDim.y;
    const unsigned per = (W + 7) / 8, xcd = lin % 8, slot = lin / 8;
    const unsigned l2 = xcd * per + slot;
    if (l2 >= W || slot >= per) return;       // (W is a multiple of 8 in this probe)
    env = l2 / gridDim.x; blk = l2 % gridDim.x;
  }
  double2* ae = a + (size_t)env * n2_env;
  double2* be = b + (size_t)env * n2_env;
  const long long ntiles = (n2_env + BLOCK - 1) / BLOCK;
  if (MODE == 5) {   // chunks dealt to the workgroups by a fixed permutation over the whole buffer: the resident set is spread over all pages
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, W = gridDim.x * gridDim.y;
    const unsigned l2 = (unsigned)(((unsigned long long)lin * 2654435761ull) % W);   // a bijection when W is not a multiple of the prime factors of the constant
    env = l2 / gridDim.x; blk = l2 % gridDim.x;
    ae = a + (size_t)env * n2_env; be = b + (size_t)env * n2_env;
  }
  for (int k = 0; k < tiles_per_chunk; ++k) {
    long long t;
    if (MODE == 1) t = (long long)k * nblk + blk;
    else if (MODE == 2) t = (long long)blk * tiles_per_chunk + (k + blk) % tiles_per_chunk;
    else t = (long long)blk * tiles_per_chunk + k;
    if (t >= ntiles) { if (MODE == 2) continue; else break; }
    const long long i = t * BLOCK + threadIdx.x;
    if (i < n2_env) {
      double2 u = ae[i], w = be[i];
      u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0;
      ae[i] = u; be[i] = w;
    }
  }
}

template <int MODE>
float run(double2* a, double2* b, int envs, long long n2_env, int tpc, int reps) {
  const long long ntiles = (n2_env + BLOCK - 1) / BLOCK;
  const int nblk = (int)((ntiles + tpc - 1) / tpc);
  dim3 grid = MODE == 3 ? dim3(envs, nblk) : dim3(nblk, envs);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stream<MODE>, grid, dim3(BLOCK), 0, 0, a, b, n2_env, tpc, nblk);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream<MODE>, grid, dim3(BLOCK), 0, 0, a, b, n2_env, tpc, nblk);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int envs = 64, K = argc > 1 ? atoi(argv[1]) : 8;
  const long long N = 1000000, n2_env = N / 2;
  const size_t arr = (size_t)envs * n2_env * 16;
  std::vector<double2*> A, B;
  for (int k = 0; k < K; ++k) {
    void* p;
    if (argc > 2) CHK(hipExtMallocWithFlags(&p, 2 * arr, hipDeviceMallocContiguous));   // physically contiguous: always of the slow kind
    else CHK(hipMalloc(&p, 2 * arr));
    CHK(hipMemset(p, 0, 2 * arr));
    A.push_back(static_cast<double2*>(p));
    B.push_back(reinterpret_cast<double2*>(static_cast<char*>(p) + arr));
  }
  struct P { const char* name; int mode, tpc; };
  const P pats[] = {{"P0 chunks of 8 tiles      ", 0, 8}, {"P1 chunks of 9 tiles      ", 0, 9}, {"P2 chunks of 7 tiles      ", 0, 7},
                    {"P3 chunks of 11 tiles     ", 0, 11}, {"P4 cyclic tiles (8 each)  ", 1, 8}, {"P5 rotated start (8)      ", 2, 8},
                    {"P6 transposed grid (8)    ", 3, 8}, {"P7 transposed grid (9)    ", 3, 9}, {"P8 chunks of 4 tiles      ", 0, 4},
                    {"P9 chunks of 16 tiles     ", 0, 16}, {"P10 chunks of 13 tiles    ", 0, 13}, {"P11 cyclic tiles (4 each) ", 1, 4},
                    {"P12 XCD-aware, 8 tiles    ", 4, 8}, {"P13 XCD-aware, 4 tiles    ", 4, 4}, {"P14 XCD-aware, 16 tiles   ", 4, 16},
                    {"P15 chunks of 2 tiles     ", 0, 2}, {"P16 XCD-aware, 2 tiles    ", 4, 2},
                    {"P17 permuted chunks, 8    ", 5, 8}, {"P18 permuted chunks, 4    ", 5, 4}, {"P19 permuted chunks, 2    ", 5, 2},
                    {"P20 permuted chunks, 1    ", 5, 1}, {"P21 permuted chunks, 16   ", 5, 16}};
  for (int round = 0; round < 2; ++round)
    for (const P& p : pats) {
      printf("%s:", p.name);
      float sum = 0, mx = 0;
      for (int k = 0; k < K; ++k) {
        float t = p.mode == 0 ? run<0>(A[k], B[k], envs, n2_env, p.tpc, 5) : p.mode == 1 ? run<1>(A[k], B[k], envs, n2_env, p.tpc, 5)
                : p.mode == 2 ? run<2>(A[k], B[k], envs, n2_env, p.tpc, 5) : p.mode == 3 ? run<3>(A[k], B[k], envs, n2_env, p.tpc, 5)
                : p.mode == 4 ? run<4>(A[k], B[k], envs, n2_env, p.tpc, 5) : run<5>(A[k], B[k], envs, n2_env, p.tpc, 5);
        printf(" %6.1f", t); sum += t; mx = t > mx ? t : mx;
      }
      printf("   mean %6.1f max %6.1f us\n", sum / K, mx);
      fflush(stdout);
    }
  return 0;
}
