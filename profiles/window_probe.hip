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
// the same stream with the environments in G groups, each group's x and v at their own addresses
struct Bases { double2* a[8]; double2* b[8]; int per; };
__global__ __launch_bounds__(BLOCK) void stream_groups(Bases bs, long long n2_env, long long chunk2) {
  const int env = blockIdx.y, blk = blockIdx.x, g = env / bs.per, e = env % bs.per;
  const long long begin = (long long)blk * chunk2;
  const long long end = begin + chunk2 < n2_env ? begin + chunk2 : n2_env;
  double2* ae = bs.a[g] + (size_t)e * n2_env; double2* be = bs.b[g] + (size_t)e * n2_env;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = ae[i], w = be[i];
    u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0;
    ae[i] = u; be[i] = w;
  }
}
float run_groups(const Bases& bs) {
  const int envs = 64, nblk = 123, reps = 4; const long long n2_env = 500000;
  const long long chunk2 = ((n2_env + nblk - 1) / nblk + BLOCK - 1) / BLOCK * BLOCK;
  dim3 grid((unsigned)((n2_env + chunk2 - 1) / chunk2), envs);
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stream_groups, grid, dim3(BLOCK), 0, 0, bs, n2_env, chunk2);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_groups, grid, dim3(BLOCK), 0, 0, bs, n2_env, chunk2);
  CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
}
// out of place: read a, b; write c, d
__global__ __launch_bounds__(BLOCK) void stream_oop(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ c,
                                                    double2* __restrict__ d, long long n2_env, long long chunk2) {
  const int env = blockIdx.y, blk = blockIdx.x;
  const long long begin = (long long)blk * chunk2;
  const long long end = begin + chunk2 < n2_env ? begin + chunk2 : n2_env;
  const size_t o = (size_t)env * n2_env;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = a[o + i], w = b[o + i];
    u.x += w.x; u.y += w.y; w.x += 1.0; w.y += 1.0;
    c[o + i] = u; d[o + i] = w;
  }
}
float run_oop(double2* a, double2* b, double2* c, double2* d) {
  const int envs = 64, nblk = 123, reps = 4; const long long n2_env = 500000;
  const long long chunk2 = ((n2_env + nblk - 1) / nblk + BLOCK - 1) / BLOCK * BLOCK;
  dim3 grid((unsigned)((n2_env + chunk2 - 1) / chunk2), envs);
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(stream_oop, grid, dim3(BLOCK), 0, 0, a, b, c, d, n2_env, chunk2);
  CHK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_oop, grid, dim3(BLOCK), 0, 0, a, b, c, d, n2_env, chunk2);
  CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3f;
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
  {
    auto at = [&](double gib) { return (double2*)(base + ((size_t)(gib * 1024.0) << 20)); };
    const double L4[][4] = {{1, 1.5, 2, 2.5}, {1, 33, 2, 34}, {1, 33, 81, 97}, {1, 33, 97, 81}, {1, 33, 17, 49}, {1, 97, 33, 113}, {1, 2, 33, 34}, {1, 33, 113, 129}};
    for (auto& l : L4) {
      if (((size_t)(l[3] * 1024.0) << 20) + arr > total || ((size_t)(l[2] * 1024.0) << 20) + arr > total) continue;
      printf("out of place: read x at %g, v at %g GiB -> write at %g, %g GiB: %.0f %.0f us\n", l[0], l[1], l[2], l[3],
             run_oop(at(l[0]), at(l[1]), at(l[2]), at(l[3])), run_oop(at(l[0]), at(l[1]), at(l[2]), at(l[3])));
    }
  }
  // environments in G groups; group g's x at (1 + xs g) GiB, its v at (1 + vo + vs g) GiB
  struct Lay { const char* name; int G; double xs, vo, vs; };
  const Lay lays[] = {{"1 group, x and v in one region", 1, 0, 0.4768, 0}, {"1 group, v one region further", 1, 0, 33, 0},
                      {"2 groups: x in regions 0,1, v in 2,3", 2, 32, 64, 32}, {"2 groups: x in regions 0,1, v in 1,0 (crossed)", 2, 32, 32.5, -32},
                      {"2 groups: x in 0,0, v in 1,1", 2, 0.25, 33, 0.25}, {"3 groups (22 envs): x in 0,1,2 v in 1,2,3", 3, 32, 32.5, 32},
                      {"8 groups: x in 0, v spread over 1,1,2,2,3,3,1,2", 8, 0.06, 33, 8}};
  for (const Lay& l : lays) {
    Bases bs{}; bs.per = (64 + l.G - 1) / l.G;
    bool fits = true;
    for (int g = 0; g < l.G; ++g) {
      const double xg = 1.0 + l.xs * g, vg = 1.0 + l.vo + l.vs * g;
      const size_t xo = (size_t)(xg * 1024.0) << 20, vo = (size_t)(vg * 1024.0) << 20;
      if (xo + arr > total || vo + arr > total || vg < 0) fits = false;
      bs.a[g] = (double2*)(base + xo); bs.b[g] = (double2*)(base + vo);
    }
    if (!fits) { printf("%s: does not fit\n", l.name); continue; }
    printf("%-52s: %.0f %.0f us\n", l.name, run_groups(bs), run_groups(bs));
  }
  return 0;
}
