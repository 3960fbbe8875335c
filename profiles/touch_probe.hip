// touch_probe.hip -- what does a candidate block of pic_create's placement search cost, and can a WINDOW of it be timed instead?
// (round 4).  x = one 512,000,000 B block (config 2's positions); then candidates of the same size one after
// the other, as alloc_particles takes them.  Per candidate, in this order, wall clock (steady_clock around enqueue + sync) and HIP
// events:
//   malloc                       wall
//   W0  first touch: window of `win` MiB at offset 0 of the candidate against a window of x            (cold: never touched)
//   W0' the same window again                                                                            (in the Infinity Cache)
//   flush + W0'' the same window after x has been streamed once entirely (512 MB: the window has left the 256 MB cache)
//   W1  first touch of a second window, half a block further on
//   F1  first full pass over the rest (x, candidate), F2 / F3 full passes again = what the shipped search times
// If first touch is paid per BYTE touched, W0 is cheap and F1 dear; if per ALLOCATION, W0 carries it all.  The class of a pair
// (same region / two regions) is F3's rate: does W0'' tell the same?
// usage: touch_probe <candidates> <win MiB> [stride blocks]      Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/touch_probe profiles/touch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BLOCK = 512;
// (`scale` = 1.0 at run time: a literal would let the compiler drop the whole loop)
__global__ __launch_bounds__(BLOCK) void stream(double2* __restrict__ a, double2* __restrict__ b, long long n2, long long chunk2, double scale) {
  const long long begin = (long long)blockIdx.x * chunk2;
  const long long end = begin + chunk2 < n2 ? begin + chunk2 : n2;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = a[i], w = b[i];
    u.x *= scale; u.y *= scale; w.x *= scale; w.y *= scale;
    a[i] = u; b[i] = w;
  }
}
__global__ __launch_bounds__(BLOCK) void stream1(double2* __restrict__ a, long long n2, long long chunk2, double scale) {
  const long long begin = (long long)blockIdx.x * chunk2;
  const long long end = begin + chunk2 < n2 ? begin + chunk2 : n2;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = a[i];
    u.x *= scale; u.y *= scale;
    a[i] = u;
  }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct T { float ev_us; double wall_us; };
static hipEvent_t e0, e1;
static T pass(double2* a, double2* b, size_t bytes) {
  const long long n2 = (long long)(bytes / sizeof(double2));
  long long nb = n2 / ((long long)BLOCK * 8); if (nb < 256) nb = 256;
  const long long chunk2 = (n2 + nb - 1) / nb;
  const double t0 = now();
  CHK(hipEventRecord(e0, 0));
  if (b) hipLaunchKernelGGL(stream, dim3((unsigned)nb), dim3(BLOCK), 0, 0, a, b, n2, chunk2, 1.0);
  else hipLaunchKernelGGL(stream1, dim3((unsigned)nb), dim3(BLOCK), 0, 0, a, n2, chunk2, 1.0);
  CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
  T t; t.wall_us = (now() - t0) * 1e6; CHK(hipEventElapsedTime(&t.ev_us, e0, e1)); t.ev_us *= 1e3f;
  return t;
}
int main(int argc, char** argv) {
  const int cands = argc > 1 ? atoi(argv[1]) : 12;
  const size_t win = (size_t)(argc > 2 ? atoi(argv[2]) : 64) << 20;
  const int stride = argc > 3 ? atoi(argv[3]) : 1;
  const size_t pbytes = 512000000;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double t = now();
  char* x = nullptr; CHK(hipMalloc(&x, pbytes));
  printf("malloc x %.0f us\n", (now() - t) * 1e6);
  T tx = pass((double2*)x, nullptr, pbytes);
  printf("x first full pass (x alone): events %.0f us, wall %.0f us\n", tx.ev_us, tx.wall_us);
  tx = pass((double2*)x, nullptr, pbytes);
  printf("x second full pass: events %.0f us, wall %.0f us\n", tx.ev_us, tx.wall_us);
  const double gb_full = 4.0 * pbytes / 1e3, gb_win = 4.0 * win / 1e3;      // GB/s = this / us
  std::vector<char*> all;
  const double t_all = now();
  for (int c = 0; c < cands; ++c) {
    char* b = nullptr;
    double tm = 0;
    for (int s = 0; s < stride; ++s) {                  // stride - 1 untouched blocks between timed ones
      t = now();
      if (hipMalloc(&b, pbytes) != hipSuccess) { printf("out of memory at candidate %d\n", c); (void)hipGetLastError(); b = nullptr; break; }
      tm += (now() - t) * 1e6;
      all.push_back(b);
    }
    if (!b) break;
    char* xw = x + (size_t)(c % (int)(pbytes / win)) * win;                   // a different window of x every time
    const T w0 = pass((double2*)xw, (double2*)b, win);
    const T w0b = pass((double2*)xw, (double2*)b, win);
    pass((double2*)x, nullptr, pbytes);
    const T w0c = pass((double2*)xw, (double2*)b, win);
    const T w1 = pass((double2*)xw, (double2*)(b + pbytes / 2 / 4096 * 4096), win);
    const T f1 = pass((double2*)x, (double2*)b, pbytes);
    const T f2 = pass((double2*)x, (double2*)b, pbytes);
    const T f3 = pass((double2*)x, (double2*)b, pbytes);
    printf("cand %2d @%p malloc %6.0f us | W0 first %7.0f (wall %7.0f) warm %5.1f cold-again %5.1f = %4.0f GB/s | W1 first %7.0f (wall %7.0f) | "
           "F1 %7.0f (wall %7.0f) F2 %6.1f F3 %6.1f = %4.0f GB/s | t=%.1f ms\n",
           c, (void*)b, tm, w0.ev_us, w0.wall_us, w0b.ev_us, w0c.ev_us, gb_win / w0c.ev_us, w1.ev_us, w1.wall_us, f1.ev_us, f1.wall_us,
           f2.ev_us, f3.ev_us, gb_full / f3.ev_us, (now() - t_all) * 1e3);
    fflush(stdout);
  }
  // second look at every candidate through windows only (all memory touched by now): cold window rate vs the full-pass class
  printf("windows again, everything touched (flush, then one cold pass of the window):\n");
  for (size_t k = stride - 1; k < all.size(); k += stride) {
    char* xw = x + (size_t)((k / stride) % (pbytes / win)) * win;
    pass((double2*)x, nullptr, pbytes);
    const T a = pass((double2*)xw, (double2*)all[k], win);
    pass((double2*)x, nullptr, pbytes);
    const T b2 = pass((double2*)xw, (double2*)(all[k] + pbytes / 2 / 4096 * 4096), win);
    printf("  cand %2zu: window@0 %5.1f us = %4.0f GB/s, window@half %5.1f us = %4.0f GB/s\n", k / stride, a.ev_us, gb_win / a.ev_us, b2.ev_us, gb_win / b2.ev_us);
  }
  t = now();
  for (char* b : all) CHK(hipFree(b));
  printf("free of %zu blocks: %.1f ms\n", all.size(), (now() - t) * 1e3);
  t = now();
  char* again = nullptr; CHK(hipMalloc(&again, pbytes));
  const double tm = (now() - t) * 1e6;
  const T fa = pass((double2*)x, (double2*)again, pbytes);
  printf("malloc right after the frees: %.0f us, first full pass %.0f us (wall %.0f)\n", tm, fa.ev_us, fa.wall_us);
  return 0;
}
