// d2h_probe.hip -- how should a small state (2 x 40 KB: the reference's N = 5000) come back to the host after every step?
// A Gym-style loop reads it each iteration; pic_get_particles took 23 us on one run and 108 us on the next on the same box.
// Variants, each behind a ~20 us kernel on the same stream and followed by ONE hipStreamSynchronize:
//   2d      hipMemcpy2DAsync of the two rows into pinned memory (what pic_get_particles did)
//   1d x2   two hipMemcpyAsync
//   kernel  a copy kernel that writes the pinned (host-coherent) buffer itself
//   hipcc -O3 --offload-arch=gfx950 -o profiles/bin/d2h_probe profiles/d2h_probe.hip && profiles/bin/d2h_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void busy_kernel(double* p, int n, int rounds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = p[i];
  for (int k = 0; k < rounds; ++k) v = v * 1.0000001 + 1e-9;
  p[i] = v;
}

__global__ void pack_kernel(const double* __restrict__ x, const double* __restrict__ v, double* __restrict__ host, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    host[i] = x[i];
    host[N + i] = v[i];
  }
}

int main() {
  const int N = 5000, ld = 5056, iters = 3000;
  double *dev, *pinned;
  hipStream_t s;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
  if (hipMalloc(&dev, 2 * ld * sizeof(double)) != hipSuccess) return 1;
  if (hipHostMalloc(&pinned, 2 * N * sizeof(double), hipHostMallocDefault) != hipSuccess) return 1;
  (void)hipMemset(dev, 0, 2 * ld * sizeof(double));
  (void)hipDeviceSynchronize();
  const char* names[] = {"2d", "1d x2", "kernel", "none (kernel + sync only)", "kernel, waiting by hipStreamQuery polls", "none, waiting by hipStreamQuery polls"};
  printf("| variant | median us | p10 | p90 | max |\n|---|---|---|---|---|\n");
  for (int round = 0; round < 2; ++round)
    for (int variant = 0; variant < 6; ++variant) {
      std::vector<double> t(iters);
      for (int it = 0; it < iters; ++it) {
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(busy_kernel, dim3(20), dim3(256), 0, s, dev, N, 900);
        if (variant == 0)
          (void)hipMemcpy2DAsync(pinned, N * sizeof(double), dev, ld * sizeof(double), N * sizeof(double), 2, hipMemcpyDeviceToHost, s);
        else if (variant == 1) {
          (void)hipMemcpyAsync(pinned, dev, N * sizeof(double), hipMemcpyDeviceToHost, s);
          (void)hipMemcpyAsync(pinned + N, dev + ld, N * sizeof(double), hipMemcpyDeviceToHost, s);
        } else if (variant == 2 || variant == 4)
          hipLaunchKernelGGL(pack_kernel, dim3((N + 255) / 256), dim3(256), 0, s, dev, dev + ld, pinned, N);
        if (variant >= 4) {
          while (hipStreamQuery(s) == hipErrorNotReady) {}
        } else {
          (void)hipStreamSynchronize(s);
        }
        t[it] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      std::sort(t.begin(), t.end());
      printf("| %s (round %d) | %.1f | %.1f | %.1f | %.1f |\n", names[variant], round, t[iters / 2], t[iters / 10], t[iters * 9 / 10], t[iters - 1]);
    }
  return 0;
}
