// mailbox_probe.hip -- what would a trainer's one-action-per-step loop gain if the resident kernel stayed on the device for K steps
// and took each step's action from a pinned-host mailbox, instead of one launch + one wait per step?  (round 4)
// Priced BEFORE building anything into the library.  Both forms do the same thing per iteration: the host hands over an action
// (10 doubles), every workgroup does `work_us` microseconds of dependent arithmetic (standing in for the 15-18 us step of one
// environment of N = 5000), writes its observation (2 x 40 KB per environment) to pinned host memory, and the host sees it:
//
//   launch   hipLaunchKernelGGL with the action in the argument block + hipStreamSynchronize          (pic_step_observe today)
//   mailbox  ONE launch for all iterations; per iteration thread 0 of each workgroup spins on a host sequence word (system-scope
//            acquire loads over PCIe), bounded by a wall-clock timeout (the kernel exits with a code, never spins unbounded and
//            needs no co-residency: no workgroup waits for another); after the stores a system-scope release store of the
//            workgroup's own done word, which the host spins on
//
// usage: mailbox_probe [envs=1] [work_us=16] [iters=3000]     Build: hipcc -O3 --offload-arch=gfx950 -o profiles/bin/mailbox_probe profiles/mailbox_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BLOCK = 512, N = 5000, NACT = 10;
struct Action { double a[NACT]; };

__device__ __forceinline__ double work(double seed, int rounds) {        // dependent float64 arithmetic, ~4 cycles per round
  double v = seed;
  for (int k = 0; k < rounds; ++k) v = v * 1.0000001 + 1e-9;
  return v;
}
__device__ __forceinline__ void observe(double* __restrict__ host_obs, const double* __restrict__ state, double bias) {
  double* out = host_obs + (size_t)blockIdx.x * 2 * N;
  for (int i = threadIdx.x; i < 2 * N; i += BLOCK) out[i] = state[(size_t)blockIdx.x * 2 * N + i] + bias;
}

__global__ __launch_bounds__(BLOCK) void step_once(const double* __restrict__ state, double* __restrict__ host_obs, Action act, int rounds) {
  double s = 0;
  for (int k = 0; k < NACT; ++k) s += act.a[k];
  const double w = work(s + threadIdx.x, rounds);
  observe(host_obs, state, w * 1e-300);
}

// cmd: host word, the number of the action that is ready; done: [envs] host words; status: [envs] device words (0 ok, 1 timed out)
__global__ __launch_bounds__(BLOCK) void step_mailbox(const double* __restrict__ state, double* __restrict__ host_obs,
                                                      const unsigned long long* cmd, const Action* mailbox,
                                                      unsigned long long* done, int* status, int iters, int rounds,
                                                      long long timeout_ticks) {
  __shared__ int quit;
  __shared__ double sum;
  for (int it = 1; it <= iters; ++it) {
    if (threadIdx.x == 0) {
      quit = 0;
      const long long t0 = wall_clock64();                                // 100 MHz
      while (__hip_atomic_load(cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)it) {
        if (wall_clock64() - t0 > timeout_ticks) { quit = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      double s = 0;
      if (!quit) for (int k = 0; k < NACT; ++k) s += __hip_atomic_load(&mailbox->a[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      sum = s;
    }
    __syncthreads();
    if (quit) {                                                           // every wave reaches this: the grid drains
      if (threadIdx.x == 0) status[blockIdx.x] = 1;
      return;
    }
    const double w = work(sum + threadIdx.x, rounds);
    observe(host_obs, state, w * 1e-300);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done + blockIdx.x, (unsigned long long)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static void report(const char* name, std::vector<double>& t) {
  std::sort(t.begin(), t.end());
  const size_t n = t.size();
  printf("| %-64s | %7.1f | %7.1f | %7.1f | %8.1f |\n", name, t[n / 2], t[n / 10], t[n * 9 / 10], t[n - 1]);
}

int main(int argc, char** argv) {
  const int envs = argc > 1 ? atoi(argv[1]) : 1;
  const double work_us = argc > 2 ? atof(argv[2]) : 16.0;
  const int iters = argc > 3 ? atoi(argv[3]) : 3000;
  hipStream_t s;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *state, *host_obs;
  CHK(hipMalloc(&state, (size_t)envs * 2 * N * sizeof(double)));
  CHK(hipMemset(state, 0, (size_t)envs * 2 * N * sizeof(double)));
  CHK(hipHostMalloc(&host_obs, (size_t)envs * 2 * N * sizeof(double), hipHostMallocDefault));
  unsigned long long *cmd, *done;
  Action* mailbox;
  int* status;
  CHK(hipHostMalloc(&cmd, 64, hipHostMallocDefault));
  CHK(hipHostMalloc(&done, (size_t)envs * sizeof(unsigned long long), hipHostMallocDefault));
  CHK(hipHostMalloc(&mailbox, sizeof(Action), hipHostMallocDefault));
  CHK(hipMalloc(&status, envs * sizeof(int)));
  CHK(hipMemset(status, 0, envs * sizeof(int)));
  // calibrate `rounds` so that the arithmetic alone takes work_us
  int rounds = 1000;
  {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    Action a{};
    for (int k = 0; k < 3; ++k) {
      hipLaunchKernelGGL(step_once, dim3(envs), dim3(BLOCK), 0, s, state, host_obs, a, 0);
      CHK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(step_once, dim3(envs), dim3(BLOCK), 0, s, state, host_obs, a, 0);
      CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
      float base = 0; CHK(hipEventElapsedTime(&base, e0, e1));
      CHK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(step_once, dim3(envs), dim3(BLOCK), 0, s, state, host_obs, a, 20000);
      CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
      float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
      rounds = (int)(20000.0 * work_us / std::max(1.0, (ms - base) * 1e3));
      if (k == 2) printf("calibration: kernel without arithmetic %.1f us (stores to pinned memory included), 20000 rounds %.1f us -> %d rounds for %.0f us\n",
                         base * 1e3, ms * 1e3, rounds, work_us);
    }
  }
  printf("%d environment(s) of N = %d, %.0f us of arithmetic per step, %d iterations; us per iteration\n", envs, N, work_us, iters);
  printf("| %-64s | median  | p10     | p90     | max      |\n|---|---|---|---|---|\n", "form");
  for (int round = 0; round < 2; ++round) {
    {   // one launch + one wait per iteration
      std::vector<double> t(iters);
      Action a{};
      for (int it = 0; it < iters; ++it) {
        const auto t0 = std::chrono::steady_clock::now();
        a.a[0] = it;
        hipLaunchKernelGGL(step_once, dim3(envs), dim3(BLOCK), 0, s, state, host_obs, a, rounds);
        CHK(hipStreamSynchronize(s));
        t[it] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      report("launch + hipStreamSynchronize per step", t);
    }
    {   // one launch, mailbox per iteration
      std::vector<double> t(iters);
      *cmd = 0;
      for (int e = 0; e < envs; ++e) done[e] = 0;
      std::atomic_thread_fence(std::memory_order_seq_cst);
      hipLaunchKernelGGL(step_mailbox, dim3(envs), dim3(BLOCK), 0, s, state, host_obs, cmd, mailbox, done, status, iters, rounds,
                         (long long)200000);   // 2 ms at 100 MHz
      CHK(hipGetLastError());
      bool lost = false;
      for (int it = 1; it <= iters && !lost; ++it) {
        const auto t0 = std::chrono::steady_clock::now();
        mailbox->a[0] = it;
        __atomic_store_n(cmd, (unsigned long long)it, __ATOMIC_RELEASE);
        for (int e = 0; e < envs; ++e) {
          while (__atomic_load_n(done + e, __ATOMIC_ACQUIRE) < (unsigned long long)it) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5) { lost = true; break; }
          }
          if (lost) break;
        }
        t[it - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      CHK(hipStreamSynchronize(s));
      std::vector<int> st(envs);
      CHK(hipMemcpy(st.data(), status, envs * sizeof(int), hipMemcpyDeviceToHost));
      int timed_out = 0; for (int v : st) timed_out += v;
      if (lost || timed_out) printf("mailbox run ended early: host gave up %d, workgroups timed out %d\n", (int)lost, timed_out);
      else report("one launch, action by mailbox, observation + done word in pinned", t);
      CHK(hipMemset(status, 0, envs * sizeof(int)));
    }
  }
  // the bounded wait itself: a kernel nobody feeds must come back by its own timeout
  {
    *cmd = 0;
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(step_mailbox, dim3(envs), dim3(BLOCK), 0, s, state, host_obs, cmd, mailbox, done, status, 5, rounds, (long long)200000);
    CHK(hipStreamSynchronize(s));
    std::vector<int> st(envs);
    CHK(hipMemcpy(st.data(), status, envs * sizeof(int), hipMemcpyDeviceToHost));
    int timed_out = 0; for (int v : st) timed_out += v;
    printf("unfed kernel: returned after %.2f ms with %d of %d workgroups reporting a timeout\n",
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), timed_out, envs);
  }
  return 0;
}
