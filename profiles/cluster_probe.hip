// cluster_probe.hip -- what would an all-to-all exchange of partial meshes between the G workgroups of one environment cost?
// (experiments_r3.md 12: pricing a multi-workgroup register-resident schedule before building it.)
// Every workgroup of a group writes `words` 8-byte granules {61-bit value | 3-bit tag} with write-through (sc1) stores into
// its own slab (two buffers, alternating) and then polls the other G - 1 slabs with sc1 loads until every tag is the
// current phase's, sums, and goes on.  One iteration = one exchange + one workgroup barrier.  Bounded polls: a lost member
// ends the kernel with an error count instead of a hang.
//   hipcc -O3 --offload-arch=gfx950 -o profiles/bin/cluster_probe profiles/cluster_probe.hip && profiles/bin/cluster_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
constexpr int NT = 512;
constexpr u64 kValueMask = (1ull << 61) - 1;

__device__ __forceinline__ u64 encode(u64 v, unsigned tag) { return (v & kValueMask) | ((u64)tag << 61); }

template <int PER_THREAD, int G>
__global__ __launch_bounds__(NT) void exchange_kernel(u64* __restrict__ slabs, int words, int iters, u64* __restrict__ sums,
                                                      u64* __restrict__ ticks, unsigned* __restrict__ errors, int work) {
  const int tid = threadIdx.x;
  const int group = blockIdx.x / G, m = blockIdx.x % G;
  u64* base = slabs + (size_t)group * 2 * G * words;
  __shared__ u64 mesh[2048];
  u64 acc = 0;
  unsigned err = 0;
  u64 t0 = 0;
  for (int it = 0; it < iters; ++it) {
    if (it == iters / 4) t0 = __builtin_amdgcn_s_memrealtime();
    const int buf = it & 1;
    const unsigned tag = (unsigned)((it >> 1) % 3) + 1u;
    u64* mine = base + ((size_t)buf * G + m) * words;
    // some private work between exchanges (the particle phase's stand-in): `work` dependent LDS round trips
    for (int k = 0; k < work; ++k) {
      mesh[tid] = acc + k;
      __syncthreads();
      acc += mesh[(tid + 1) & (NT - 1)];
      __syncthreads();
    }
    u64 own[PER_THREAD];
#pragma unroll
    for (int c = 0; c < PER_THREAD; ++c) {
      const int i = tid + c * NT;
      own[c] = (u64)(it * 131 + i * 7 + m);
      if (i < words) __hip_atomic_store(mine + i, encode(own[c], tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int c = 0; c < PER_THREAD; ++c) {
      const int i = tid + c * NT;
      if (i < words) {
        u64 total = own[c];
        // all G - 1 slabs are requested at once and polled together (one after the other the exchange costs 0.35-0.45 us per member)
        constexpr int MAXG = G;
        u64 w[MAXG > 1 ? MAXG : 2];
        int tries = 0;
        for (;;) {
#pragma unroll
          for (int o = 1; o < MAXG; ++o)
            if (o < G) w[o] = __hip_atomic_load(base + ((size_t)buf * G + (m + o) % G) * words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool ok = true;
#pragma unroll
          for (int o = 1; o < MAXG; ++o)
            if (o < G) ok = ok && (unsigned)(w[o] >> 61) == tag;
          if (ok) break;
          if (++tries > (1 << 20)) { ++err; break; }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int o = 1; o < MAXG; ++o)
          if (o < G) {
            total += w[o] & kValueMask;
            if ((w[o] & kValueMask) != (u64)(it * 131 + i * 7 + (m + o) % G) && (unsigned)(w[o] >> 61) == tag) ++err;
          }
        mesh[i] = total;
      }
    }
    __syncthreads();
    acc += mesh[(tid * 5) % words];
    __syncthreads();
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  sums[(size_t)blockIdx.x * NT + tid] = acc;
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  if (err) atomicAdd(errors, err);
}

int main() {
  const int iters = 4000;
  u64 *slabs, *sums, *ticks;
  unsigned* errors;
  const int maxG = 16, maxGroups = 16, maxWords = 1024;
  hipMalloc(&slabs, (size_t)maxGroups * 2 * maxG * maxWords * sizeof(u64));
  hipMalloc(&sums, (size_t)maxGroups * maxG * NT * sizeof(u64));
  hipMalloc(&ticks, (size_t)maxGroups * maxG * sizeof(u64));
  hipMalloc(&errors, sizeof(unsigned));
  printf("| groups | G | words | work | us per exchange (+1 barrier) | errors |\n|---|---|---|---|---|---|\n");
  for (int groups : {1, 8})
    for (int G : {1, 2, 4, 8, 16})
      for (int words : {256, 512, 1024})
        for (int work : {0, 4}) {
          if (groups * G > 256) continue;
          hipMemset(slabs, 0, (size_t)maxGroups * 2 * maxG * maxWords * sizeof(u64));
          hipMemset(errors, 0, sizeof(unsigned));
          hipDeviceSynchronize();
#define LAUNCH(PT, GG) hipLaunchKernelGGL((exchange_kernel<PT, GG>), dim3(groups * G), dim3(NT), 0, 0, slabs, words, iters, sums, ticks, errors, work)
#define LAUNCH_G(PT) do { switch (G) { case 1: LAUNCH(PT, 1); break; case 2: LAUNCH(PT, 2); break; case 4: LAUNCH(PT, 4); break; case 8: LAUNCH(PT, 8); break; default: LAUNCH(PT, 16); } } while (0)
          if (words <= 512) LAUNCH_G(1);
          else LAUNCH_G(2);
          if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
          std::vector<u64> t(groups * G);
          unsigned e = 0;
          hipMemcpy(t.data(), ticks, t.size() * sizeof(u64), hipMemcpyDeviceToHost);
          hipMemcpy(&e, errors, sizeof(unsigned), hipMemcpyDeviceToHost);
          u64 worst = 0;
          for (u64 v : t) worst = v > worst ? v : worst;
          printf("| %d | %d | %d | %d | %.3f | %u |\n", groups, G, words, work, worst * 0.01 / (iters - iters / 4), e);
        }
  return 0;
}
