// pic_aux.h -- kernels off the step path: particle-field gather, CIC bookkeeping, actuator, Fourier modes,
// Philox sampler, phase-space histogram, streaming probe.
#pragma once
#include "pic_device.h"

namespace {

// PIC.E (pic.py:120) and the CIC bookkeeping attributes (pic.py:104-107), on demand.
template <typename P, int SHAPE>
__global__ __launch_bounds__(BLOCK) void gather_E_kernel(const typename P::X* __restrict__ x,
                                                         const double* __restrict__ E_mesh,
                                                         typename P::W* __restrict__ E_out, long long N, long long ld,
                                                         int Ng, double Ld, double dxd) {
  using T = typename P::W;
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* Es = reinterpret_cast<T*>(smem_raw);
  const int env = blockIdx.y;
  for (int i = threadIdx.x; i < Ng + 2; i += BLOCK) {
    int node = i - OFF;
    node = node < 0 ? node + Ng : (node >= Ng ? node - Ng : node);
    Es[i] = (T)E_mesh[(size_t)env * Ng + node];
  }
  __syncthreads();
  const Consts<P> k(Ld, dxd, Ng);
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    T w[3];
    typename P::X xw;
    int j;
    unsigned frac, bad = 0;
    locate<P, SHAPE>(x[(size_t)env * ld + i], k, xw, j, w, frac, bad);
    E_out[(size_t)env * N + i] = gather_field<T, SHAPE>(Es, j, w);
  }
}

// Shape-function bookkeeping of every particle: mesh indices and weights as CIC / TSC return them
// (interpolate.py:8-14: l = floor(x/dx), r = (l+1) mod Ng;  interpolate.py:26-36: m = floor(x/dx), l = (m-1) mod Ng,
// r = (m+1) mod Ng).  x: [env][ld]; idx, w: [env][3][N], rows l, r, (unused) for CIC and l, m, r for TSC.
template <typename P, int SHAPE>
__global__ __launch_bounds__(BLOCK) void shape_query_kernel(const typename P::X* __restrict__ x, long long N,
                                                            long long ld, int Ng, double Ld, double dxd,
                                                            long long* __restrict__ idx, double* __restrict__ w_out) {
  using T = typename P::W;
  const Consts<P> k(Ld, dxd, Ng);
  const int env = blockIdx.y;
  long long* je = idx + (size_t)env * 3 * N;
  double* we = w_out + (size_t)env * 3 * N;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    T w[3];
    typename P::X xw;
    int j;
    unsigned frac, bad = 0;
    locate<P, SHAPE>(x[(size_t)env * ld + i], k, xw, j, w, frac, bad);
    if (SHAPE == PIC_CIC) {
      je[i] = j;
      je[N + i] = (j + 1 == Ng) ? 0 : j + 1;
      je[2 * N + i] = 0;
    } else {                                   // locate's j is the node m itself (LDS slot of node m-1)
      je[i] = (j == 0) ? Ng - 1 : j - 1;
      je[N + i] = j;
      je[2 * N + i] = (j + 1 == Ng) ? 0 : j + 1;
    }
    we[i] = (double)w[0];
    we[N + i] = (double)w[1];
    we[2 * N + i] = (double)w[2];
  }
}

// Fixed-point positions at the boundary: callers hand over and receive positions as floats of the particle dtype.
// in: dense [env][N] floats -> padded [env][ld] position units; out: the reverse (a value that rounds up to L
// as a float is returned as 0, its periodic image).
template <typename P, typename F>
__global__ __launch_bounds__(BLOCK) void positions_in_kernel(const F* __restrict__ src, typename P::X* __restrict__ dst,
                                                             long long N, long long ld, double L,
                                                             unsigned long long* __restrict__ bad_count) {
  const int env = blockIdx.y;
  unsigned bad = 0;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK)
    dst[(size_t)env * ld + i] = pos_from_length<P>((double)src[(size_t)env * N + i], L, bad);
  if (bad) atomicAdd(bad_count, (unsigned long long)bad);
}

template <typename P, typename F>
__global__ __launch_bounds__(BLOCK) void positions_out_kernel(const typename P::X* __restrict__ src, F* __restrict__ dst,
                                                              long long N, long long ld, double L) {
  const int env = blockIdx.y;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    F f = (F)pos_to_length<P>(src[(size_t)env * ld + i], L);
    if (f >= (F)L) f = F(0);
    dst[(size_t)env * N + i] = f;
  }
}

// What a Gym-style loop reads after a step, written by the device itself into PINNED host memory: x | v rows of every environment
// ([2][num_envs][N] words of W bytes) and the 3 num_envs energies that follow KE in memory.  Behind a 20 us step one more small
// kernel costs 2.6 us up to the synchronisation, a strided copy command 7 and a second copy for the energies 6 more
// (profiles/d2h_probe.hip, profiles/gym_breakdown.py).  grid (ceil(N / BLOCK), 2 num_envs).
template <typename W>
__global__ __launch_bounds__(BLOCK) void observe_kernel(const W* __restrict__ x, const W* __restrict__ v, long long N, long long ld,
                                                        int num_envs, W* __restrict__ host_part, const double* __restrict__ energies,
                                                        double* __restrict__ host_scal) {
  const int r = blockIdx.y;
  const W* src = r < num_envs ? x + (size_t)r * ld : v + (size_t)(r - num_envs) * ld;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK)
    host_part[(size_t)r * N + i] = src[i];
  if (host_scal && blockIdx.x == 0 && r == 0)
    for (int i = threadIdx.x; i < 3 * num_envs; i += BLOCK) host_scal[i] = energies[i];
}

// n | E_mesh | phi of every environment into pinned host memory (a host-side feedback loop reads E_mesh after every step:
// run_feedback.py:133; three copy commands into pageable memory cost 25 us behind a 22 us step).  grid (ceil(count / BLOCK), 3).
__global__ __launch_bounds__(BLOCK) void fields_out_kernel(const double* __restrict__ n, const double* __restrict__ E,
                                                           const double* __restrict__ phi, double* __restrict__ host, long long count) {
  const double* src = blockIdx.y == 0 ? n : (blockIdx.y == 1 ? E : phi);
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < count; i += (long long)gridDim.x * BLOCK)
    host[(size_t)blockIdx.y * count + i] = src[i];
}

// particles of the step just finished -> slot `s` of a snapshot array [steps][2][num_envs][N] (floats of the particle dtype)
template <typename P>
__global__ __launch_bounds__(BLOCK) void record_particles_kernel(const typename P::X* __restrict__ x,
                                                                 const typename P::V* __restrict__ v,
                                                                 typename P::V* __restrict__ snap, int s, long long N,
                                                                 long long ld, double L) {
  using F = typename P::V;
  const int env = blockIdx.y, E = gridDim.y;
  F* sx = snap + ((size_t)s * 2 * E + env) * (size_t)N;
  F* sv = sx + (size_t)E * (size_t)N;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    F xf = (F)pos_to_length<P>(x[(size_t)env * ld + i], L);
    if (P::kFixed && xf >= (F)L) xf = F(0);
    sx[i] = xf;
    sv[i] = v[(size_t)env * ld + i];
  }
}

// Twiddle table of the Fourier modes: tw[m-1][j] = cos(2 pi m j / Ng), tw[rows + m-1][j] = sin(2 pi m j / Ng), m = 1..rows.
// The angle is reduced exactly in integers before the trig call.
__global__ __launch_bounds__(BLOCK) void twiddle_kernel(double* __restrict__ tw, int Ng, int rows) {
  const int m = blockIdx.y + 1;
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= Ng) return;
  const long long r = ((long long)m * j) % Ng;
  double sn, cs;
  sincospi(2.0 * (double)r / (double)Ng, &sn, &cs);
  tw[(size_t)(m - 1) * Ng + j] = cs;
  tw[((size_t)rows + m - 1) * Ng + j] = sn;
}

// compute_E_k_spectrum rows 1..M (src/interpret/spectrum.py:16): Ek[m] = fft(E_mesh)[m] / Ng * 2.
__global__ __launch_bounds__(BLOCK) void modes_kernel(const double* __restrict__ E_mesh, const double* __restrict__ tw, int rows,
                                                      double* __restrict__ re, double* __restrict__ im, int Ng, int M) {
  __shared__ double ws[2 * WAVES];
  const int env = blockIdx.y, m = blockIdx.x;
  double a, b;
  mesh_mode<WAVES>(E_mesh + (size_t)env * Ng, tw + (size_t)m * Ng, tw + ((size_t)rows + m) * Ng, Ng, ws, a, b);
  if (threadIdx.x == 0) {
    re[(size_t)env * M + m] = a;
    im[(size_t)env * M + m] = b;
  }
}

// The feedback law's action from the mesh field as it stands (first step of a pic_step_feedback call on the streaming
// schedule; later steps: the post-step solve computes it, pic_device.h: solve_block).
__global__ __launch_bounds__(BLOCK) void feedback_kernel(const double* __restrict__ E_mesh, Feedback fb, int Ng) {
  __shared__ double ws[2 * WAVES];
  feedback_action<WAVES>(E_mesh + (size_t)blockIdx.x * Ng, fb, blockIdx.x, Ng, ws, nullptr);
}

// ---------------------------------------------------------------------------------------------
// Device-side initial conditions (the distributions of src/env/dist.py:27-194, not its RNG stream).
// Philox4x32-10 counter-based generator: particle i of environment e draws from counter (i, attempt)
// under key (seed, e), so a sample is reproducible and independent of the launch geometry.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ double u01(uint32_t a, uint32_t b) {       // 53 random bits -> (0, 1)
  const unsigned long long bits = ((unsigned long long)a << 21) ^ ((unsigned long long)b >> 11);
  return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
}

// kind 0: two-stream, halves at +v0 / -v0 (dist.py:70-102); kind 1: bump-on-tail, int(N/(1+a)) bulk
// particles from N(0,1) then the beam from N(v0, sigma) (dist.py:151-189, same ordering as high_indx).
// Velocities are truncated to [-10, 10] like the reference's uniform proposal; then v *= 1 + A sin(2 pi
// n_mode x / L) (src/env/pic.py:68).
template <typename P>
__global__ __launch_bounds__(BLOCK) void sample_kernel(typename P::X* __restrict__ x, typename P::V* __restrict__ v,
                                                       long long N, long long ld,
                                                       int kind, double a, double v0, double sigma, double A,
                                                       int n_mode, double L, unsigned long long seed, int env_base) {
  const int env = blockIdx.y;
  const long long n_first = kind == 0 ? N / 2 : (long long)((double)N * (1.0 / (1.0 + a)));
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    double mu, sg;
    if (kind == 0) { mu = i < n_first ? v0 : -v0; sg = sigma; }
    else { mu = i < n_first ? 0.0 : v0; sg = i < n_first ? 1.0 : sigma; }
    const uint32_t k0 = (uint32_t)seed ^ (0x85EBCA6Bu * (uint32_t)(env_base + env + 1)), k1 = (uint32_t)(seed >> 32);
    uint32_t c[4] = {(uint32_t)i, (uint32_t)((unsigned long long)i >> 32), 0u, 0x50494331u};
    philox4x32_10(c, k0, k1);
    double xs = u01(c[0], c[1]) * L;
    if (xs >= L) xs = 0.0;
    double ua = u01(c[2], c[3]), vs = 0.0;
    for (uint32_t attempt = 1; attempt < 64; ++attempt) {
      uint32_t d[4] = {(uint32_t)i, (uint32_t)((unsigned long long)i >> 32), attempt, 0x50494332u};
      philox4x32_10(d, k0, k1);
      double sn, cs;
      sincospi(2.0 * u01(d[0], d[1]), &sn, &cs);
      vs = mu + sg * sqrt(-2.0 * log(ua)) * cs;
      if (vs >= -10.0 && vs <= 10.0) break;
      ua = u01(d[2], d[3]);                        // rejected (outside the proposal's support): redraw
    }
    vs *= 1.0 + A * sin(2.0 * 3.14159265358979323846 * n_mode * xs / L);
    unsigned bad = 0;
    if constexpr (!P::kFixed) {       // the float32 image of a value just below L may be L itself: its periodic image is 0
      if ((double)(typename P::X)xs >= L) xs = 0.0;
    }
    x[(size_t)env * ld + i] = pos_from_length<P>(xs, L, bad);
    v[(size_t)env * ld + i] = (typename P::V)vs;
  }
}

// np.histogram2d bin of `val` for edges = np.linspace(lo, hi, nb + 1) (edges[i] = lo + i*step, last = hi):
// searchsorted(edges, val, 'right') - 1, the last edge inclusive, -1 for values outside [lo, hi].
__device__ __forceinline__ int hist_bin(double val, double lo, double hi, double step, int nb) {
  if (!(val >= lo && val <= hi)) return -1;
  int b = (int)((val - lo) / step);
  b = b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);
  auto edge = [&](int i) { return i == nb ? hi : lo + (double)i * step; };
  while (b > 0 && val < edge(b)) --b;
  while (b < nb - 1 && val >= edge(b + 1)) ++b;
  return b;
}

// Phase-space histogram of the KL diagnostic (src/control/objective.py:8-14): counts[env][ix][iv].
template <typename P>
__global__ __launch_bounds__(BLOCK) void phase_hist_kernel(const typename P::X* __restrict__ x,
                                                           const typename P::V* __restrict__ v,
                                                           unsigned* __restrict__ counts, long long N, long long ld,
                                                           int nb, double L, double vmin, double vmax) {
  const int env = blockIdx.y;
  const double sx = (L - 0.0) / nb, sv = (vmax - vmin) / nb;      // np.linspace step = (stop - start) / div
  unsigned* c = counts + (size_t)env * nb * nb;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    const int ix = hist_bin(pos_to_length<P>(x[(size_t)env * ld + i], L), 0.0, L, sx, nb);
    const int iv = hist_bin((double)v[(size_t)env * ld + i], vmin, vmax, sv, nb);
    if (ix >= 0 && iv >= 0) atomicAdd(&c[(size_t)ix * nb + iv], 1u);
  }
}

// KL cost of the phase-space density against a target (src/control/objective.py:16-18, Reward.compute_kl_divergence,
// src/control/rl/reward.py:43-46): sum_ij rel_entr(f_ij, feq_ij + 1e-12) dx dv with f = counts n0 / dx / dv / N
// (objective.py:12) and rel_entr(a, b) = a log(a / b) for a > 0, 0 for a = 0.  One workgroup per environment.
__global__ __launch_bounds__(BLOCK) void phase_kl_kernel(const unsigned* __restrict__ counts, const double* __restrict__ feq,
                                                         double* __restrict__ kl, int nb2, double norm, double dxdv) {
  __shared__ double ws[WAVES];
  const unsigned* c = counts + (size_t)blockIdx.x * nb2;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb2; i += BLOCK) {
    const double f = (double)c[i] * norm;
    if (f > 0.0) acc += f * log(f / (feq[i] + 1e-12));
  }
  const double t = block_sum<WAVES>(acc, ws);
  if (threadIdx.x == 0) kl[blockIdx.x] = t * dxdv;
}

// Streaming ceiling of this box for the sweeps' access shape: read two arrays, write two arrays, 16 B
// per lane, same grid -- what a sweep would take if it did no arithmetic at all.
__global__ __launch_bounds__(BLOCK) void stream_probe_kernel(double2* __restrict__ a, double2* __restrict__ b,
                                                             long long n2, long long chunk2, double scale,
                                                             int reverse) {
  const long long bid = reverse ? (long long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
  long long begin = bid * chunk2;
  long long end = begin + chunk2 < n2 ? begin + chunk2 : n2;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = a[i], w = b[i];
    u.x *= scale; u.y *= scale; w.x *= scale; w.y *= scale;
    a[i] = u;
    b[i] = w;
  }
}

}  // namespace
