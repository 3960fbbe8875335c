// pic_device.h -- launch constants, kernel argument blocks and the per-particle device helpers of picstep.hip:
// periodic wrap, division by dx, cell location and shape weights, LDS gather / deposit, wave and block scans.
// Included by picstep.hip only (one translation unit; everything lives in its anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "picstep.h"

namespace {

#ifndef PIC_BLOCK
#define PIC_BLOCK 512               // sweep workgroup size: 512 beats 256 by 2.7 % and 128 by 11 % at config 2
#endif
constexpr int BLOCK = PIC_BLOCK;    // 8 waves of 64
constexpr int WAVES = BLOCK / 64;

enum Stage : int {
  ST_A = 0,        // drift(c) from x,v ; deposit ; nothing stored
  ST_B = 1,        // recompute q1 = x + (c_prev v) dt ; gather ; kick ; drift ; deposit ; store
  ST_C = 2,        // gather ; kick ; drift ; deposit ; store
  ST_D = 3,        // as C, then wrap, KE ; store wrapped x
  ST_REFRESH = 4,  // wrap x ; deposit ; KE ; store wrapped x            (pic.py:93-112 on reset)
  ST_PROBE = 5     // deposit positions of a scratch array, nothing stored (util.py:73-116 callers)
};

struct SweepArgs {
  long long N;        // particles per env
  long long ld;       // leading dimension of x, v
  long long chunk;    // particles per workgroup (multiple of BLOCK * VEC)
  int Ng;
  int nblk;           // workgroups per env
  int R;              // LDS mesh replicas per workgroup (1, 2 or 4)
  int reverse;        // walk environments and chunks from the far end (alternates sweep to sweep)
  int env0;           // first environment of this launch (launches may cover a group of environments)
  double L, dx, rdx, dt;   // rdx = 1/dx (for float particles: 1/(float)dx)
  double c_prev, c_cur, d_cur, c_next;
  double scale, n0;   // density scale n0 L / N / dx and mean density, for the in-prologue field solve
};

struct SolveArgs {
  long long N;
  int Ng;
  int nblk;
  double L, dx, n0;
  int env0;            // first environment of this launch
  double scale;        // n0 * L / N / dx, evaluated left to right as interpolate.py:18
  double N_over_L;
};

// ---------------------------------------------------------------------------------------------
// np.mod(np.mod(q, L), L): PIC.update_state wraps once (pic.py:139) and compute_n wraps the same
// array again in place (util.py:51) before CIC wraps its copy (interpolate.py:6), so a value that
// the first mod rounds up to exactly L ends as 0.  The three fast ranges are bit-identical to
// fmod-based np.mod (Sterbenz: q-L is exact for L <= q < 2L).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __noinline__ T wrap_periodic_far(T q, T L) {   // |q| beyond one box length: rare
  T r = fmod(q, L);
  if (r < T(0)) {
    r += L;
    if (r >= L) r = T(0);
  } else if (r == T(0)) {
    r = T(0);   // np.mod returns +0 for a zero remainder
  }
  return r;
}

template <typename T>
__device__ __forceinline__ T wrap_periodic(T q, T L) {
#ifdef PIC_EXP_BRANCHY_WRAP
  T r;
  if (q >= T(0) && q < L) {
    r = q;
  } else if (q >= L && q < L + L) {
    r = q - L;
  } else if (q < T(0) && q >= -L) {
    r = q + L;
    if (r >= L) r = T(0);
  } else {
    r = wrap_periodic_far(q, L);
  }
  return r;
#else
  // the three near ranges as selects (a particle moves a small fraction of L per sub-stage)
  T up = q + L;                       // q in [-L, 0)
  up = (up >= L) ? T(0) : up;         // tiny negative q: q + L rounds to L, the second mod gives 0
  T r = (q < T(0)) ? up : q;
  r = (q >= L) ? q - L : r;           // q in [L, 2L): exact (Sterbenz)
  if (__builtin_expect(!(q >= -L && q < L + L), 0)) r = wrap_periodic_far(q, L);
  return r;
#endif
}

// a / dx for the loop-invariant divisor dx, with rdx = 1/dx rounded once on the host: one Newton
// correction on the reciprocal product, q0 = a rdx; q = q0 + (a - q0 dx) rdx, both steps fused.
// The value before the final rounding is within ~2^-104 relative of a/dx, so the result is the
// IEEE quotient unless a/dx lies that close to a rounding boundary (probability ~2^-52 per
// operation, then 1 ulp off) -- 3 instructions instead of the ~12 of the full v_div_* sequence,
// which made sweep D division-bound.  tests/ check it bit for bit against true division.
template <typename T>
__device__ __forceinline__ T div_dx(T a, T dx, T rdx) {
#if defined(PIC_EXP_TRUEDIV)
  return a / dx;
#elif defined(PIC_EXP_RCPDIV)
  return a * rdx;
#else
  T q0 = a * rdx;
  T rem = fma(-q0, dx, a);
  return fma(rem, rdx, q0);
#endif
}

// Cell index and shape-function weights at position q.  j is the LDS index of the leftmost
// touched node (mesh node + OFF, OFF = 1 for TSC so that node -1 has a slot).
//   CIC (interpolate.py:6-13): jl = floor(xw/dx); wl = ((jl+1) dx - xw)/dx; wr = (xw - jl dx)/dx
//   TSC (interpolate.py:24-34): d = (xw - jm dx)/dx; wl = .5(1.5-d)^2; wm = .75-(d-1)^2; wr = .5(d-.5)^2
template <typename T, int SHAPE>
__device__ __forceinline__ void locate(T q, T L, T dx, T rdx, int Ng, T& xw, int& j, T (&w)[3], unsigned& bad) {
  xw = wrap_periodic(q, L);
  if (!(xw >= T(0) && xw < L)) {   // NaN / inf position: count it, park it on node 0, never index with it
    bad += 1u;
    xw = T(0);
  }
  T jf = floor(div_dx(xw, dx, rdx));
  j = (int)jf;
  // j == Ng happens when xw/dx rounds up to Ng (undefined in the reference: bincount grows a bin and
  // solve.py:32 raises); it is folded to node 0 with the weights of the unfolded index.
  if ((unsigned)j >= (unsigned)Ng) j = 0;
  if (SHAPE == PIC_CIC) {
    w[0] = div_dx((jf + T(1)) * dx - xw, dx, rdx);
    w[1] = div_dx(xw - jf * dx, dx, rdx);
    w[2] = T(0);
  } else {
    T d = div_dx(xw - jf * dx, dx, rdx);
    T a = T(1.5) - d, b = d - T(1), c = d - T(0.5);
    w[0] = T(0.5) * (a * a);
    w[1] = T(0.75) - b * b;
    w[2] = T(0.5) * (c * c);
  }
}

template <typename T, int SHAPE>
__device__ __forceinline__ T gather_field(const T* __restrict__ Es, int j, const T (&w)[3]) {
  T e = w[0] * Es[j] + w[1] * Es[j + 1];
  if (SHAPE == PIC_TSC) e = e + w[2] * Es[j + 2];
  return e;
}

// Packed fixed-point LDS accumulator for float32 particles (accum_dtype PIC_FIXED, CIC only).  A particle in
// cell j adds w_l = 1 - w_r to node j and w_r to node j+1, so per cell the pair (count, sum of w_r) carries the
// whole deposit: n_j = count_j - S_j + S_{j-1}.  Both live in one 64-bit word -- count in the top 20 bits,
// S in 2^-24 units below -- and one native ds_add_u64 replaces two ds_add_f64 (sweep D of config 3:
// 0.458 -> 0.395 ms; integer sums are also order-independent).  2^-24 is below the rounding of a float32
// weight; a workgroup handles fewer than 2^20 particles (pic_create sees to it), so neither field overflows.
using fix_t = unsigned long long;
constexpr int FX_FRAC = 24;
constexpr int FX_LOW = 44;

template <typename A, typename T, int SHAPE>
__device__ __forceinline__ void deposit(A* __restrict__ acc, int j, const T (&w)[3]) {
#ifdef PIC_EXP_NODEPOSIT   // timing experiment only: keep the operands alive, drop the LDS atomics
  asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(j));
  (void)acc;
#else
  if constexpr (std::is_same<A, fix_t>::value) {
    // one integer atomic per particle into its own cell: count in the high field, w_r in the low one
    static_assert(SHAPE == PIC_CIC, "the packed accumulator is CIC only");
    const float wr = fminf(fmaxf((float)w[1], 0.0f), 1.0f);
    atomicAdd(&acc[j], (1ull << FX_LOW) + (unsigned long long)(unsigned)(wr * (float)(1u << FX_FRAC) + 0.5f));
  } else {
    atomicAdd(&acc[j], (A)w[0]);
    atomicAdd(&acc[j + 1], (A)w[1]);
    if (SHAPE == PIC_TSC) atomicAdd(&acc[j + 2], (A)w[2]);
  }
#endif
}

#ifndef PIC_PIPE
#define PIC_PIPE 0      // tiles prefetched ahead of the one being pushed (experiment; the compiler sinks them)
#endif
#ifndef PIC_TILES
#define PIC_TILES 1     // 16-B tiles per lane per loop iteration
#endif
#define PIC_LOAD(p) (*(p))
#define PIC_STORE(v, p) (*(p) = (v))

template <typename T> struct VecOf;
typedef double pic_v2d __attribute__((ext_vector_type(2)));   // 16 B per lane either way
typedef float pic_v4f __attribute__((ext_vector_type(4)));
template <> struct VecOf<double> { using type = pic_v2d; static constexpr int n = 2; };
template <> struct VecOf<float> { using type = pic_v4f; static constexpr int n = 4; };

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}

__device__ __forceinline__ double wave_incl_scan(double v) {
  const int lane = threadIdx.x & 63;
  for (int off = 1; off < 64; off <<= 1) {
    double t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// exclusive prefix of `v` over a workgroup of NW waves (ws: NW doubles of LDS); total in `total`
template <int NW>
__device__ __forceinline__ double block_excl_scan(double v, double* ws, double& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double inc = wave_incl_scan(v);
  if (lane == 63) ws[w] = inc;
  __syncthreads();
  double off = 0.0, tot = 0.0;
  for (int i = 0; i < NW; ++i) {
    double s = ws[i];
    if (i < w) off += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return off + (inc - v);
}

template <int NW>
__device__ __forceinline__ double block_sum(double v, double* ws) {
  double w = wave_sum(v);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = w;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < NW; ++i) s += ws[i];
  __syncthreads();
  return s;
}

}  // namespace
