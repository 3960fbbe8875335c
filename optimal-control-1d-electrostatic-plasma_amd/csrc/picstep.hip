// picstep.hip -- MI355X (gfx950 / CDNA4) 1-D electrostatic PIC stepper behind the C ABI of
// include/picstep.h.  Written for wave64, LDS-resident per-block mesh tiles and coalesced SoA
// particle streams; there is no other backend and no CPU fallback.
//
// One environment step = PIC.update_state of the reference (src/env/pic.py:131-146), i.e. the
// Yoshida-4 composition of src/env/integration.py:60-75 restated as kick/drift sub-stages:
//
//   [sweep A: q1 = x + (c1 v) dt ; deposit(q1)]   -- normally NOT run: the previous sweep D (or the
//                                                    reset sweep) has already deposited this q1
//   solve   : E = field(deposit of q1) + E_ext
//   sweep B : p1 = v + (d1 (-E(q1))) dt ; q2 = q1 + (c2 p1) dt ; deposit(q2) ; store q2, p1
//   solve
//   sweep C : p2 = p1 + (d2 (-E(q2))) dt ; q3 = q2 + (c3 p2) dt ; deposit(q3) ; store
//   solve
//   sweep D : p3 = p2 + (d3 (-E(q3))) dt ; q4 = q3 + (c4 p3) dt ; x' = mod(q4, L) ; deposit(x') ;
//             KE partials ; store x', p3 ; deposit(next q1 = x' + (c1 p3) dt) into a second mesh
//   solve   : n, E_mesh (no E_ext), phi, KE, PE, PE_reward        (pic.py:145-146, util.py:119-147)
//
// 7 launches and 3 read+write passes over the particles per step (96 B per particle-step in fp64).
//
// Arithmetic inside a sub-stage keeps the reference's operand order and is compiled with
// -ffp-contract=off so that fp64 results track NumPy to rounding (tests/ hold the bounds).
//
// Deposit: every workgroup owns LDS copies of its environment's mesh (one per wave, `R` copies, two
// sets in sweep D), accumulates with LDS float atomics (ds_add_f64; ds_add_f32 is selectable but
// measured ~4x slower), then stores its partial mesh as one row of a slab [env][block][Ng] with plain
// coalesced stores.  The field-solve kernel sums the rows in a fixed order (no global atomics, no
// memset between sweeps), scales to a density and solves the periodic Poisson problem with two prefix
// scans.
//
// Compile-time switches (all off in the shipped build; results of each in profiles/experiments_r1.md):
// PIC_EXP_* are timing/diagnostic experiments, PIC_PIPE / PIC_TILES alternative loop forms.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

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

// Field tile for a sweep workgroup, computed in its own prologue (256 threads) from the slab the previous
// sweep wrote: density -> G = dx cumsum(n - n0) - mean -> E_j = -(G_{j+1/2} + G_{j-1/2})/2 (+ E_ext), the same
// scan solve as field_solve_kernel.  Every workgroup of an environment repeats it (the rows come from L2);
// in exchange a step needs no field-solve launch between sweeps.  sb: Ng doubles of LDS scratch.
template <typename T, int OFF>
__device__ __forceinline__ void prologue_field(const double* __restrict__ slab, int nblk, const double* __restrict__ ext,
                                               int Ng, double scale, double n0, double dx, double* __restrict__ sb,
                                               double* __restrict__ ws, T* __restrict__ Es) {
  const int tid = threadIdx.x;
  for (int j = tid; j < Ng; j += BLOCK) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
    int b = 0;
    for (; b + 7 < nblk; b += 8) {          // 8 independent loads in flight per lane
      s0 += slab[(size_t)b * Ng + j];
      s1 += slab[(size_t)(b + 1) * Ng + j];
      s2 += slab[(size_t)(b + 2) * Ng + j];
      s3 += slab[(size_t)(b + 3) * Ng + j];
      s4 += slab[(size_t)(b + 4) * Ng + j];
      s5 += slab[(size_t)(b + 5) * Ng + j];
      s6 += slab[(size_t)(b + 6) * Ng + j];
      s7 += slab[(size_t)(b + 7) * Ng + j];
    }
    for (; b < nblk; ++b) s0 += slab[(size_t)b * Ng + j];
    sb[j] = (((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))) * scale - n0;
  }
  __syncthreads();
  const int m = (Ng + BLOCK - 1) / BLOCK;
  const int lo = min(tid * m, Ng), hi = min(lo + m, Ng);
  double loc = 0.0;
  for (int j = lo; j < hi; ++j) loc += sb[j];
  double tot;
  double run = block_excl_scan<WAVES>(loc, ws, tot);
  loc = 0.0;
  for (int j = lo; j < hi; ++j) {
    run += sb[j];
    const double g = run * dx;
    sb[j] = g;
    loc += g;
  }
  const double gmean = block_sum<WAVES>(loc, ws) / (double)Ng;     // syncs: sb holds G everywhere
  for (int i = tid; i < Ng + 2; i += BLOCK) {
    int node = i - OFF;
    node = node < 0 ? node + Ng : (node >= Ng ? node - Ng : node);
    const double gp = sb[node] - gmean;
    const double gm = sb[node == 0 ? Ng - 1 : node - 1] - gmean;
    double E = -0.5 * (gp + gm);
    if (ext) E += ext[node];
    Es[i] = (T)E;
  }
  __syncthreads();
}

// One particle through one sub-stage.  Stages D / REFRESH also deposit the NEXT step's first drift
// position q1 = x' + (c1 p) dt into a second mesh (acc2), which is exactly what sweep A of the next
// step would deposit from the stored x', p -- so that sweep (a full read of x and v) is skipped.
template <typename T, typename A, int SHAPE, int STAGE>
__device__ __forceinline__ void push_one(T& xq, T& vp, const T* __restrict__ Es, A* __restrict__ acc,
                                         A* __restrict__ acc2, T L, T dx, T rdx, T dt, T c_prev, T c_cur, T d_cur,
                                         T c_next, int Ng, double& ke, unsigned& bad) {
#if defined(PIC_EXP_LEVEL) && PIC_EXP_LEVEL <= 1      // timing experiment: stream only (results are wrong)
  xq = xq + T(0); vp = vp + T(0);
  return;
#endif
  T w[3];
  T xw;
  int j;
  T q = xq, p = vp;
  if (STAGE == ST_A) {
    q = q + (c_cur * p) * dt;                                   // integration.py:42, c1
  } else if (STAGE == ST_B || STAGE == ST_C || STAGE == ST_D) {
    if (STAGE == ST_B) q = q + (c_prev * p) * dt;               // q1 again (it is never stored)
    locate<T, SHAPE>(q, L, dx, rdx, Ng, xw, j, w, bad);
#if defined(PIC_EXP_LEVEL) && PIC_EXP_LEVEL == 2      // timing experiment: arithmetic only, no LDS traffic
    T E = w[0] * T(0.25) + w[1] * T(0.5) + T(j) * T(1e-30);
#else
    T E = gather_field<T, SHAPE>(Es, j, w);                     // util.py:105 / pic.py:120
#endif
    p = p + (d_cur * (-E)) * dt;                                // integration.py:32, pic.py:127
    q = q + (c_cur * p) * dt;                                   // integration.py:42
  }
  locate<T, SHAPE>(q, L, dx, rdx, Ng, xw, j, w, bad);
#if defined(PIC_EXP_LEVEL) && (PIC_EXP_LEVEL == 2 || PIC_EXP_LEVEL == 3)   // no deposit (3: gather kept)
  asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(j));
#else
  deposit<A, T, SHAPE>(acc, j, w);
#endif
  if (STAGE == ST_D || STAGE == ST_REFRESH) {
    q = xw;                                                     // pic.py:139 (+ util.py:51)
    ke += (double)p * (double)p;
    T qn = q + (c_next * p) * dt;                               // next step's q1 (integration.py:42, c1)
    T xn;
    locate<T, SHAPE>(qn, L, dx, rdx, Ng, xn, j, w, bad);
    deposit<A, T, SHAPE>(acc2, j, w);
  }
  xq = q;
  vp = p;
}

// fold the periodic ghost slots and the replicas of one LDS mesh, store it as this workgroup's slab row
template <typename A, int SHAPE>
__device__ __forceinline__ void flush_mesh(const A* __restrict__ acc_all, int R, int stride, int Ng,
                                           double* __restrict__ row) {
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  if constexpr (std::is_same<A, fix_t>::value) {
    for (int c = threadIdx.x; c < Ng; c += BLOCK) {
      const int cm = c == 0 ? Ng - 1 : c - 1;
      unsigned long long own = 0ull, left = 0ull;
      for (int r = 0; r < R; ++r) {
        own += acc_all[(size_t)r * stride + c];
        left += acc_all[(size_t)r * stride + cm];
      }
      const long long mask = (1ll << FX_LOW) - 1;
      const long long q = ((long long)(own >> FX_LOW) << FX_FRAC) - ((long long)own & mask) + ((long long)left & mask);
      row[c] = (double)q * (1.0 / (double)(1 << FX_FRAC));      // exact: |q| < 2^45
    }
    return;
  }
  for (int c = threadIdx.x; c < Ng; c += BLOCK) {
    double s = 0.0;
    for (int r = 0; r < R; ++r) {
      const A* ar = acc_all + (size_t)r * stride;
      double t = (double)ar[c + OFF];
      if (SHAPE == PIC_CIC) {
        if (c == 0) t += (double)ar[Ng];
      } else {
        if (c == 0) t += (double)ar[Ng + 1];
        if (c == Ng - 1) t += (double)ar[0];
      }
      s += t;
    }
    row[c] = s;
  }
}

template <typename T, typename A, int SHAPE, int STAGE>
__global__ __launch_bounds__(BLOCK) void sweep_kernel(T* __restrict__ x, T* __restrict__ v,
                                                      const double* __restrict__ Ef,
                                                      const double* __restrict__ slab_in,
                                                      const double* __restrict__ ext_in,
                                                      double* __restrict__ part, double* __restrict__ part2,
                                                      double* __restrict__ ke_part,
                                                      unsigned long long* __restrict__ bad_count, SweepArgs a) {
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  constexpr int VEC = VecOf<T>::n;
  using V = typename VecOf<T>::type;
  constexpr bool kGather = (STAGE == ST_B || STAGE == ST_C || STAGE == ST_D);
  constexpr bool kStore = (STAGE == ST_B || STAGE == ST_C || STAGE == ST_D || STAGE == ST_REFRESH);
  constexpr bool kStoreV = (STAGE == ST_B || STAGE == ST_C || STAGE == ST_D);
  constexpr bool kReadV = (STAGE != ST_PROBE);
  constexpr bool kDual = (STAGE == ST_D || STAGE == ST_REFRESH);

  // LDS: [R meshes: acc][R meshes: acc2 (dual stages)][field tile Es]
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int Ng = a.Ng;
  const int stride = Ng + 2;
  const int nacc = (kDual ? 2 : 1) * a.R * stride;
  A* acc_all = reinterpret_cast<A*>(smem_raw);
  A* acc2_all = acc_all + (size_t)a.R * stride;
  T* Es = reinterpret_cast<T*>(smem_raw + (size_t)2 * a.R * stride * sizeof(A));
  __shared__ double red[WAVES];

  const int tid = threadIdx.x;
  // Consecutive sweeps walk memory in opposite directions: what the previous sweep wrote last (still
  // in the 256 MB Infinity Cache) is what this one reads first.
  const int env = a.env0 + (a.reverse ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y);
  const int blk = a.reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;

#ifdef PIC_EXP_STAMPB
  const unsigned long long sb0 = wall_clock64();
#endif
#if defined(PIC_EXP_LEVEL) && PIC_EXP_LEVEL <= 0
  constexpr bool kPrologue = false;      // timing experiment: no LDS zeroing / field tile / barrier
#else
  constexpr bool kPrologue = true;
#endif
  // the mesh region doubles as scratch of the in-prologue field solve, so it is zeroed after that solve
  const bool solve_here = kGather && slab_in != nullptr;
  if (kPrologue && !solve_here) for (int i = tid; i < nacc; i += BLOCK) acc_all[i] = A(0);
#ifdef PIC_EXP_STAMPB
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long sbz = wall_clock64();
#endif
  if (kGather && kPrologue) {
    if (slab_in) {
      // no field-solve launch ran before this sweep: solve here, with the (not yet zeroed) mesh region as scratch
      prologue_field<T, OFF>(slab_in + (size_t)env * a.nblk * Ng, a.nblk, ext_in ? ext_in + (size_t)env * Ng : nullptr,
                             Ng, a.scale, a.n0, a.dx, reinterpret_cast<double*>(smem_raw), red, Es);
      for (int i = tid; i < nacc; i += BLOCK) acc_all[i] = A(0);
    } else {
      const double* Ee = Ef + (size_t)env * Ng;
      for (int i = tid; i < stride; i += BLOCK) {
        int node = i - OFF;
        node = node < 0 ? node + Ng : (node >= Ng ? node - Ng : node);
        Es[i] = (T)Ee[node];
      }
    }
  }
#ifdef PIC_EXP_STAMPB
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long sbe = wall_clock64();
#endif
  if (kPrologue) __syncthreads();

  const int rep = (tid >> 6) & (a.R - 1);
  A* acc = acc_all + (size_t)rep * stride;
  A* acc2 = acc2_all + (size_t)rep * stride;
  const T L = (T)a.L, dx = (T)a.dx, rdx = (T)a.rdx, dt = (T)a.dt;
  const T c_prev = (T)a.c_prev, c_cur = (T)a.c_cur, d_cur = (T)a.d_cur, c_next = (T)a.c_next;

  T* xe = x + (size_t)env * a.ld;
  T* ve = v + (size_t)env * a.ld;
  const long long step = (long long)BLOCK * VEC;

  double ke = 0.0;
  unsigned bad = 0u;
#ifdef PIC_EXP_STAMPB
  const unsigned long long sb1 = wall_clock64();
#endif
  // A workgroup owns the runs blk, blk + nblk, blk + 2 nblk, ... of `chunk` particles of its environment.
  // chunk = ceil(N / nblk) gives every workgroup one contiguous region; a chunk of a few tiles interleaves
  // the workgroups of an environment, so that the addresses in flight form a compact moving window.
  for (long long begin = (long long)blk * a.chunk; begin < a.N; begin += (long long)a.nblk * a.chunk) {
  long long end = begin + a.chunk;
  if (end > a.N) end = a.N;
  long long i = begin + (long long)tid * VEC;
#if PIC_PIPE == 0
  // PIC_TILES tiles per lane per iteration: all their loads are issued before the first particle is
  // pushed, so a wave keeps PIC_TILES x 2 KB of requests in flight while it waits.
#ifdef PIC_EXP_STAMP   // diagnostic build: wall-clock (10 ns ticks) spent waiting for loads vs pushing, per wave 0
  unsigned long long st_mem = 0, st_cmp = 0, st_n = 0;
#endif
  for (; i + (long long)(PIC_TILES - 1) * step + VEC <= end; i += (long long)PIC_TILES * step) {
    V xv[PIC_TILES], vv[PIC_TILES];
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long st0 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int t = 0; t < PIC_TILES; ++t) {
      xv[t] = PIC_LOAD(reinterpret_cast<const V*>(xe + i + (long long)t * step));
      vv[t] = V{};
      if (kReadV) vv[t] = PIC_LOAD(reinterpret_cast<const V*>(ve + i + (long long)t * step));
    }
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st1 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int t = 0; t < PIC_TILES; ++t) {
      T* xs = reinterpret_cast<T*>(&xv[t]);
      T* vs = reinterpret_cast<T*>(&vv[t]);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        T pv = kReadV ? vs[k] : T(0);
        push_one<T, A, SHAPE, STAGE>(xs[k], pv, Es, acc, acc2, L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next, Ng, ke, bad);
        if (kReadV) vs[k] = pv;
      }
      if (kStore) {
        PIC_STORE(xv[t], reinterpret_cast<V*>(xe + i + (long long)t * step));
        if (kStoreV) PIC_STORE(vv[t], reinterpret_cast<V*>(ve + i + (long long)t * step));
      }
    }
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long st2 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    st_mem += st1 - st0;
    st_cmp += st2 - st1;
    st_n += 1;
#endif
  }
#ifdef PIC_EXP_STAMP
  if (tid == 0 && (STAGE == ST_C)) {
    atomicAdd(&bad_count[1], st_mem);
    atomicAdd(&bad_count[2], st_cmp);
    atomicAdd(&bad_count[3], st_n);
  }
#endif
  for (; i + VEC <= end; i += step) {          // leftover whole tiles
    V xv = PIC_LOAD(reinterpret_cast<const V*>(xe + i));
    V vv = {};
    if (kReadV) vv = PIC_LOAD(reinterpret_cast<const V*>(ve + i));
    T* xs = reinterpret_cast<T*>(&xv);
    T* vs = reinterpret_cast<T*>(&vv);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      T pv = kReadV ? vs[k] : T(0);
      push_one<T, A, SHAPE, STAGE>(xs[k], pv, Es, acc, acc2, L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next, Ng, ke, bad);
      if (kReadV) vs[k] = pv;
    }
    if (kStore) {
      PIC_STORE(xv, reinterpret_cast<V*>(xe + i));
      if (kStoreV) PIC_STORE(vv, reinterpret_cast<V*>(ve + i));
    }
  }
#else
  // Software pipeline (double buffer).  The next tile's loads are issued by inline asm BEFORE the current
  // tile is pushed: written as plain C loads, hipcc proves they cannot alias the stores and sinks them back
  // down to their use, so memory wait and push never overlap (stamped: 2.0 us + 1.8 us per iteration, all
  // waves of a SIMD in lockstep).  hipcc does not count asm loads in its own s_waitcnt, so the wait is
  // explicit: in issue order the younger VMEM operations at that point are exactly this iteration's stores
  // (kNumStores), hence vmcnt(kNumStores).  The "+v" ties keep every use of the prefetched registers behind
  // the wait (cdna_hip_programming.md 5.7).
  constexpr int kNumStores = kStore ? (kStoreV ? 2 : 1) : 0;
  bool have = (i + VEC <= end);
  V cx = {}, cv = {};
  if (have) {
    cx = PIC_LOAD(reinterpret_cast<const V*>(xe + i));
    if (kReadV) cv = PIC_LOAD(reinterpret_cast<const V*>(ve + i));
  }
#ifdef PIC_EXP_STAMP
  unsigned long long st_mem = 0, st_cmp = 0, st_n = 0;
#endif
  while (have) {
    const long long in = i + step;
    const bool hn = (in + VEC <= end);
    V nx = {}, nv = {};
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long st0 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (hn) {
      if (kReadV)
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off"
                     : "=&v"(nx), "=&v"(nv) : "v"(xe + in), "v"(ve + in) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(nx) : "v"(xe + in) : "memory");
    }
    T* xs = reinterpret_cast<T*>(&cx);
    T* vs = reinterpret_cast<T*>(&cv);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      T pv = kReadV ? vs[k] : T(0);
      push_one<T, A, SHAPE, STAGE>(xs[k], pv, Es, acc, acc2, L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next, Ng, ke, bad);
      if (kReadV) vs[k] = pv;
    }
    if (kStore) {
      PIC_STORE(cx, reinterpret_cast<V*>(xe + i));
      if (kStoreV) PIC_STORE(cv, reinterpret_cast<V*>(ve + i));
    }
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long st1 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (kNumStores == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(nx), "+v"(nv) : : "memory");
    else if (kNumStores == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(nx), "+v"(nv) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx), "+v"(nv) : : "memory");
#ifdef PIC_EXP_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long st2 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    st_cmp += st1 - st0;      // issue prefetch + push + issue stores
    st_mem += st2 - st1;      // residual wait for the prefetched tile
    st_n += 1;
#endif
    cx = nx;
    cv = nv;
    i = in;
    have = hn;
  }
#ifdef PIC_EXP_STAMP
  if (tid == 0 && (STAGE == ST_C)) {
    atomicAdd(&bad_count[1], st_mem);
    atomicAdd(&bad_count[2], st_cmp);
    atomicAdd(&bad_count[3], st_n);
  }
#endif
#endif
  for (long long k = i; k < end; ++k) {       // ragged tail (fewer than VEC particles left for this lane)
    T xq = xe[k];
    T pv = kReadV ? ve[k] : T(0);
    push_one<T, A, SHAPE, STAGE>(xq, pv, Es, acc, acc2, L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next, Ng, ke, bad);
    if (kStore) {
      xe[k] = xq;
      if (kStoreV) ve[k] = pv;
    }
  }
  }   // runs
#ifdef PIC_EXP_STAMPB
  const unsigned long long sb2 = wall_clock64();      // this wave's loop is done
#endif
  __syncthreads();
#ifdef PIC_EXP_STAMPB
  const unsigned long long sb3 = wall_clock64();      // every wave's loop is done
#endif

#if defined(PIC_EXP_LEVEL) && PIC_EXP_LEVEL <= -1
  if (ke < -1.0) part[0] = ke;           // timing experiment: no flush, no KE reduction
  return;
#endif
  const size_t rowi = ((size_t)env * a.nblk + blk) * Ng;
  flush_mesh<A, SHAPE>(acc_all, a.R, stride, Ng, part + rowi);
  if (kDual) flush_mesh<A, SHAPE>(acc2_all, a.R, stride, Ng, part2 + rowi);

  if (kDual) {
    double w = wave_sum(ke);
    if ((tid & 63) == 0) red[tid >> 6] = w;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int k = 0; k < WAVES; ++k) s += red[k];
      ke_part[(size_t)env * a.nblk + blk] = s;
    }
  }
  if (bad) atomicAdd(bad_count, (unsigned long long)bad);
#ifdef PIC_EXP_STAMPB
  if (STAGE == ST_C && (tid & 63) == 0) {
    const unsigned long long sb4 = wall_clock64();
    // [1] prologue+loop of this wave, [2] wait for the slowest wave of the workgroup, [3] flush; counts in [0]'s upper bits unused
    const bool late = (unsigned)(blockIdx.y * gridDim.x + blockIdx.x) >= 2048u;   // not in the first resident set
    if (late) {
      atomicAdd(&bad_count[1], ((sbz - sb0) << 32) | (sbe - sbz));     // zero LDS | field tile load
      atomicAdd(&bad_count[2], ((sb1 - sbe) << 32) | (sb2 - sb1));     // barrier | loop
      atomicAdd(&bad_count[3], ((sb3 - sb2) << 32) | 1ull);            // straggler wait | count
    }
    (void)sb4;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Field solve (one workgroup per environment).
// Replaces Gaussian_Elimination_Periodic + dense grad matvec (src/env/solve.py:27-53,
// src/env/util.py:99-103, pic.py:116-117).  With G_{j+1/2} = (phi_{j+1}-phi_j)/dx the 3-point
// periodic Poisson equation reads G_{j+1/2} - G_{j-1/2} = b_j dx, so G = cumsum(b) dx - mean and
// E_j = -(phi_{j+1}-phi_{j-1})/(2dx) = -(G_{j+1/2} + G_{j-1/2})/2.  phi follows from a second
// scan and is returned with zero mean.
// ---------------------------------------------------------------------------------------------
constexpr int SBLOCK = 1024;         // field-solve workgroup: 16 waves
constexpr int SWAVES = SBLOCK / 64;
constexpr int SGROUPS = 4;           // slab rows are summed by 4 groups of 256 lanes

// inputs / outputs of one field solve; a launch carries up to two independent ones (blockIdx.y), e.g. the
// post-step refresh of step s and the first force evaluation of step s+1
struct SolveIO {
  const double* part;      // slab [env][nblk][Ng] to reduce
  const double* ext;       // E_ext [env][Ng] or null
  const double* ke_part;   // [env][nblk] or null
  double *n, *Ef, *E, *phi, *KE, *PE, *PEr;   // any may be null
};

__global__ __launch_bounds__(SBLOCK) void field_solve_kernel(SolveIO io0, SolveIO io1, SolveArgs a) {
  const SolveIO io = blockIdx.y == 0 ? io0 : io1;
  const double* __restrict__ part = io.part;
  const double* __restrict__ E_ext = io.ext;
  const double* __restrict__ ke_part = io.ke_part;
  double* __restrict__ n_out = io.n;
  double* __restrict__ Ef_out = io.Ef;
  double* __restrict__ E_out = io.E;
  double* __restrict__ phi_out = io.phi;
  double* __restrict__ KE_out = io.KE;
  double* __restrict__ PE_out = io.PE;
  double* __restrict__ PEr_out = io.PEr;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* sb = reinterpret_cast<double*>(smem_raw);   // b, then G_{j+1/2}
  double* se = sb + a.Ng;                             // E, then phi
  double* sp = se + a.Ng;                             // [SGROUPS][Ng] partial row sums
  __shared__ double ws[SWAVES];

  const int tid = threadIdx.x;
  const int env = a.env0 + blockIdx.x;
  const int Ng = a.Ng;
  const int m = (Ng + SBLOCK - 1) / SBLOCK;
  const int lo = min(tid * m, Ng), hi = min(lo + m, Ng);

  // density: slab rows summed in a fixed order (group g takes rows g, g+4, ...; 4 loads in flight per
  // lane), scaled (interpolate.py:16-18), b = n - n0 (pic.py:116)
  const double* slab = part + (size_t)env * a.nblk * Ng;
  const int g = tid >> 8, lane = tid & 255;
  for (int j = lane; j < Ng; j += 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
    int b = g;
    for (; b + 7 * SGROUPS < a.nblk; b += 8 * SGROUPS) {       // 8 independent loads in flight per lane
      s0 += slab[(size_t)b * Ng + j];
      s1 += slab[(size_t)(b + SGROUPS) * Ng + j];
      s2 += slab[(size_t)(b + 2 * SGROUPS) * Ng + j];
      s3 += slab[(size_t)(b + 3 * SGROUPS) * Ng + j];
      s4 += slab[(size_t)(b + 4 * SGROUPS) * Ng + j];
      s5 += slab[(size_t)(b + 5 * SGROUPS) * Ng + j];
      s6 += slab[(size_t)(b + 6 * SGROUPS) * Ng + j];
      s7 += slab[(size_t)(b + 7 * SGROUPS) * Ng + j];
    }
    for (; b < a.nblk; b += SGROUPS) s0 += slab[(size_t)b * Ng + j];
    sp[g * Ng + j] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  }
  __syncthreads();
  for (int j = tid; j < Ng; j += SBLOCK) {
    double s = (sp[j] + sp[Ng + j]) + (sp[2 * Ng + j] + sp[3 * Ng + j]);
    double nj = s * a.scale;
    if (n_out) n_out[(size_t)env * Ng + j] = nj;
    sb[j] = nj - a.n0;
  }
  __syncthreads();

  // G_{j+1/2} = dx * inclusive_scan(b)
  double loc = 0.0;
  for (int j = lo; j < hi; ++j) loc += sb[j];
  double tot;
  double run = block_excl_scan<SWAVES>(loc, ws, tot);
  loc = 0.0;
  for (int j = lo; j < hi; ++j) {
    run += sb[j];
    double gj = run * a.dx;
    sb[j] = gj;
    loc += gj;
  }
  const double gmean = block_sum<SWAVES>(loc, ws) / (double)Ng;   // syncs: all of sb is G now

  // E_j = -(G_{j+1/2} + G_{j-1/2}) / 2, plus the external field for force evaluations (util.py:102-103)
  double e2 = 0.0;
  for (int j = tid; j < Ng; j += SBLOCK) {
    double gp = sb[j] - gmean;
    double gm = sb[j == 0 ? Ng - 1 : j - 1] - gmean;
    double E = -0.5 * (gp + gm);
    se[j] = E;
    double Et = E_ext ? E + E_ext[(size_t)env * Ng + j] : E;
    if (Ef_out) Ef_out[(size_t)env * Ng + j] = Et;
    if (E_out) E_out[(size_t)env * Ng + j] = Et;
    e2 += Et * Et;
  }
  const double S = block_sum<SWAVES>(e2, ws);
  if (tid == 0) {
    double pe = 0.5 * S * a.dx;                       // objective.py:33 / util.py:129
    if (PEr_out) PEr_out[env] = pe;
    if (PE_out) PE_out[env] = pe * a.N_over_L;        // util.py:130
  }

  if (KE_out) {
    double k = 0.0;
    for (int b = tid; b < a.nblk; b += SBLOCK) k += ke_part[(size_t)env * a.nblk + b];
    k = block_sum<SWAVES>(k, ws);
    if (tid == 0) KE_out[env] = 0.5 * k;              // util.py:144
  }

  if (phi_out) {
    // phi_{j+1} = phi_j + dx G_{j+1/2}: exclusive scan, then remove the mean
    loc = 0.0;
    for (int j = lo; j < hi; ++j) loc += (sb[j] - gmean) * a.dx;
    run = block_excl_scan<SWAVES>(loc, ws, tot);
    double ploc = 0.0;
    for (int j = lo; j < hi; ++j) {
      se[j] = run;
      ploc += run;
      run += (sb[j] - gmean) * a.dx;
    }
    const double pmean = block_sum<SWAVES>(ploc, ws) / (double)Ng;
    for (int j = tid; j < Ng; j += SBLOCK) phi_out[(size_t)env * Ng + j] = se[j] - pmean;
  }
}

// PIC.E (pic.py:120) and the CIC bookkeeping attributes (pic.py:104-107), on demand.
template <typename T, int SHAPE>
__global__ __launch_bounds__(BLOCK) void gather_E_kernel(const T* __restrict__ x, const double* __restrict__ E_mesh,
                                                         T* __restrict__ E_out, long long N, long long ld, int Ng,
                                                         double Ld, double dxd) {
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
  const T L = (T)Ld, dx = (T)dxd;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    T w[3], xw;
    int j;
    unsigned bad = 0;
    locate<T, SHAPE>(x[(size_t)env * ld + i], L, dx, T(1) / dx, Ng, xw, j, w, bad);
    E_out[(size_t)env * N + i] = gather_field<T, SHAPE>(Es, j, w);
  }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void cic_query_kernel(const T* __restrict__ x, long long N, int Ng, double Ld,
                                                          double dxd, long long* __restrict__ jl,
                                                          long long* __restrict__ jr, double* __restrict__ wl,
                                                          double* __restrict__ wr) {
  const T L = (T)Ld, dx = (T)dxd;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    T w[3], xw;
    int j;
    unsigned bad = 0;
    locate<T, PIC_CIC>(x[i], L, dx, T(1) / dx, Ng, xw, j, w, bad);
    if (jl) jl[i] = j;
    if (jr) jr[i] = (j + 1 == Ng) ? 0 : j + 1;
    if (wl) wl[i] = (double)w[0];
    if (wr) wr[i] = (double)w[1];
  }
}

// E_field.compute_E (src/control/actuator.py:54-63) for every environment:
// E_ext[e][j] = sum_m basis_cos[j][m] a[e][m] + sum_m basis_sin[j][m] a[e][M+m].  The basis tables come from the
// host mirror (they carry the reference's linspace(0, L, Ng) mesh, actuator.py:13).
__global__ __launch_bounds__(BLOCK) void actuator_kernel(const double* __restrict__ bc, const double* __restrict__ bs,
                                                         const double* __restrict__ act, double* __restrict__ ext,
                                                         int Ng, int M) {
  const int env = blockIdx.y;
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= Ng) return;
  const double* a = act + (size_t)env * 2 * M;
  double c = 0.0, s = 0.0;
  for (int m = 0; m < M; ++m) c += bc[(size_t)j * M + m] * a[m];
  for (int m = 0; m < M; ++m) s += bs[(size_t)j * M + m] * a[M + m];
  ext[(size_t)env * Ng + j] = c + s;
}

// compute_E_k_spectrum rows 1..M (src/interpret/spectrum.py:16): Ek[m] = fft(E_mesh)[m] / Ng * 2.
__global__ __launch_bounds__(BLOCK) void modes_kernel(const double* __restrict__ E_mesh, double* __restrict__ re,
                                                      double* __restrict__ im, int Ng, int M) {
  __shared__ double wr[WAVES], wi[WAVES];
  const int env = blockIdx.y, m = blockIdx.x + 1;
  double sr = 0.0, si = 0.0;
  for (int j = threadIdx.x; j < Ng; j += BLOCK) {
    // angle = 2 pi m j / Ng, reduced exactly in integers before the trig call
    const long long r = ((long long)m * j) % Ng;
    double sn, cs;
    sincospi(2.0 * (double)r / (double)Ng, &sn, &cs);
    const double e = E_mesh[(size_t)env * Ng + j];
    sr += e * cs;
    si -= e * sn;
  }
  sr = wave_sum(sr);
  si = wave_sum(si);
  if ((threadIdx.x & 63) == 0) { wr[threadIdx.x >> 6] = sr; wi[threadIdx.x >> 6] = si; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < WAVES; ++w) { a += wr[w]; b += wi[w]; }
    re[(size_t)env * M + (m - 1)] = a / Ng * 2.0;
    im[(size_t)env * M + (m - 1)] = b / Ng * 2.0;
  }
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
template <typename T>
__global__ __launch_bounds__(BLOCK) void sample_kernel(T* __restrict__ x, T* __restrict__ v, long long N, long long ld,
                                                       int kind, double a, double v0, double sigma, double A,
                                                       int n_mode, double L, unsigned long long seed) {
  const int env = blockIdx.y;
  const long long n_first = kind == 0 ? N / 2 : (long long)((double)N * (1.0 / (1.0 + a)));
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    double mu, sg;
    if (kind == 0) { mu = i < n_first ? v0 : -v0; sg = sigma; }
    else { mu = i < n_first ? 0.0 : v0; sg = i < n_first ? 1.0 : sigma; }
    const uint32_t k0 = (uint32_t)seed ^ (0x85EBCA6Bu * (uint32_t)(env + 1)), k1 = (uint32_t)(seed >> 32);
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
    x[(size_t)env * ld + i] = (T)xs;
    v[(size_t)env * ld + i] = (T)vs;
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
template <typename T>
__global__ __launch_bounds__(BLOCK) void phase_hist_kernel(const T* __restrict__ x, const T* __restrict__ v,
                                                           unsigned* __restrict__ counts, long long N, long long ld,
                                                           int nb, double L, double vmin, double vmax) {
  const int env = blockIdx.y;
  const double sx = (L - 0.0) / nb, sv = (vmax - vmin) / nb;      // np.linspace step = (stop - start) / div
  unsigned* c = counts + (size_t)env * nb * nb;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (long long)gridDim.x * BLOCK) {
    const int ix = hist_bin((double)x[(size_t)env * ld + i], 0.0, L, sx, nb);
    const int iv = hist_bin((double)v[(size_t)env * ld + i], vmin, vmax, sv, nb);
    if (ix >= 0 && iv >= 0) atomicAdd(&c[(size_t)ix * nb + iv], 1u);
  }
}

// Streaming ceiling of this box for the sweeps' access shape: read two arrays, write two arrays, 16 B
// per lane, same grid -- what a sweep would take if it did no arithmetic at all.
__global__ __launch_bounds__(BLOCK) void stream_probe_kernel(double2* __restrict__ a, double2* __restrict__ b,
                                                             long long n2, long long chunk2, double scale,
                                                             int reverse, int work) {
  const long long bid = reverse ? (long long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
  long long begin = bid * chunk2;
  long long end = begin + chunk2 < n2 ? begin + chunk2 : n2;
  for (long long i = begin + threadIdx.x; i < end; i += BLOCK) {
    double2 u = a[i], w = b[i];
    u.x *= scale; u.y *= scale; w.x *= scale; w.y *= scale;
    for (int k = 0; k < work; ++k) {   // experiment: dependent fp64 work between the load and the store
      u.x = fma(u.x, scale, w.x * 1e-300); w.x = fma(w.x, scale, u.y * 1e-300);
      u.y = fma(u.y, scale, w.y * 1e-300); w.y = fma(w.y, scale, u.x * 1e-300);
    }
    a[i] = u;
    b[i] = w;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
// one captured environment step (7 kernel nodes), valid for one external-field pointer and start parity
struct StepGraph {
  hipGraphExec_t exec = nullptr;
  const double* ext = nullptr;
  int parity = 0, parity_out = 0;
};

struct pic_handle {
  pic_config cfg{};
  int vec = 2;
  size_t esz = 8;          // particle element size
  size_t asz = 8;          // LDS accumulator element size
  long long ld = 0;
  long long chunk = 0;
  int nblk = 0;
  int R = 1;
  size_t sweep_lds = 0, solve_lds = 0;
  double dx = 0, scale = 0;
  double cs[4]{}, ds[4]{};
  hipStream_t stream = nullptr;       // the stream every call works on (own_stream, or the caller's)
  hipStream_t own_stream = nullptr;   // created by pic_create, destroyed by pic_destroy
  // Cache-resident schedule: pic_step walks the environments in groups whose particles fit the
  // Infinity Cache, each group running all its sweeps back to back on one of `wstreams`.
  std::vector<hipStream_t> wstreams;
  std::vector<hipEvent_t> join_ev;
  hipEvent_t fork_ev = nullptr;
  int group_envs = 0;             // 0 = no grouping (every launch covers all environments)
  bool group_major = true;        // all nsteps of a group before the next group (else step by step)
  void* x = nullptr;
  void* v = nullptr;
  void* scratch = nullptr;        // [env][ld] staging (eval_field positions, dense<->padded copies)
  double* part = nullptr;         // [env][nblk][Ng] deposit of the sweep just run
  double* part2 = nullptr;        // [env][nblk][Ng] deposit of the NEXT step's q1 (sweeps D / REFRESH)
  double* part_b = nullptr;       // second buffer for `part` (sweep C writes it while late C workgroups still read `part`)
  bool fused_solve = false;       // force-evaluation solves in the sweep prologues (4 launches per step)
  bool pair_solves = true;        // multi-step calls: final solve of step s + first force solve of s+1 in one launch
  int sweep_parity = 0;           // direction of the next push sweep
  bool use_graph = false;         // replay steps from hipGraphs (launch-bound sizes)
  std::vector<StepGraph> graphs;
  bool q1_ready = false;          // part2 matches the stored particles, dt and c1: sweep A can be skipped
  double* ke_part = nullptr;      // [env][nblk]
  double* Ef = nullptr;           // field used by the gathers (E + E_ext)
  double* n = nullptr;
  double* E_mesh = nullptr;
  double* phi = nullptr;
  double* ext = nullptr;          // device copy of a host E_ext / output of the device actuator
  double* basis = nullptr;        // [2][Ng][M] actuator tables (cos, sin)
  double* act = nullptr;          // [env][2M] actions
  double* modes = nullptr;        // [2][env][M] Fourier modes (re, im)
  int act_modes = 0;
  int modes_cap = 0;
  double* aux_n = nullptr;        // eval_field outputs
  double* aux_E = nullptr;
  double* aux_pe = nullptr;
  double* KE = nullptr;
  double* PE = nullptr;
  double* PEr = nullptr;
  double* h_scal = nullptr;       // pinned host staging for KE | PE | PE_reward
  unsigned long long* bad = nullptr;
  bool has_state = false;
  // profiling
  bool prof = false;
  std::vector<hipEvent_t> ev;     // pairs
  std::vector<int> ev_kind;
  double ms_sum[8]{};
  int64_t launches[8]{};
  std::string err;
};

namespace {

thread_local std::string g_create_error;

#define HIPCHK(h, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return PIC_EHIP;                                                                        \
    }                                                                                         \
  } while (0)

int fail(pic_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

void yoshida_coefficients(double (&c)[4], double (&d)[4]) {
  // integration.py:62-69, same expressions in the same order
  const double cbrt2 = std::pow(2.0, 1.0 / 3.0);
  const double w0 = (-1) * cbrt2 / (2 - cbrt2);
  const double w1 = 1 / (2 - cbrt2);
  c[0] = c[3] = 0.5 * w1;
  c[1] = c[2] = 0.5 * (w0 + w1);
  d[0] = 0.0;
  d[1] = d[3] = w1;
  d[2] = w0;
}

// what a sweep reads its field from and where its deposits go
struct SweepIO {
  const double* slab_in = nullptr;   // non-null: solve the field in the prologue from this slab (+ ext)
  const double* ext = nullptr;
  double* out = nullptr;             // slab receiving this sweep's deposit
  double* out2 = nullptr;            // slab receiving the next step's q1 deposit (dual stages)
};

// where a launch goes: stream + the block of environments it covers
struct Lane {
  hipStream_t stream;
  int env0, nenv;
  int parity;       // direction of this lane's next push sweep
};

template <typename T, typename A, int SHAPE, int STAGE>
void launch_sweep_t(pic_handle* h, const Lane& ln, const SweepIO& io, void* x, void* v, const SweepArgs& a) {
  dim3 grid(h->nblk, ln.nenv);
  hipLaunchKernelGGL((sweep_kernel<T, A, SHAPE, STAGE>), grid, dim3(BLOCK), h->sweep_lds, ln.stream,
                     static_cast<T*>(x), static_cast<T*>(v), h->Ef, io.slab_in, io.ext, io.out ? io.out : h->part,
                     io.out2 ? io.out2 : h->part2, h->ke_part, h->bad, a);
}

template <typename T, typename A, int SHAPE>
void launch_sweep_s(pic_handle* h, const Lane& ln, const SweepIO& io, int stage, void* x, void* v, const SweepArgs& a) {
  switch (stage) {
    case ST_A: launch_sweep_t<T, A, SHAPE, ST_A>(h, ln, io, x, v, a); break;
    case ST_B: launch_sweep_t<T, A, SHAPE, ST_B>(h, ln, io, x, v, a); break;
    case ST_C: launch_sweep_t<T, A, SHAPE, ST_C>(h, ln, io, x, v, a); break;
    case ST_D: launch_sweep_t<T, A, SHAPE, ST_D>(h, ln, io, x, v, a); break;
    case ST_REFRESH: launch_sweep_t<T, A, SHAPE, ST_REFRESH>(h, ln, io, x, v, a); break;
    default: launch_sweep_t<T, A, SHAPE, ST_PROBE>(h, ln, io, x, v, a); break;
  }
}

template <typename T, typename A>
void launch_sweep_i(pic_handle* h, const Lane& ln, const SweepIO& io, int stage, void* x, void* v, const SweepArgs& a) {
  if (h->cfg.interpol == PIC_TSC) launch_sweep_s<T, A, PIC_TSC>(h, ln, io, stage, x, v, a);
  else launch_sweep_s<T, A, PIC_CIC>(h, ln, io, stage, x, v, a);
}

// Per-launch HIP-event brackets on the handle's stream.  Events come from a pool that is only grown
// (never created inside a timed loop once warm) and recycled by prof_drain.
void prof_drain(pic_handle* h) {
  for (size_t i = 0; i < h->ev_kind.size(); ++i) {
    float ms = 0.f;
    hipEventSynchronize(h->ev[2 * i + 1]);
    if (hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]) == hipSuccess) {
      h->ms_sum[h->ev_kind[i]] += ms;
      h->launches[h->ev_kind[i]] += 1;
    }
  }
  h->ev_kind.clear();
}
void prof_reserve(pic_handle* h, size_t pairs) {
  while (h->ev.size() < 2 * pairs) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) break;
    h->ev.push_back(e);
  }
}
void prof_begin(pic_handle* h, hipStream_t st, int kind) {
  if (!h->prof) return;
  if (h->ev_kind.size() >= 16384) prof_drain(h);
  const size_t i = h->ev_kind.size();
  prof_reserve(h, i + 1);
  hipEventRecord(h->ev[2 * i], st);
  h->ev_kind.push_back(kind);
}
void prof_end(pic_handle* h, hipStream_t st) {
  if (!h->prof) return;
  hipEventRecord(h->ev[2 * (h->ev_kind.size() - 1) + 1], st);
}

Lane whole(pic_handle* h) { return Lane{h->stream, 0, h->cfg.num_envs, 0}; }

void launch_sweep(pic_handle* h, Lane& ln, int stage, void* x, void* v, double c_prev, double c_cur, double d_cur,
                  const SweepIO& io = SweepIO()) {
  SweepArgs a;
  a.c_next = h->cs[0];
  a.env0 = ln.env0;
  a.scale = h->scale;
  a.n0 = h->cfg.n0;
  a.N = h->cfg.N; a.ld = h->ld; a.chunk = h->chunk; a.Ng = h->cfg.Ng; a.nblk = h->nblk; a.R = h->R;
#ifdef PIC_EXP_NOREVERSE
  a.reverse = 0;
#else
  a.reverse = (stage <= ST_D) ? (ln.parity ^= 1) : 0;
#endif
  a.L = h->cfg.L; a.dx = h->dx; a.dt = h->cfg.dt;
  a.rdx = h->cfg.particle_dtype == PIC_F64 ? 1.0 / h->dx : (double)(1.0f / (float)h->dx);
  a.c_prev = c_prev; a.c_cur = c_cur; a.d_cur = d_cur;
  prof_begin(h, ln.stream, stage <= ST_D ? stage : 5);
  if (h->cfg.particle_dtype == PIC_F64) launch_sweep_i<double, double>(h, ln, io, stage, x, v, a);
  else if (h->cfg.accum_dtype == PIC_F64) launch_sweep_i<float, double>(h, ln, io, stage, x, v, a);
  else if (h->cfg.accum_dtype == PIC_FIXED) launch_sweep_s<float, fix_t, PIC_CIC>(h, ln, io, stage, x, v, a);
  else launch_sweep_i<float, float>(h, ln, io, stage, x, v, a);
  prof_end(h, ln.stream);
}

struct SolveOut {
  const double* slab = nullptr;   // default: h->part
  const double* ext = nullptr;
  const double* ke_part = nullptr;
  double* n = nullptr; double* Ef = nullptr; double* E = nullptr; double* phi = nullptr;
  double* KE = nullptr; double* PE = nullptr; double* PEr = nullptr;
};

SolveIO solve_io(pic_handle* h, const SolveOut& o) {
  return SolveIO{o.slab ? o.slab : h->part, o.ext, o.ke_part, o.n, o.Ef, o.E, o.phi, o.KE, o.PE, o.PEr};
}

// one solve, or two independent ones in the same launch (second = nullptr for one)
void launch_solve(pic_handle* h, const Lane& ln, const SolveOut& o, const SolveOut* second = nullptr) {
  SolveArgs a;
  a.env0 = ln.env0;
  a.N = h->cfg.N; a.Ng = h->cfg.Ng; a.nblk = h->nblk; a.L = h->cfg.L; a.dx = h->dx; a.n0 = h->cfg.n0;
  a.scale = h->scale; a.N_over_L = (double)h->cfg.N / h->cfg.L;
  const SolveIO io0 = solve_io(h, o);
  const SolveIO io1 = second ? solve_io(h, *second) : io0;
  prof_begin(h, ln.stream, 4);
  hipLaunchKernelGGL(field_solve_kernel, dim3(ln.nenv, second ? 2 : 1), dim3(SBLOCK), h->solve_lds, ln.stream, io0, io1, a);
  prof_end(h, ln.stream);
}

int refresh_fields(pic_handle* h) {
  Lane ln = whole(h);
  launch_sweep(h, ln, ST_REFRESH, h->x, h->v, 0, 0, 0);
  SolveOut o;
  o.ke_part = h->ke_part; o.n = h->n; o.Ef = h->Ef; o.E = h->E_mesh; o.phi = h->phi;
  o.KE = h->KE; o.PE = h->PE; o.PEr = h->PEr;
  launch_solve(h, ln, o);
  HIPCHK(h, hipGetLastError());
  h->q1_ready = true;   // ST_REFRESH also deposited the next step's q1 into part2
  return PIC_OK;
}

// copy a dense [env][N] caller array into a padded [env][ld] device array
int upload(pic_handle* h, void* dst_padded, const void* src, int mem_kind) {
  const size_t row = (size_t)h->cfg.N * h->esz;
  if (mem_kind == PIC_HOST) {
    HIPCHK(h, hipMemcpy2DAsync(dst_padded, (size_t)h->ld * h->esz, src, row, row, h->cfg.num_envs,
                               hipMemcpyHostToDevice, h->stream));
  } else {
    HIPCHK(h, hipMemcpy2DAsync(dst_padded, (size_t)h->ld * h->esz, src, row, row, h->cfg.num_envs,
                               hipMemcpyDeviceToDevice, h->stream));
  }
  return PIC_OK;
}

int download(pic_handle* h, void* dst, const void* src_padded, int mem_kind) {
  const size_t row = (size_t)h->cfg.N * h->esz;
  HIPCHK(h, hipMemcpy2DAsync(dst, row, src_padded, (size_t)h->ld * h->esz, row, h->cfg.num_envs,
                             mem_kind == PIC_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream));
  return PIC_OK;
}

}  // namespace

extern "C" {

int pic_abi_version(void) { return PICSTEP_ABI_VERSION; }

const char* pic_last_error(pic_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pic_create(const pic_config* cfg, pic_handle** out) {
  if (!cfg || !out) return fail(nullptr, PIC_EINVAL, "pic_create: null argument");
  *out = nullptr;
  if (cfg->N < 1 || cfg->Ng < 4 || cfg->num_envs < 1 || !(cfg->L > 0) || !(cfg->dt > 0) || !(cfg->n0 > 0))
    return fail(nullptr, PIC_EINVAL, "pic_create: need N>=1, Ng>=4, num_envs>=1, L>0, dt>0, n0>0");
  if (cfg->num_envs > 65535) return fail(nullptr, PIC_EINVAL, "pic_create: num_envs > 65535");
  if (cfg->particle_dtype != PIC_F64 && cfg->particle_dtype != PIC_F32)
    return fail(nullptr, PIC_EINVAL, "pic_create: particle_dtype must be PIC_F64 or PIC_F32");
  if (cfg->accum_dtype != PIC_F64 && cfg->accum_dtype != PIC_F32 && cfg->accum_dtype != PIC_FIXED)
    return fail(nullptr, PIC_EINVAL, "pic_create: accum_dtype must be PIC_F64, PIC_F32 or PIC_FIXED");
  if (cfg->accum_dtype != PIC_F64 && cfg->particle_dtype != PIC_F32)
    return fail(nullptr, PIC_EINVAL, "pic_create: a float32 or fixed-point accumulator needs float32 particles");
  if (cfg->interpol != PIC_CIC && cfg->interpol != PIC_TSC)
    return fail(nullptr, PIC_EINVAL, "pic_create: interpol must be PIC_CIC or PIC_TSC");
  if (cfg->accum_dtype == PIC_FIXED && cfg->interpol != PIC_CIC)
    return fail(nullptr, PIC_EINVAL, "pic_create: the fixed-point accumulator is CIC only");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(nullptr, PIC_EHIP, "pic_create: no HIP device visible (this library has no CPU path)");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, PIC_EINVAL, "pic_create: bad device_id");

  pic_handle* h = new (std::nothrow) pic_handle();
  if (!h) return fail(nullptr, PIC_ENOMEM, "pic_create: out of host memory");
  h->cfg = *cfg;
  h->esz = cfg->particle_dtype == PIC_F64 ? 8 : 4;
  h->asz = cfg->accum_dtype == PIC_F32 ? 4 : 8;
  h->vec = cfg->particle_dtype == PIC_F64 ? 2 : 4;
  h->dx = cfg->L / cfg->Ng;                                   // pic.py:36
  h->scale = cfg->n0 * cfg->L / (double)cfg->N / h->dx;       // interpolate.py:18
  yoshida_coefficients(h->cs, h->ds);
  h->ld = (cfg->N + 63) / 64 * 64;

  // workgroups per environment: enough in total to fill 256 CUs several times, at least one
  // BLOCK*VEC tile each
  const long long tile = (long long)BLOCK * h->vec;
  long long nblk = cfg->blocks_per_env;
  if (nblk <= 0) {
    const long long target_total = 8192;      // ~128 workgroups per env at 64 envs (profiles/experiments_r1.md)
    nblk = (target_total + cfg->num_envs - 1) / cfg->num_envs;
    // Large problems: >= 8 tiles per workgroup (amortises the prologue and the slab row).  Small, launch-bound
    // problems (profiles/smalln_bpe.py: N = 1e4 41 -> 30 us/step, N = 5e3 36 -> 28 us/step): one tile per
    // workgroup, at most 64 workgroups per environment so that the fused prologue solve stays cheap.
    const bool small = (double)cfg->N * cfg->num_envs <= 4.0e6;
    const long long tiles_min = small ? 1 : 8;
    long long max_by_work = (cfg->N + tiles_min * tile - 1) / (tiles_min * tile);
    if (small && max_by_work > 64) max_by_work = 64;
    if (nblk > max_by_work) nblk = max_by_work;
    if (nblk < 1) nblk = 1;
  }
  if (cfg->accum_dtype == PIC_FIXED) {       // count field of the packed accumulator: < 2^20 particles per workgroup
    const long long cap = (1ll << 20) - tile;
    if (nblk < (cfg->N + cap - 1) / cap) nblk = (cfg->N + cap - 1) / cap;
  }
  long long chunk = (cfg->N + nblk - 1) / nblk;
  chunk = (chunk + tile - 1) / tile * tile;
  nblk = (cfg->N + chunk - 1) / chunk;
  if (nblk > 65535) { delete h; return fail(nullptr, PIC_EINVAL, "pic_create: blocks_per_env too large"); }
  if (const char* rt = getenv("PICSTEP_RUN_TILES")) {      // experiment knob: interleave runs of this many tiles
    const long long k = atoll(rt);
    if (k > 0) chunk = k * tile;
  }
  h->chunk = chunk;
  h->nblk = (int)nblk;

  const size_t stride = (size_t)cfg->Ng + 2;
  // LDS: 2 R meshes (sweep D deposits two) + the field tile; up to 4 mesh copies (waves w and w+4 share
  // one: 8 copies measured no better) while the workgroup stays within 40 KB, i.e. 4 workgroups per CU
  h->R = WAVES < 4 ? WAVES : 4;
  if (const char* mr = getenv("PICSTEP_MAX_R")) { const int m = atoi(mr); while (m >= 1 && h->R > m) h->R >>= 1; }
  while (h->R > 1 && 2 * h->R * stride * h->asz + stride * h->esz > 40 * 1024) h->R >>= 1;
  h->sweep_lds = 2 * h->R * stride * h->asz + stride * h->esz;
  h->solve_lds = (2 + SGROUPS) * (size_t)cfg->Ng * sizeof(double);
  if (h->sweep_lds > 64 * 1024 || h->solve_lds > 150 * 1024) {
    delete h;
    return fail(nullptr, PIC_EINVAL, "pic_create: Ng too large for the LDS-resident mesh (max 2700 cells)");
  }

#define CREATE_CHK(call)                                                                      \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      std::string m = std::string("pic_create: " #call ": ") + hipGetErrorString(e_);         \
      pic_destroy(h);                                                                         \
      return fail(nullptr, e_ == hipErrorOutOfMemory ? PIC_ENOMEM : PIC_EHIP, m);            \
    }                                                                                         \
  } while (0)

  CREATE_CHK(hipSetDevice(cfg->device_id));
  if (h->solve_lds > 64 * 1024)
    CREATE_CHK(hipFuncSetAttribute((const void*)field_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->solve_lds));
  CREATE_CHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  {
    // hipGraph replay of a step (PICSTEP_GRAPH=1).  Off by default: at N = 1e4 the 7 dependent kernels cost
    // ~7 us each on the device whichever way they are launched (50.4 us/step eager, 53.9 us/step replayed).
    const char* ug = getenv("PICSTEP_GRAPH");
    h->use_graph = ug && atoi(ug) != 0;
  }
  {
    // group size: particles (x and v) of a group <= PICSTEP_GROUP_MB; 0 disables grouping
    const char* gm = getenv("PICSTEP_GROUP_MB");
    const char* ns = getenv("PICSTEP_STREAMS");
    const char* sm = getenv("PICSTEP_STEP_MAJOR");
    const double group_mb = gm ? atof(gm) : 0.0;
    const int nstreams = ns ? atoi(ns) : 2;
    h->group_major = !(sm && atoi(sm) != 0);
    const double env_mb = 2.0 * (double)h->ld * (double)h->esz / (1024.0 * 1024.0);
    if (group_mb > 0 && nstreams >= 1) {
      int G = (int)(group_mb / env_mb);
      if (G < 1) G = 1;
      if (G < cfg->num_envs) {
        h->group_envs = G;
        CREATE_CHK(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
        for (int w = 0; w < nstreams; ++w) {
          hipStream_t st;
          hipEvent_t ev;
          CREATE_CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
          h->wstreams.push_back(st);
          CREATE_CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
          h->join_ev.push_back(ev);
        }
      }
    }
  }
  const size_t pbytes = (size_t)cfg->num_envs * h->ld * h->esz;
  const size_t gbytes = (size_t)cfg->num_envs * cfg->Ng * sizeof(double);
  CREATE_CHK(hipMalloc(&h->x, pbytes));
  CREATE_CHK(hipMalloc(&h->v, pbytes));
  CREATE_CHK(hipMemsetAsync(h->x, 0, pbytes, h->stream));
  CREATE_CHK(hipMemsetAsync(h->v, 0, pbytes, h->stream));
  CREATE_CHK(hipMalloc((void**)&h->part, gbytes * h->nblk));
  CREATE_CHK(hipMalloc((void**)&h->part2, gbytes * h->nblk));
  CREATE_CHK(hipMalloc((void**)&h->part_b, gbytes * h->nblk));
  {
    // Every sweep workgroup re-sums its environment's nblk slab rows, so this pays where a step is
    // launch-bound (config 1: 50.0 -> 40.7 us/step) and is neutral once the sweeps are HBM-bound (config 2:
    // 1.229 vs 1.224 ms/step): on by default for small problems only.  PICSTEP_FUSED_SOLVE=0/1 overrides.
    if (const char* ps = getenv("PICSTEP_PAIR_SOLVES")) h->pair_solves = atoi(ps) != 0;
    const char* fs = getenv("PICSTEP_FUSED_SOLVE");
    h->fused_solve = fs ? atoi(fs) != 0 : (h->nblk <= 64 && (double)cfg->N * cfg->num_envs <= 4.0e6);
  }
  CREATE_CHK(hipMalloc((void**)&h->ke_part, (size_t)cfg->num_envs * h->nblk * sizeof(double)));
  CREATE_CHK(hipMemsetAsync(h->ke_part, 0, (size_t)cfg->num_envs * h->nblk * sizeof(double), h->stream));
  double** grids[] = {&h->Ef, &h->n, &h->E_mesh, &h->phi, &h->ext, &h->aux_n, &h->aux_E};
  for (double** g : grids) {
    CREATE_CHK(hipMalloc((void**)g, gbytes));
    CREATE_CHK(hipMemsetAsync(*g, 0, gbytes, h->stream));
  }
  // KE | PE | PE_reward live in one allocation so that a getter is a single small D2H copy into
  // pinned memory (a Python RL loop reads them every step)
  const size_t sbytes = (size_t)cfg->num_envs * sizeof(double);
  CREATE_CHK(hipMalloc((void**)&h->KE, 3 * sbytes));
  CREATE_CHK(hipMemsetAsync(h->KE, 0, 3 * sbytes, h->stream));
  h->PE = h->KE + cfg->num_envs;
  h->PEr = h->KE + 2 * (size_t)cfg->num_envs;
  CREATE_CHK(hipHostMalloc((void**)&h->h_scal, 3 * sbytes, hipHostMallocDefault));
  CREATE_CHK(hipMalloc((void**)&h->aux_pe, sbytes));
  CREATE_CHK(hipMemsetAsync(h->aux_pe, 0, sbytes, h->stream));
  CREATE_CHK(hipMalloc((void**)&h->bad, 4 * sizeof(unsigned long long)));   // [0] bad positions, [1..3] diagnostics
  CREATE_CHK(hipMemsetAsync(h->bad, 0, 4 * sizeof(unsigned long long), h->stream));
  CREATE_CHK(hipStreamSynchronize(h->stream));
#undef CREATE_CHK
  *out = h;
  return PIC_OK;
}

int pic_destroy(pic_handle* h) {
  if (!h) return PIC_OK;
  hipSetDevice(h->cfg.device_id);
  if (h->stream) hipStreamSynchronize(h->stream);
  prof_drain(h);
  for (hipEvent_t e : h->ev) hipEventDestroy(e);
  if (h->basis) hipFree(h->basis);
  if (h->act) hipFree(h->act);
  if (h->modes) hipFree(h->modes);
  void* bufs[] = {h->x, h->v, h->scratch, h->part, h->part2, h->part_b, h->ke_part, h->Ef, h->n, h->E_mesh, h->phi, h->ext,
                  h->aux_n, h->aux_E, h->aux_pe, h->KE, h->bad};
  for (void* b : bufs)
    if (b) hipFree(b);
  if (h->h_scal) hipHostFree(h->h_scal);
  for (StepGraph& g : h->graphs)
    if (g.exec) hipGraphExecDestroy(g.exec);
  for (hipStream_t st : h->wstreams) hipStreamDestroy(st);
  for (hipEvent_t ev : h->join_ev) hipEventDestroy(ev);
  if (h->fork_ev) hipEventDestroy(h->fork_ev);
  if (h->own_stream) hipStreamDestroy(h->own_stream);
  delete h;
  return PIC_OK;
}

int pic_set_stream(pic_handle* h, void* hip_stream) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->stream));      // drain the old stream: later work must see its results
  prof_drain(h);
  for (StepGraph& g : h->graphs)                    // captured on the old stream's behalf; cheap to rebuild
    if (g.exec) hipGraphExecDestroy(g.exec);
  h->graphs.clear();
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return PIC_OK;
}

int pic_sync(pic_handle* h) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PIC_OK;
}

int pic_set_particles(pic_handle* h, const void* x, const void* v, int mem_kind) {
  if (!h || !x || !v) return fail(h, PIC_EINVAL, "pic_set_particles: null argument");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int rc = upload(h, h->x, x, mem_kind);
  if (rc) return rc;
  rc = upload(h, h->v, v, mem_kind);
  if (rc) return rc;
  if (mem_kind == PIC_HOST) HIPCHK(h, hipStreamSynchronize(h->stream));
  h->has_state = true;
  h->q1_ready = false;
  return PIC_OK;
}

int pic_refresh(pic_handle* h) {
  if (!h) return PIC_EINVAL;
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_refresh: no particles loaded (call pic_reset first)");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return refresh_fields(h);
}

int pic_reset(pic_handle* h, const void* x0, const void* v0, int mem_kind) {
  int rc = pic_set_particles(h, x0, v0, mem_kind);
  if (rc) return rc;
  HIPCHK(h, hipMemsetAsync(h->bad, 0, sizeof(unsigned long long), h->stream));
  return refresh_fields(h);
}

int pic_step(pic_handle* h, const double* E_ext, int mem_kind, int nsteps) {
  if (!h) return PIC_EINVAL;
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_step: call pic_reset first");
  if (nsteps < 0) return fail(h, PIC_EINVAL, "pic_step: nsteps < 0");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const double* ext = nullptr;
  if (E_ext) {
    if (mem_kind == PIC_HOST) {
      HIPCHK(h, hipMemcpyAsync(h->ext, E_ext, (size_t)h->cfg.num_envs * h->cfg.Ng * sizeof(double),
                               hipMemcpyHostToDevice, h->stream));
      ext = h->ext;
    } else {
      ext = E_ext;
    }
  }
  // One environment step on a lane (7 launches): solve(q1 slab) B solve C solve D solve(final).
  const double* c = h->cs;
  const double* d = h->ds;
  const bool q1_ready = h->q1_ready;
  auto one_step = [&](Lane& ln, bool have_q1, bool ef_ready = false, bool pair_next = false) {
    if (h->fused_solve) {
      // 4 launches: the three force-evaluation solves run in the prologue of the sweep that needs the field.
      // Slabs are double-buffered: a workgroup that starts late must still find the rows of the PREVIOUS sweep.
      SweepIO io;
      if (!have_q1) {
        io.out = h->part2;                       // sweep A deposits q1 where sweep D normally leaves it
        launch_sweep(h, ln, ST_A, h->x, h->v, 0.0, c[0], 0.0, io);
      }
      io = SweepIO(); io.slab_in = h->part2; io.ext = ext; io.out = h->part;
      launch_sweep(h, ln, ST_B, h->x, h->v, c[0], c[1], d[1], io);
      io = SweepIO(); io.slab_in = h->part; io.ext = ext; io.out = h->part_b;
      launch_sweep(h, ln, ST_C, h->x, h->v, 0.0, c[2], d[2], io);
      io = SweepIO(); io.slab_in = h->part_b; io.ext = ext; io.out = h->part; io.out2 = h->part2;
      launch_sweep(h, ln, ST_D, h->x, h->v, 0.0, c[3], d[3], io);
      SolveOut o;           // post-step refresh: no external field (pic.py:114-117)
      o.ke_part = h->ke_part; o.n = h->n; o.E = h->E_mesh; o.phi = h->phi;
      o.KE = h->KE; o.PE = h->PE; o.PEr = h->PEr;
      launch_solve(h, ln, o);
      return;
    }
    SolveOut f;           // force evaluation: only the gather field is needed
    f.ext = ext; f.Ef = h->Ef;
    if (ef_ready) {
      // the previous step's last launch already solved for this step's first force evaluation
    } else if (have_q1) {   // the previous sweep D / reset already deposited q1 = x + (c1 v) dt
      SolveOut f1 = f;
      f1.slab = h->part2;
      launch_solve(h, ln, f1);
    } else {
      launch_sweep(h, ln, ST_A, h->x, h->v, 0.0, c[0], 0.0);
      launch_solve(h, ln, f);
    }
    launch_sweep(h, ln, ST_B, h->x, h->v, c[0], c[1], d[1]);
    launch_solve(h, ln, f);
    launch_sweep(h, ln, ST_C, h->x, h->v, 0.0, c[2], d[2]);
    launch_solve(h, ln, f);
    launch_sweep(h, ln, ST_D, h->x, h->v, 0.0, c[3], d[3]);
    SolveOut o;           // post-step refresh: no external field (pic.py:114-117)
    o.ke_part = h->ke_part; o.n = h->n; o.E = h->E_mesh; o.phi = h->phi;
    o.KE = h->KE; o.PE = h->PE; o.PEr = h->PEr;
    if (pair_next) {      // another step follows with the same E_ext: solve its first force field in this launch too
      SolveOut f1 = f;
      f1.slab = h->part2;
      launch_solve(h, ln, o, &f1);
    } else {
      launch_solve(h, ln, o);
    }
  };

  const int E = h->cfg.num_envs;
  const int G = h->group_envs;
  if (G <= 0 || G >= E || h->wstreams.empty() || nsteps == 0) {
    Lane ln = whole(h);
    ln.parity = h->sweep_parity;
    for (int s = 0; s < nsteps; ++s) {
      const bool have_q1 = s > 0 || q1_ready;
      if (!h->use_graph || !have_q1) {
        // unfused schedule: 7 launches for a lone step, 6 per step inside a multi-step call
        const bool pairing = !h->fused_solve && h->pair_solves;
        one_step(ln, have_q1, /*ef_ready=*/pairing && s > 0, /*pair_next=*/pairing && s + 1 < nsteps);
        continue;
      }
      if (h->prof) {
        one_step(ln, have_q1);
        continue;
      }
      // Launch-bound regime (small environments): the 7 launches of a step are replayed from a hipGraph.
      // A graph bakes in the kernel arguments, i.e. the external-field pointer and the direction parity
      // the step starts with (it flips every step), so executables are cached per (ext, parity).
      StepGraph* g = nullptr;
      for (StepGraph& c : h->graphs)
        if (c.exec && c.ext == ext && c.parity == ln.parity) { g = &c; break; }
      if (!g) {
        if (h->graphs.size() >= 8) {               // bounded cache: drop the oldest executable
          hipGraphExecDestroy(h->graphs.front().exec);
          h->graphs.erase(h->graphs.begin());
        }
        hipGraph_t graph = nullptr;
        Lane cap = ln;
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        one_step(cap, true);
        HIPCHK(h, hipStreamEndCapture(h->stream, &graph));
        StepGraph ng;
        ng.ext = ext; ng.parity = ln.parity; ng.parity_out = cap.parity;
        hipError_t ge = hipGraphInstantiate(&ng.exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ge != hipSuccess) return fail(h, PIC_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ge));
        h->graphs.push_back(ng);
        g = &h->graphs.back();
      }
      HIPCHK(h, hipGraphLaunch(g->exec, h->stream));
      ln.parity = g->parity_out;
    }
    h->sweep_parity = ln.parity;
  } else {
    // Environments are independent: walk them in cache-sized groups, neighbouring groups on different
    // streams so that one group's field solves and kernel tails hide under the other's sweeps.
    const int S = (int)h->wstreams.size();
    const int ngroups = (E + G - 1) / G;
    HIPCHK(h, hipEventRecord(h->fork_ev, h->stream));
    for (int w = 0; w < S; ++w) HIPCHK(h, hipStreamWaitEvent(h->wstreams[w], h->fork_ev, 0));
    int parity = h->sweep_parity;
    if (h->group_major) {
      for (int g = 0; g < ngroups; ++g) {
        Lane ln{h->wstreams[g % S], g * G, (g + 1) * G <= E ? G : E - g * G, h->sweep_parity};
        for (int s = 0; s < nsteps; ++s) one_step(ln, s > 0 || q1_ready);
        parity = ln.parity;
      }
    } else {
      for (int s = 0; s < nsteps; ++s)
        for (int g = 0; g < ngroups; ++g) {
          Lane ln{h->wstreams[g % S], g * G, (g + 1) * G <= E ? G : E - g * G, (h->sweep_parity + 3 * s) & 1};
          one_step(ln, s > 0 || q1_ready);
          parity = ln.parity;
        }
    }
    h->sweep_parity = parity;
    for (int w = 0; w < S; ++w) {
      HIPCHK(h, hipEventRecord(h->join_ev[w], h->wstreams[w]));
      HIPCHK(h, hipStreamWaitEvent(h->stream, h->join_ev[w], 0));
    }
  }
  if (nsteps > 0) h->q1_ready = true;
  HIPCHK(h, hipGetLastError());
  return PIC_OK;
}

int pic_get_particles(pic_handle* h, void* x, void* v, int mem_kind) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int rc = PIC_OK;
  if (x) rc = download(h, x, h->x, mem_kind);
  if (!rc && v) rc = download(h, v, h->v, mem_kind);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PIC_OK;
}

int pic_device_ptrs(pic_handle* h, void** x, void** v, int64_t* ld, double** n, double** E_mesh, double** phi,
                    double** KE, double** PE, double** PE_reward) {
  if (!h) return PIC_EINVAL;
  if (x) *x = h->x;
  if (v) *v = h->v;
  if (ld) *ld = h->ld;
  if (n) *n = h->n;
  if (E_mesh) *E_mesh = h->E_mesh;
  if (phi) *phi = h->phi;
  if (KE) *KE = h->KE;
  if (PE) *PE = h->PE;
  if (PE_reward) *PE_reward = h->PEr;
  return PIC_OK;
}

int pic_get_fields(pic_handle* h, double* n, double* E_mesh, double* phi) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t gbytes = (size_t)h->cfg.num_envs * h->cfg.Ng * sizeof(double);
  if (n) HIPCHK(h, hipMemcpyAsync(n, h->n, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (E_mesh) HIPCHK(h, hipMemcpyAsync(E_mesh, h->E_mesh, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (phi) HIPCHK(h, hipMemcpyAsync(phi, h->phi, gbytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PIC_OK;
}

int pic_get_energies(pic_handle* h, double* KE, double* PE, double* PE_reward) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t E = (size_t)h->cfg.num_envs, b = E * sizeof(double);
  HIPCHK(h, hipMemcpyAsync(h->h_scal, h->KE, 3 * b, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (KE) std::memcpy(KE, h->h_scal, b);
  if (PE) std::memcpy(PE, h->h_scal + E, b);
  if (PE_reward) std::memcpy(PE_reward, h->h_scal + 2 * E, b);
  return PIC_OK;
}

static int ensure_scratch(pic_handle* h) {
  if (h->scratch) return PIC_OK;
  const size_t pbytes = (size_t)h->cfg.num_envs * h->ld * h->esz;
  HIPCHK(h, hipMalloc(&h->scratch, pbytes));
  HIPCHK(h, hipMemsetAsync(h->scratch, 0, pbytes, h->stream));
  return PIC_OK;
}

int pic_gather_E(pic_handle* h, void* E_particles, int mem_kind) {
  if (!h || !E_particles) return fail(h, PIC_EINVAL, "pic_gather_E: null argument");
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_gather_E: call pic_reset first");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  void* dst = E_particles;
  if (mem_kind == PIC_HOST) {
    int rc = ensure_scratch(h);
    if (rc) return rc;
    dst = h->scratch;     // dense [env][N] fits in [env][ld]
  }
  long long gx = (h->cfg.N + BLOCK - 1) / BLOCK;
  if (gx > 1024) gx = 1024;
  dim3 grid((unsigned)gx, h->cfg.num_envs);
  const size_t lds = ((size_t)h->cfg.Ng + 2) * h->esz;
  const bool tsc = h->cfg.interpol == PIC_TSC;
  if (h->cfg.particle_dtype == PIC_F64) {
    if (tsc) hipLaunchKernelGGL((gather_E_kernel<double, PIC_TSC>), grid, dim3(BLOCK), lds, h->stream, (const double*)h->x, h->E_mesh, (double*)dst, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
    else hipLaunchKernelGGL((gather_E_kernel<double, PIC_CIC>), grid, dim3(BLOCK), lds, h->stream, (const double*)h->x, h->E_mesh, (double*)dst, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
  } else {
    if (tsc) hipLaunchKernelGGL((gather_E_kernel<float, PIC_TSC>), grid, dim3(BLOCK), lds, h->stream, (const float*)h->x, h->E_mesh, (float*)dst, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
    else hipLaunchKernelGGL((gather_E_kernel<float, PIC_CIC>), grid, dim3(BLOCK), lds, h->stream, (const float*)h->x, h->E_mesh, (float*)dst, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
  }
  HIPCHK(h, hipGetLastError());
  if (mem_kind == PIC_HOST)
    HIPCHK(h, hipMemcpyAsync(E_particles, dst, (size_t)h->cfg.num_envs * h->cfg.N * h->esz, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PIC_OK;
}

int pic_get_cic(pic_handle* h, int env, int64_t* indx_l, int64_t* indx_r, double* weight_l, double* weight_r) {
  if (!h) return PIC_EINVAL;
  if (env < 0 || env >= h->cfg.num_envs) return fail(h, PIC_EINVAL, "pic_get_cic: env out of range");
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_get_cic: call pic_reset first");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const long long N = h->cfg.N;
  long long* dj = nullptr;
  double* dw = nullptr;
  HIPCHK(h, hipMalloc((void**)&dj, 2 * N * sizeof(long long)));
  if (hipMalloc((void**)&dw, 2 * N * sizeof(double)) != hipSuccess) { hipFree(dj); return fail(h, PIC_ENOMEM, "pic_get_cic: hipMalloc"); }
  long long gx = (N + BLOCK - 1) / BLOCK;
  if (gx > 1024) gx = 1024;
  const char* xe = (const char*)h->x + (size_t)env * h->ld * h->esz;
  if (h->cfg.particle_dtype == PIC_F64)
    hipLaunchKernelGGL(cic_query_kernel<double>, dim3((unsigned)gx), dim3(BLOCK), 0, h->stream, (const double*)xe, N, h->cfg.Ng, h->cfg.L, h->dx, dj, dj + N, dw, dw + N);
  else
    hipLaunchKernelGGL(cic_query_kernel<float>, dim3((unsigned)gx), dim3(BLOCK), 0, h->stream, (const float*)xe, N, h->cfg.Ng, h->cfg.L, h->dx, dj, dj + N, dw, dw + N);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && indx_l) e = hipMemcpyAsync(indx_l, dj, N * sizeof(long long), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && indx_r) e = hipMemcpyAsync(indx_r, dj + N, N * sizeof(long long), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && weight_l) e = hipMemcpyAsync(weight_l, dw, N * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && weight_r) e = hipMemcpyAsync(weight_r, dw + N, N * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  hipFree(dj);
  hipFree(dw);
  if (e != hipSuccess) return fail(h, PIC_EHIP, std::string("pic_get_cic: ") + hipGetErrorString(e));
  return PIC_OK;
}

int pic_eval_field(pic_handle* h, const void* x, int mem_kind, const double* E_ext, double* n, double* E_mesh,
                   double* half_sum_E2_dx) {
  if (!h || !x) return fail(h, PIC_EINVAL, "pic_eval_field: null argument");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int rc = ensure_scratch(h);
  if (rc) return rc;
  rc = upload(h, h->scratch, x, mem_kind);
  if (rc) return rc;
  const size_t gbytes = (size_t)h->cfg.num_envs * h->cfg.Ng * sizeof(double);
  const double* ext = nullptr;
  if (E_ext) {
    HIPCHK(h, hipMemcpyAsync(h->ext, E_ext, gbytes, hipMemcpyHostToDevice, h->stream));
    ext = h->ext;
  }
  Lane ln = whole(h);
  launch_sweep(h, ln, ST_PROBE, h->scratch, h->scratch, 0, 0, 0);
  SolveOut o;
  o.ext = ext; o.n = h->aux_n; o.E = h->aux_E; o.PEr = h->aux_pe;
  launch_solve(h, ln, o);
  HIPCHK(h, hipGetLastError());
  if (n) HIPCHK(h, hipMemcpyAsync(n, h->aux_n, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (E_mesh) HIPCHK(h, hipMemcpyAsync(E_mesh, h->aux_E, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (half_sum_E2_dx)
    HIPCHK(h, hipMemcpyAsync(half_sum_E2_dx, h->aux_pe, (size_t)h->cfg.num_envs * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PIC_OK;
}

int pic_set_actuator(pic_handle* h, int max_mode, const double* basis_cos, const double* basis_sin) {
  if (!h || !basis_cos || !basis_sin || max_mode < 1 || max_mode > 64)
    return fail(h, PIC_EINVAL, "pic_set_actuator: need 1 <= max_mode <= 64 and both basis tables");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->basis) { hipFree(h->basis); h->basis = nullptr; }
  if (h->act) { hipFree(h->act); h->act = nullptr; }
  const size_t tb = (size_t)h->cfg.Ng * max_mode * sizeof(double);
  HIPCHK(h, hipMalloc((void**)&h->basis, 2 * tb));
  HIPCHK(h, hipMalloc((void**)&h->act, (size_t)h->cfg.num_envs * 2 * max_mode * sizeof(double)));
  HIPCHK(h, hipMemcpyAsync(h->basis, basis_cos, tb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync((char*)h->basis + tb, basis_sin, tb, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->act_modes = max_mode;
  return PIC_OK;
}

int pic_step_actions(pic_handle* h, const double* actions, int mem_kind, int nsteps) {
  if (!h || !actions) return fail(h, PIC_EINVAL, "pic_step_actions: null argument");
  if (!h->act_modes) return fail(h, PIC_ESTATE, "pic_step_actions: call pic_set_actuator first");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int M = h->act_modes, Ng = h->cfg.Ng;
  const double* a = actions;
  if (mem_kind == PIC_HOST) {
    HIPCHK(h, hipMemcpyAsync(h->act, actions, (size_t)h->cfg.num_envs * 2 * M * sizeof(double), hipMemcpyHostToDevice,
                             h->stream));
    a = h->act;
  }
  hipLaunchKernelGGL(actuator_kernel, dim3((Ng + BLOCK - 1) / BLOCK, h->cfg.num_envs), dim3(BLOCK), 0, h->stream,
                     h->basis, h->basis + (size_t)Ng * M, a, h->ext, Ng, M);
  HIPCHK(h, hipGetLastError());
  return pic_step(h, h->ext, PIC_DEVICE, nsteps);
}

int pic_get_modes(pic_handle* h, int max_mode, double* re, double* im, int mem_kind) {
  if (!h || max_mode < 1 || max_mode >= h->cfg.Ng) return fail(h, PIC_EINVAL, "pic_get_modes: bad max_mode");
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_get_modes: call pic_reset first");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t nb = (size_t)h->cfg.num_envs * max_mode * sizeof(double);
  if (max_mode > h->modes_cap) {          // (re)allocate only when a larger mode count is asked for
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->modes) { hipFree(h->modes); h->modes = nullptr; }
    HIPCHK(h, hipMalloc((void**)&h->modes, 2 * nb));
    h->modes_cap = max_mode;
  }
  double* dre = h->modes;
  double* dim_ = h->modes + (size_t)h->cfg.num_envs * max_mode;
  hipLaunchKernelGGL(modes_kernel, dim3(max_mode, h->cfg.num_envs), dim3(BLOCK), 0, h->stream, h->E_mesh, dre, dim_,
                     h->cfg.Ng, max_mode);
  HIPCHK(h, hipGetLastError());
  const hipMemcpyKind k = mem_kind == PIC_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (re) HIPCHK(h, hipMemcpyAsync(re, dre, nb, k, h->stream));
  if (im) HIPCHK(h, hipMemcpyAsync(im, dim_, nb, k, h->stream));
  if (mem_kind == PIC_HOST) HIPCHK(h, hipStreamSynchronize(h->stream));   // device outputs stay stream-ordered
  return PIC_OK;
}

int pic_reset_sampled(pic_handle* h, int kind, double a, double v0, double sigma, double A, int n_mode,
                      uint64_t seed) {
  if (!h || (kind != 0 && kind != 1) || !(sigma > 0) || (kind == 1 && !(a >= 0)))
    return fail(h, PIC_EINVAL, "pic_reset_sampled: kind must be 0 (two-stream) or 1 (bump-on-tail), sigma > 0, a >= 0");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  long long gx = (h->cfg.N + BLOCK - 1) / BLOCK;
  if (gx > 2048) gx = 2048;
  dim3 grid((unsigned)gx, h->cfg.num_envs);
  if (h->cfg.particle_dtype == PIC_F64)
    hipLaunchKernelGGL(sample_kernel<double>, grid, dim3(BLOCK), 0, h->stream, (double*)h->x, (double*)h->v, h->cfg.N,
                       h->ld, kind, a, v0, sigma, A, n_mode, h->cfg.L, (unsigned long long)seed);
  else
    hipLaunchKernelGGL(sample_kernel<float>, grid, dim3(BLOCK), 0, h->stream, (float*)h->x, (float*)h->v, h->cfg.N,
                       h->ld, kind, a, v0, sigma, A, n_mode, h->cfg.L, (unsigned long long)seed);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemsetAsync(h->bad, 0, sizeof(unsigned long long), h->stream));
  h->has_state = true;
  h->q1_ready = false;
  return refresh_fields(h);
}

int pic_phase_histogram(pic_handle* h, int nbins, double vmin, double vmax, uint32_t* counts) {
  if (!h || !counts || nbins < 1 || nbins > 4096 || !(vmax > vmin))
    return fail(h, PIC_EINVAL, "pic_phase_histogram: need counts, 1 <= nbins <= 4096, vmax > vmin");
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_phase_histogram: call pic_reset first");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t nb = (size_t)h->cfg.num_envs * nbins * nbins * sizeof(unsigned);
  unsigned* d = nullptr;
  HIPCHK(h, hipMalloc((void**)&d, nb));
  hipError_t e = hipMemsetAsync(d, 0, nb, h->stream);
  long long gx = (h->cfg.N + BLOCK - 1) / BLOCK;
  if (gx > 2048) gx = 2048;
  dim3 grid((unsigned)gx, h->cfg.num_envs);
  if (e == hipSuccess) {
    if (h->cfg.particle_dtype == PIC_F64)
      hipLaunchKernelGGL(phase_hist_kernel<double>, grid, dim3(BLOCK), 0, h->stream, (const double*)h->x,
                         (const double*)h->v, d, h->cfg.N, h->ld, nbins, h->cfg.L, vmin, vmax);
    else
      hipLaunchKernelGGL(phase_hist_kernel<float>, grid, dim3(BLOCK), 0, h->stream, (const float*)h->x,
                         (const float*)h->v, d, h->cfg.N, h->ld, nbins, h->cfg.L, vmin, vmax);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(counts, d, nb, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  hipFree(d);
  if (e != hipSuccess) return fail(h, PIC_EHIP, std::string("pic_phase_histogram: ") + hipGetErrorString(e));
  return PIC_OK;
}

int pic_stream_probe(pic_handle* h, int repeats, double* gbytes_per_s) {
  if (!h || !gbytes_per_s || repeats < 1) return fail(h, PIC_EINVAL, "pic_stream_probe: bad argument");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  size_t pbytes = (size_t)h->cfg.num_envs * h->ld * h->esz;
  if (const char* mb = getenv("PICSTEP_PROBE_MB")) pbytes = (size_t)atoll(mb) << 20;   // experiment knobs
  const int work = getenv("PICSTEP_PROBE_WORK") ? atoi(getenv("PICSTEP_PROBE_WORK")) : 0;
  void *a = nullptr, *b = nullptr;
  HIPCHK(h, hipMalloc(&a, pbytes));
  if (hipMalloc(&b, pbytes) != hipSuccess) { hipFree(a); return fail(h, PIC_ENOMEM, "pic_stream_probe: hipMalloc"); }
  hipMemsetAsync(a, 0, pbytes, h->stream);
  hipMemsetAsync(b, 0, pbytes, h->stream);
  const long long n2 = (long long)(pbytes / sizeof(double2));
  long long nb = n2 / ((long long)BLOCK * 31);       // ~31 tiles per lane, like a sweep workgroup
  if (nb < 256) nb = 256;
  const long long chunk2 = (n2 + nb - 1) / nb;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, h->stream, (double2*)a, (double2*)b, n2, chunk2, 1.0, 1, work);
  hipEventRecord(e0, h->stream);
  for (int r = 0; r < repeats; ++r)
    hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, h->stream, (double2*)a, (double2*)b, n2, chunk2, 1.0, r & 1, work);
  hipEventRecord(e1, h->stream);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(a);
  hipFree(b);
  if (e != hipSuccess) return fail(h, PIC_EHIP, std::string("pic_stream_probe: ") + hipGetErrorString(e));
  *gbytes_per_s = 4.0 * (double)pbytes * repeats / (ms * 1e-3) / 1e9;
  return PIC_OK;
}

int pic_profile(pic_handle* h, int enable) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  prof_drain(h);
  h->prof = enable != 0;
  if (enable) {
    prof_reserve(h, 1024);
    std::memset(h->ms_sum, 0, sizeof(h->ms_sum));
    std::memset(h->launches, 0, sizeof(h->launches));
  }
  return PIC_OK;
}

int pic_profile_read(pic_handle* h, double* ms_sum, int64_t* launches) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  prof_drain(h);
  for (int i = 0; i < 8; ++i) {
    if (ms_sum) ms_sum[i] = h->ms_sum[i];
    if (launches) launches[i] = h->launches[i];
  }
  return PIC_OK;
}

int pic_bad_count(pic_handle* h, int64_t* count) {
  if (!h || !count) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  unsigned long long c[4] = {0, 0, 0, 0};
  HIPCHK(h, hipMemcpyAsync(c, h->bad, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *count = (int64_t)c[0];
#ifdef PIC_EXP_STAMPB
  if (c[3]) {
    const double nw = (double)(c[3] & 0xffffffffull);
    fprintf(stderr, "[stampB] sweep C, late workgroups, per wave: zero-LDS %.2f us  field-tile load %.2f us  barrier %.2f us  loop %.2f us  straggler wait %.2f us  (waves=%.0f)\n",
            0.01 * (double)(c[1] >> 32) / nw, 0.01 * (double)(c[1] & 0xffffffffull) / nw, 0.01 * (double)(c[2] >> 32) / nw,
            0.01 * (double)(c[2] & 0xffffffffull) / nw, 0.01 * (double)(c[3] >> 32) / nw, nw);
  }
#endif
#ifdef PIC_EXP_STAMP
  if (c[3]) fprintf(stderr, "[stamp] sweep C wave0/block: iterations=%llu  mem-wait %.3f us/iter  push+store-issue %.3f us/iter\n",
                    c[3], 0.01 * (double)c[1] / (double)c[3], 0.01 * (double)c[2] / (double)c[3]);
#endif
  return PIC_OK;
}

}  // extern "C"
