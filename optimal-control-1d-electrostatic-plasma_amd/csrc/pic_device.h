// pic_device.h -- launch constants, kernel argument blocks and the per-particle device helpers of picstep.hip:
// particle formats, periodic wrap, division by dx, cell location and shape weights, LDS gather / deposit,
// fixed-point conversion, wave and block scans.  Included by picstep.hip only (one translation unit; everything
// lives in its anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "picstep.h"

namespace {

// Timeline hooks.  The product build defines nothing here and no stamp executes in libpicstep.so; the diagnostic build
// profiles/timeline/picstep_timeline.hip defines PIC_STAMP(slot) (shader clock of thread 0 of the workgroup into a
// buffer of its own) and PIC_STAMP_LOADS(slot) (the same after the wave's outstanding vector-memory operations).
#ifndef PIC_STAMP
#define PIC_STAMP(slot) ((void)0)
#define PIC_STAMP_LOADS(slot) ((void)0)
#endif

constexpr int kMaxFeedbackModes = 16;
constexpr int kInlineDoubles = 32;                 // doubles a call's actuator coefficients may number to travel inside the argument block
struct InlineDoubles { double v[kInlineDoubles]; };   // modes of the on-device feedback law (pic_step_feedback)

// A kernel argument read from the kernel-argument segment at the point of use.  By-value arguments are loaded into scalar
// registers at the kernel's entry and held until their last use; the rarely used ones (a riding solve's outputs, the control
// inputs of a step) then cost the particle loops scalar registers -- a wave of occupancy in the sweeps, spills in the
// resident kernel.  offset: byte offset of the value in the argument list (arguments lie in declaration order, each at its
// natural alignment, like the members of a struct).
template <typename T>
__device__ __forceinline__ T kernarg_at(size_t offset) {
#if defined(__HIP_DEVICE_COMPILE__)
  using KP = const __attribute__((address_space(4))) char*;
  KP ka = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));                       // opaque: the loads below stay where they are written
  T out;
  __builtin_memcpy(&out, (const __attribute__((address_space(4))) T*)(ka + offset), sizeof(T));
  return out;
#else
  (void)offset;
  return T{};                                        // (host pass of the single-source compilation: never executed)
#endif
}

// A generic pointer to a member of the argument list: small per-call inputs (one action of a Gym-style loop) travel inside the
// argument block of the kernel that uses them instead of a copy command or a launch of their own in front of it (5-6 us each
// on the stream of a 22 us step: profiles/gym_breakdown.py).
template <typename T>
__device__ __forceinline__ const T* kernarg_ptr(size_t offset) {
#if defined(__HIP_DEVICE_COMPILE__)
  using KP = const __attribute__((address_space(4))) char*;
  KP ka = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  return (const T*)(const char*)(ka + offset);
#else
  (void)offset;
  return nullptr;
#endif
}

// Touch every 64-byte line of the first BYTES of the kernel-argument segment at the kernel's entry, together with the loads the
// entry makes anyway.  The argument block of a launch is new to the scalar cache; read group by group where they are used
// (kernarg_at), its lines miss one after the other, each on the critical path of the phase that reads it: the resident kernel,
// whose step is a chain of short phases, gains 0.45 us per launch from missing them all at once (the sweeps, which read nearly
// everything at entry anyway, lose 0.1-0.2 us to the longer first wait and do not do it: profiles/experiments_r3.md 12).
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if defined(__HIP_DEVICE_COMPILE__)
  using KP = const __attribute__((address_space(4))) unsigned*;
  KP ka = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned w = 0;
#pragma unroll
  for (int line = 1; line < (BYTES + 63) / 64; ++line) w |= ka[line * 16];
  asm volatile("" ::"s"(w));                         // the loads are waited for HERE (one wait for all of them), not sunk to a later block
#endif
}

constexpr int BLOCK = 512;          // sweep workgroup: 8 waves of 64 (512 beat 256 by 2.7 % and 128 by 11 % at config 2)
constexpr int WAVES = BLOCK / 64;

enum Stage : int {
  ST_A = 0,        // drift(c) from x,v ; deposit ; nothing stored
  ST_B = 1,        // recompute q1 = x + (c_prev v) dt ; gather ; kick ; drift ; deposit ; store
  ST_C = 2,        // gather ; kick ; drift ; deposit ; store
  ST_D = 3,        // as C, then wrap, KE ; store wrapped x ; deposit the next step's q1 as well
  ST_REFRESH = 4,  // wrap x ; deposit ; KE ; store wrapped x ; deposit the next step's q1   (pic.py:93-112 on reset)
  ST_PROBE = 5,    // deposit positions of a scratch array, nothing stored             (util.py:73-116 callers)
  // Inside a multi-step call (round 4): the post-step deposit of the final positions x' -- density, E_mesh, phi, PE of the state after
  // a step, which nothing of the next step's dynamics reads -- moves from sweep D, the one sweep bound by VALU issue, into the next
  // step's sweep B, which is bound by memory and reads x' anyway.  Same positions, same cells and weights, same integer sums.
  ST_B2 = 6,       // as B, and deposits its INPUT positions (the previous step's x') into the second mesh
  ST_D2 = 7        // as D without the deposit of x' (wrap, KE, store, the next step's q1 deposit)
};

// Mesh accumulators that cross a kernel boundary: per environment and node the sum of shape weights as a 64-bit
// integer in units of 2^-fg.  Integer sums do not depend on the order of the adds, so the result of a sweep is
// the same whatever the launch geometry and however the adds of different workgroups interleave.
using acc_t = long long;

// An accumulator row is kept as S sub-rows [S][env][Ng] whose integer sum is the deposit.  With one sub-row every workgroup of
// an environment adds to the same 8 Ng bytes, and the memory-side atomic units serialise them: with ONE environment of 1e6
// particles (245 workgroups) the flush took 2-5 us of a 13 us sweep and held up the kernel boundary behind it
// (profiles/r3_timeline.md).  Few-environment handles therefore spread a row over several sub-rows; integer sums make the
// result independent of S bit for bit.
__device__ __forceinline__ acc_t acc_row_sum(const acc_t* __restrict__ row, int j, int S, long long sub) {
  const acc_t* p = row + j;
  acc_t t = 0;
  int s = 0;
  for (; s + 4 <= S; s += 4) {                       // four loads in flight at a time
    const acc_t r0 = p[0]; p += sub;
    const acc_t r1 = p[0]; p += sub;
    const acc_t r2 = p[0]; p += sub;
    const acc_t r3 = p[0]; p += sub;
    t += (r0 + r1) + (r2 + r3);
  }
  for (; s < S; ++s) { t += p[0]; p += sub; }
  return t;
}

struct SweepArgs {
  long long N;        // particles per env
  long long ld;       // leading dimension of x, v
  long long chunk;    // particles per workgroup (multiple of BLOCK * VEC)
  int Ng;
  int nblk;           // workgroups per env
  int R;              // LDS mesh replicas per workgroup (a power of two; the host uses 1: profiles/experiments_r2.md 17)
  int reverse;        // walk environments and chunks from the far end (alternates sweep to sweep)
  int fg;             // fractional bits of the fixed-point accumulators
  int S;              // sub-rows of an accumulator row: workgroup b adds its mesh into sub-row b % S, readers sum the S sub-rows
  int act_inline;     // sweeps: the actuator coefficients of the call are the kernel's last argument (SweepIO::ctl.act is a stand-in)
  long long sub;      // elements from one sub-row to the next (num_envs * Ng)
  double magic;       // 1.5 * 2^(52 - fg): (w + magic) holds round(w 2^fg) in its low mantissa bits
  double L, dx, rdx, dt;   // rdx = 1/dx (float particles: 1/(float)dx)
  double c_prev, c_cur, d_cur, c_next;
  double scale, n0;   // density scale n0 L / N / dx and mean density, for the in-prologue field solve
  double to_units;    // 2^32 / L: fixed-point position units per length (PosU32)
  double N_over_L;    // PE = PE_reward N / L (util.py:130)
};

struct SolveArgs {
  long long N;
  long long sub;       // elements between the sub-rows of an accumulator row
  int Ng;
  int nblk;
  int fg;
  int S;               // sub-rows of the accumulator row read (1 for a plain row)
  double L, dx, n0;
  double scale;        // n0 * L / N / dx, evaluated left to right as interpolate.py:18
  double N_over_L;
};

// The same in two halves for callers that have other loads to issue in between (S <= 4, which is what pic_create chooses): the
// request leaves the sub-row values unsummed, so that nothing waits for them before acc_row_finish.
struct AccRequest { acc_t r[4]; };
__device__ __forceinline__ AccRequest acc_row_request(const acc_t* __restrict__ row, int j, int S, long long sub) {
  AccRequest q;                        // four unconditional loads, untouched until acc_row_finish: a sub-row that does not exist
#pragma unroll                         // re-reads sub-row 0 (loads under a branch, or a select right behind them, would make the
  for (int s = 0; s < 4; ++s) q.r[s] = row[(size_t)(s < S ? s : 0) * sub + j];       // compiler wait here)
  return q;
}
__device__ __forceinline__ acc_t acc_row_finish(const AccRequest& q, int S) {
  return (q.r[0] + (S > 1 ? q.r[1] : 0)) + ((S > 2 ? q.r[2] : 0) + (S > 3 ? q.r[3] : 0));
}

// How the external field of a step's force evaluations is obtained (util.py:102-103 adds it to E_mesh): given on the mesh,
// or built from actuator coefficients inside the field phase, E_ext = basis_cos @ a[:M] + basis_sin @ a[M:]
// (src/control/actuator.py:54-63) -- no actuator launch and no mesh-sized array per step.  Pointers are those of environment 0
// (and of the first step of the call).
struct Control {
  const double* ext;     // [env][Ng], or null
  const double* act;     // [env][2M] coefficients (cos half, then sin half), or null; wins over ext
  const double* basis;   // [2][Ng][M] cos | sin tables of the host mirror (they carry the reference's linspace(0, L, Ng) mesh)
  int M;
};

// E_field.compute_E at mesh node j (actuator.py:54-63): both products summed over m in order, then added -- the operation
// order every caller of the device actuator has had since round 1 (results are compared bit for bit across schedules).
__device__ __forceinline__ double actuator_field(const double* __restrict__ bc, const double* __restrict__ bs,
                                                 const double* __restrict__ a, int j, int M) {
  double c = 0.0, s = 0.0;
  for (int m = 0; m < M; ++m) c += bc[(size_t)j * M + m] * a[m];
  for (int m = 0; m < M; ++m) s += bs[(size_t)j * M + m] * a[M + m];
  return c + s;
}

// ---------------------------------------------------------------------------------------------
// Particle formats.  X / V: storage types of position and velocity; W: type locate, the shape weights,
// the gather and the kick are evaluated in; XV / VV: the 16-byte vectors a lane streams.
//   PosF64  the parity format: everything float64, the reference's arithmetic operand by operand
//   PosF32  float32 positions and velocities
//   PosU32  positions as 32-bit fixed point x = u L / 2^32 (wrap = integer overflow, cell = high word of
//           u Ng, weight = low word), float32 velocities: uniform 1.2e-8 resolution on a 50-long box instead
//           of float32's 3.8e-6 near x = L (profiles/fp32_error_model.md)
// ---------------------------------------------------------------------------------------------
typedef double pic_v2d __attribute__((ext_vector_type(2)));
typedef float pic_v4f __attribute__((ext_vector_type(4)));
typedef unsigned pic_v4u __attribute__((ext_vector_type(4)));

struct PosF64 { using X = double;   using V = double; using W = double; using XV = pic_v2d; using VV = pic_v2d; static constexpr int VEC = 2; static constexpr bool kFixed = false; };
struct PosF32 { using X = float;    using V = float;  using W = float;  using XV = pic_v4f; using VV = pic_v4f; static constexpr int VEC = 4; static constexpr bool kFixed = false; };
struct PosU32 { using X = unsigned; using V = float;  using W = float;  using XV = pic_v4u; using VV = pic_v4f; static constexpr int VEC = 4; static constexpr bool kFixed = true; };

// The 16-byte accesses of the particle streams.
// Stores are WRITE-THROUGH (`sc1`): the bytes go to the memory side at once and the line is dropped from the XCD's L2.  A plain
// store leaves its line dirty in L2 until the end of the kernel, and the write-back of what a sweep has left there sits between
// it and the next one on the stream: with ONE environment of 1e6 particles (16 MB written per sweep, all of it still in the
// 32 MB of L2 at the end) that was 5-7 us per kernel boundary, 47 -> 31 us per step; 2 / 4 / 12 environments 50.8 -> 45 /
// 77.6 -> 74 / 194 -> 190.5, config 2 989 -> 974 (profiles/experiments_r3.md 2; `nt` stores keep the line and gain nothing).
// Nothing re-reads a particle in the sweep that stored it, and the next sweep runs on other CUs anyway: no reuse is given up.
// HIP has no 16-byte store with a scope; the buffer-store builtin takes the cache policy as an operand (16 = sc1) and, unlike
// inline assembly, leaves hazards and vmcnt to the compiler (an `asm` store was tried first: the compiler then scheduled a
// VALU write of the store's data registers right behind it, which gfx950 does not interlock -- tests/ caught it).
template <typename VT>
__device__ __forceinline__ VT stream_load(const VT* p) { return *p; }

// base: wave-uniform (a workgroup's chunk of one array, < 2 GiB: pic_create sees to it); off: this lane's byte offset from it
struct StreamOut {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ explicit StreamOut(void* base) : rsrc(__builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000)) {}
  template <typename VT>
  __device__ __forceinline__ void store(int off, VT v) const {
    static_assert(sizeof(VT) == 16, "particle tiles are 16 bytes per lane");
    typedef unsigned pic_v4raw __attribute__((ext_vector_type(4)));
    pic_v4raw w;
    __builtin_memcpy(&w, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, off, 0, 16);
  }
};

// loop-invariant scalars of a sweep in the format's arithmetic type
template <typename P>
struct Consts {
  typename P::W L, dx, rdx, dt, c_prev, c_cur, d_cur, c_next;
  float to_units;
  double magic;
  int Ng;
  __device__ explicit Consts(const SweepArgs& a)
      : L((typename P::W)a.L), dx((typename P::W)a.dx), rdx((typename P::W)a.rdx), dt((typename P::W)a.dt),
        c_prev((typename P::W)a.c_prev), c_cur((typename P::W)a.c_cur), d_cur((typename P::W)a.d_cur),
        c_next((typename P::W)a.c_next), to_units((float)a.to_units), magic(a.magic), Ng(a.Ng) {}
  __device__ Consts(double L_, double dx_, int Ng_)
      : L((typename P::W)L_), dx((typename P::W)dx_), rdx((typename P::W)1 / (typename P::W)dx_), dt(0), c_prev(0),
        c_cur(0), d_cur(0), c_next(0), to_units((float)(4294967296.0 / L_)), magic(0), Ng(Ng_) {}
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
__device__ __forceinline__ T wrap_periodic(T q, T L, unsigned& bad) {
  // the three near ranges as selects (a particle moves a small fraction of L per sub-stage)
  T up = q + L;                       // q in [-L, 0)
  up = (up >= L) ? T(0) : up;         // tiny negative q: q + L rounds to L, the second mod gives 0
  T r = (q < T(0)) ? up : q;
  r = (q >= L) ? q - L : r;           // q in [L, 2L): exact (Sterbenz)
  // For q in [-L, 2L) r now lies in [0, L).  One range test therefore catches both a position further away
  // (fmod path) and a NaN / inf one (counted, parked on node 0, never used as an index).
  if (__builtin_expect(!(r >= T(0) && r < L), 0)) {
    r = wrap_periodic_far(q, L);
    if (!(r >= T(0) && r < L)) {
      bad += 1u;
      r = T(0);
    }
  }
  return r;
}

// a / dx for the loop-invariant divisor dx, with rdx = 1/dx rounded once on the host: one Newton
// correction on the reciprocal product, q0 = a rdx; q = q0 + (a - q0 dx) rdx, both steps fused.
// The value before the final rounding is within ~2^-104 relative of a/dx, so the result is the
// IEEE quotient unless a/dx lies that close to a rounding boundary (probability ~2^-52 per
// operation, then 1 ulp off) -- 3 instructions instead of the ~12 of the full v_div_* sequence,
// which made sweep D division-bound.  tests/ check it bit for bit against true division.
template <typename T>
__device__ __forceinline__ T div_dx(T a, T dx, T rdx) {
  T q0 = a * rdx;
  T rem = fma(-q0, dx, a);
  return fma(rem, rdx, q0);
}

// The part of `locate` behind the wrap: cell and weights of a position already in [0, L).  Sweep B2 uses it on the positions
// sweep D2 stored (wrapped there): the same xw, hence the same cell and weights as the deposit D would have made.
template <typename P, int SHAPE>
__device__ __forceinline__ void locate_in_box(typename P::X xw, const Consts<P>& k, int& j, typename P::W (&w)[3], unsigned& frac) {
  using T = typename P::W;
  static_assert(!P::kFixed, "fixed-point positions need no wrap: locate() is the whole of it");
  frac = 0u;
  T jf = floor(div_dx(xw, k.dx, k.rdx));
  j = (int)jf;
  // j == Ng happens when xw/dx rounds up to Ng (undefined in the reference: bincount grows a bin and
  // solve.py:32 raises); it is folded to node 0 with the weights of the unfolded index.
  if ((unsigned)j >= (unsigned)k.Ng) j = 0;
  if (SHAPE == PIC_CIC) {
    w[0] = div_dx((jf + T(1)) * k.dx - xw, k.dx, k.rdx);
    w[1] = div_dx(xw - jf * k.dx, k.dx, k.rdx);
    w[2] = T(0);
  } else {
    T d = div_dx(xw - jf * k.dx, k.dx, k.rdx);
    T a = T(1.5) - d, b = d - T(1), c = d - T(0.5);
    w[0] = T(0.5) * (a * a);
    w[1] = T(0.75) - b * b;
    w[2] = T(0.5) * (c * c);
  }
}

// Cell index and shape-function weights at position q.  j is the LDS index of the leftmost
// touched node (mesh node + OFF, OFF = 1 for TSC so that node -1 has a slot).
//   CIC (interpolate.py:6-13): jl = floor(xw/dx); wl = ((jl+1) dx - xw)/dx; wr = (xw - jl dx)/dx
//   TSC (interpolate.py:24-34): d = (xw - jm dx)/dx; wl = .5(1.5-d)^2; wm = .75-(d-1)^2; wr = .5(d-.5)^2
// xw: the wrapped position (what sweep D stores).  frac: PosU32 only, the position inside the cell in
// units of 2^-32 cell (the right CIC weight, exactly).
template <typename P, int SHAPE>
__device__ __forceinline__ void locate(typename P::X q, const Consts<P>& k, typename P::X& xw, int& j,
                                       typename P::W (&w)[3], unsigned& frac, unsigned& bad) {
  using T = typename P::W;
  if constexpr (P::kFixed) {
    xw = q;                                            // every 32-bit pattern is a position in [0, L)
    j = (int)__umulhi(q, (unsigned)k.Ng);              // floor(u Ng / 2^32)
    frac = q * (unsigned)k.Ng;                         // (u Ng) mod 2^32
    const T d = (T)frac * T(2.3283064365386963e-10);   // 2^-32
    if (SHAPE == PIC_CIC) {
      w[1] = d;
      w[0] = T(1) - d;
      w[2] = T(0);
    } else {
      T a = T(1.5) - d, b = d - T(1), c = d - T(0.5);
      w[0] = T(0.5) * (a * a);
      w[1] = T(0.75) - b * b;
      w[2] = T(0.5) * (c * c);
    }
  } else {
    xw = wrap_periodic(q, k.L, bad);
    locate_in_box<P, SHAPE>(xw, k, j, w, frac);
  }
}

// position formats <-> length units
template <typename P>
__device__ __forceinline__ double pos_to_length(typename P::X q, double L) {
  if constexpr (P::kFixed) return (double)q * (L * 2.3283064365386963e-10);     // u L / 2^32
  else return (double)q;
}

template <typename P>
__device__ __forceinline__ typename P::X pos_from_length(double xs, double L, unsigned& bad) {
  if constexpr (P::kFixed) {
    double r = xs - floor(xs / L) * L;                 // np.mod for any finite xs
    if (!(r >= 0.0 && r < L)) { if (!(r == L)) bad += 1u; r = 0.0; }
    return (unsigned)((unsigned long long)rint(r / L * 4294967296.0) & 0xFFFFFFFFull);
  } else {
    return (typename P::X)xs;
  }
}

// q + (c p) dt  (integration.py:42).  PosU32: the displacement is rounded to position units and added modulo
// 2^32, which is the periodic wrap; a displacement of half a box or more per sub-stage cannot be represented
// (and is far outside any CFL-limited step): counted as a bad position.
template <typename P>
__device__ __forceinline__ typename P::X drift(typename P::X q, typename P::V p, typename P::W c, const Consts<P>& k,
                                               unsigned& bad) {
  using T = typename P::W;
  if constexpr (P::kFixed) {
    const float d = ((c * (T)p) * k.dt) * k.to_units;
    if (!(fabsf(d) < 2147483648.0f)) bad += 1u;
    return q + (unsigned)__float2int_rn(d);
  } else {
    return q + (c * (T)p) * k.dt;
  }
}

template <typename T, int SHAPE>
__device__ __forceinline__ T gather_field(const T* __restrict__ Es, int j, const T (&w)[3]) {
  T e = w[0] * Es[j] + w[1] * Es[j + 1];
  if (SHAPE == PIC_TSC) e = e + w[2] * Es[j + 2];
  return e;
}

// ---------------------------------------------------------------------------------------------
// LDS accumulators (template parameter A of the sweeps)
//   acc_t   round(w 2^fg) per weight, integer ds_add_u64: the default.  Sums are exact integers, hence
//           independent of the order in which the waves' atomics land: a step is bitwise reproducible.
//           Rounding a weight to 2^-fg (fg = 38..50 by particle count) is below what one float64 add of
//           the running sum rounds away.
//   double  ds_add_f64 (accum_dtype PIC_ACC_F64): float64 running sums, order-dependent in the last bits.
//   fix_t   packed (count, sum of w_r) per cell in one word (accum_dtype PIC_FIXED, single-precision CIC):
//           a particle in cell j adds w_l = 1 - w_r to node j and w_r to node j+1, so per cell the pair
//           carries the whole deposit, n_j = count_j - S_j + S_{j-1}: ONE ds_add_u64 per particle instead of
//           two.  count in the top 20 bits, S in 2^-24 units below; a workgroup handles fewer than 2^20
//           particles (pic_create sees to it), so neither field overflows.
// ---------------------------------------------------------------------------------------------
struct fix_t { unsigned long long v; };
constexpr int FX_FRAC = 24;
constexpr int FX_LOW = 44;

__device__ __forceinline__ acc_t to_fixed(double w, double magic) {
  // w + magic has ulp 2^-fg, so its mantissa differs from magic's by round-to-nearest-even(w 2^fg)
  return __double_as_longlong(w + magic) - __double_as_longlong(magic);
}

template <typename A, typename P, int SHAPE>
__device__ __forceinline__ void deposit(A* __restrict__ acc, int j, const typename P::W (&w)[3], unsigned frac,
                                        double magic) {
  if constexpr (std::is_same<A, fix_t>::value) {
    static_assert(SHAPE == PIC_CIC, "the packed accumulator is CIC only");
    unsigned long long lo;
    if constexpr (P::kFixed) {
      lo = (frac >> 8) + ((frac >> 7) & 1u);                        // w_r rounded to 2^-24
    } else {
      const float wr = fminf(fmaxf((float)w[1], 0.0f), 1.0f);
      lo = (unsigned long long)(unsigned)(wr * (float)(1u << FX_FRAC) + 0.5f);
    }
    atomicAdd(reinterpret_cast<unsigned long long*>(acc) + j, (1ull << FX_LOW) + lo);
  } else if constexpr (std::is_same<A, acc_t>::value) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(acc);
    atomicAdd(a + j, (unsigned long long)to_fixed((double)w[0], magic));
    atomicAdd(a + j + 1, (unsigned long long)to_fixed((double)w[1], magic));
    if (SHAPE == PIC_TSC) atomicAdd(a + j + 2, (unsigned long long)to_fixed((double)w[2], magic));
  } else {
    atomicAdd(&acc[j], (A)w[0]);
    atomicAdd(&acc[j + 1], (A)w[1]);
    if (SHAPE == PIC_TSC) atomicAdd(&acc[j + 2], (A)w[2]);
  }
}

// Wave-level sums on the DPP path of the VALU (no LDS round trip: a ds_bpermute shuffle costs ~150 cycles per step of
// a dependent chain, a DPP move ~10).  dpp_add<CTRL>(v) = v + (v of the lane the control selects, +0.0 where there is
// none or the row is masked off).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
  return v + __hiloint2double(hi, lo);
}

// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16 lanes (row_shr:1,2,4,8), then the row
// totals travel on (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  All 64 lanes must be active.
__device__ __forceinline__ double wave_incl_scan(double v) {
  v = dpp_add<0x111>(v);
  v = dpp_add<0x112>(v);
  v = dpp_add<0x114>(v);
  v = dpp_add<0x118>(v);
  v = dpp_add<0x142, 0xA>(v);
  v = dpp_add<0x143, 0xC>(v);
  return v;
}

// sum over the 64 lanes, the same value in every lane
__device__ __forceinline__ double wave_sum(double v) {
  v = wave_incl_scan(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
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

// sums of two values over a workgroup of NW waves at the cost of one (ws: 2 NW doubles of LDS)
template <int NW>
__device__ __forceinline__ void block_sum2(double a, double b, double* ws, double& sa, double& sb) {
  const double wa = wave_sum(a), wb = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    ws[threadIdx.x >> 6] = wa;
    ws[NW + (threadIdx.x >> 6)] = wb;
  }
  __syncthreads();
  sa = 0.0;
  sb = 0.0;
  for (int i = 0; i < NW; ++i) {
    sa += ws[i];
    sb += ws[NW + i];
  }
  __syncthreads();
}

// Fourier mode m of a mesh row E [Ng] by a workgroup of NW waves: fft(E)[m] / Ng * 2 (src/interpret/spectrum.py:16).
// twc / tws: cos and sin of 2 pi m j / Ng, j = 0..Ng-1 (twiddle_kernel, pic_aux.h).  Thread t takes the nodes t + k 64 NW; E may
// be global memory the same threads wrote a moment ago (each reads back its own nodes only).  ws: 2 NW doubles of LDS.
template <int NW>
__device__ __forceinline__ void mesh_mode(const double* __restrict__ E, const double* __restrict__ twc,
                                          const double* __restrict__ tws, int Ng, double* __restrict__ ws, double& re,
                                          double& im) {
  double sr = 0.0, si = 0.0;
  int t0 = threadIdx.x;
  asm volatile("" : "+v"(t0));      // (keeps the per-lane addresses below from being hoisted out of a caller's step loop into registers held across it)
  for (int j = t0; j < Ng; j += NW * 64) {
    const double e = E[j];
    sr += e * twc[j];
    si -= e * tws[j];
  }
  double a, b;
  block_sum2<NW>(sr, si, ws, a, b);
  re = a / Ng * 2.0;
  im = b / Ng * 2.0;
}

// Linear feedback law of run_feedback.py:133-135 (and the behaviour-cloning action of src/control/rl/ddpg.py:369-371): the
// actuator coefficients of the next step are (-Re, +Im) of modes 1..M of the mesh field the last step left.
struct Feedback {
  const double* tw;      // [2][rows][Ng] cos | sin twiddles, row m-1 = mode m
  int rows;
  int M;                 // 0 = no feedback
  double* act_out;       // [env][2M] the action computed (read by the next force evaluations), or null
  double* act_hist;      // [env][2M] of the step the action is for, in a per-step record, or null
};

// a_lds (2M doubles of LDS, or null) also receives the action; the caller puts a barrier before reading it
template <int NW>
__device__ __forceinline__ void feedback_action(const double* __restrict__ E, const Feedback& fb, int env, int Ng,
                                                double* __restrict__ ws, double* __restrict__ a_lds) {
  for (int m = 0; m < fb.M; ++m) {
    double re, im;
    mesh_mode<NW>(E, fb.tw + (size_t)m * Ng, fb.tw + ((size_t)fb.rows + m) * Ng, Ng, ws, re, im);
    if (threadIdx.x == 0) {
      const double c = -re, s = im;
      if (a_lds) { a_lds[m] = c; a_lds[fb.M + m] = s; }
      if (fb.act_out) { fb.act_out[(size_t)env * 2 * fb.M + m] = c; fb.act_out[(size_t)env * 2 * fb.M + fb.M + m] = s; }
      if (fb.act_hist) { fb.act_hist[(size_t)env * 2 * fb.M + m] = c; fb.act_hist[(size_t)env * 2 * fb.M + fb.M + m] = s; }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Periodic Poisson solve of one environment, in LDS.
// Replaces Gaussian_Elimination_Periodic + the dense grad matvec (src/env/solve.py:27-53, src/env/util.py:99-103,
// pic.py:116-117).  With G_{j+1/2} = (phi_{j+1}-phi_j)/dx the 3-point periodic Poisson equation reads
// G_{j+1/2} - G_{j-1/2} = b_j dx, so G = cumsum(b) dx - mean and E_j = -(phi_{j+1}-phi_{j-1})/(2dx)
// = -(G_{j+1/2} + G_{j-1/2})/2; phi_{j+1} = phi_j + dx G_{j+1/2} is a second scan.
//
// scan_fields: ONE wave (number `wave` of the workgroup) does the scans with shuffles alone (lane l owns the
// m = ceil(Ng / 64) consecutive nodes from l m) while the other waves of the workgroup go on to the caller's barrier: no barrier inside, and the same
// rounding whatever the size of the calling workgroup (sweep prologue, resident kernel, field_solve_kernel).
// In: sb[0..Ng) = b = n - n0.  Out: sb = G_{j+1/2} (mean NOT removed), slot[0] = mean(G); with sp != null also
// sp[j] = phi_j (before its mean is removed) and slot[1] = mean(phi).  The caller puts a barrier between this call
// and any use of sb, sp or slot, and does not reuse slot before its next barrier after that.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void scan_fields(double* __restrict__ sb, double* __restrict__ sp, int Ng, double dx,
                                            double* __restrict__ slot, int wave = 0) {
  if ((int)(threadIdx.x >> 6) != wave) return;
  const int lane = threadIdx.x & 63;
  const int m = (Ng + 63) / 64;
  const int lo = min(lane * m, Ng), hi = min(lo + m, Ng);
  double loc = 0.0;
  for (int j = lo; j < hi; ++j) loc += sb[j];
  double run = wave_incl_scan(loc) - loc;            // exclusive prefix over the lanes
  loc = 0.0;
  for (int j = lo; j < hi; ++j) {
    run += sb[j];
    const double g = run * dx;
    sb[j] = g;
    loc += g;
  }
  const double gmean = wave_sum(loc) / (double)Ng;
  if (lane == 0) slot[0] = gmean;
  if (sp) {
    loc = 0.0;
    for (int j = lo; j < hi; ++j) loc += (sb[j] - gmean) * dx;
    run = wave_incl_scan(loc) - loc;
    double ploc = 0.0;
    for (int j = lo; j < hi; ++j) {
      sp[j] = run;
      ploc += run;
      run += (sb[j] - gmean) * dx;
    }
    ploc = wave_sum(ploc);
    if (lane == 0) slot[1] = ploc / (double)Ng;
  }
}

// outputs of a field solve (any pointer may be null) and what goes into them besides the density
struct SolveOut {
  const double* ext;       // E_ext [env][Ng] added to E (force evaluations, util.py:102-103), or null
  double *E, *phi;         // [env][Ng]
  double *KE, *PE, *PEr;   // [env]
  double* hist;            // [3][num_envs] KE, PE, PE_reward of this solve once more (one step's entry of an energy history), or null
  int num_envs;
  Feedback fb;             // fb.M > 0: the feedback action from the E just solved (needs E != null)
};

// From sb = b = n - n0 (filled by the caller, barrier included) to E (+ E_ext), zero-mean phi, PE = 0.5 sum(E^2) dx N/L
// (util.py:129-130), PE_reward = 0.5 sum(E^2) dx (objective.py:33) and KE = 0.5 * (workgroup sum of `ke_share`)
// (util.py:144), by a workgroup of NW waves.  se: Ng doubles, ws: 2 NW doubles, slot: 2 doubles of LDS.
template <int NW>
__device__ __forceinline__ void solve_block(const SolveOut& o, int env, int Ng, double dx, double N_over_L, double ke_share,
                                            double* __restrict__ sb, double* __restrict__ se, double* __restrict__ ws,
                                            double* __restrict__ slot) {
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x;
  const size_t row = (size_t)env * Ng;
  scan_fields(sb, o.phi ? se : nullptr, Ng, dx, slot);
  __syncthreads();
  const double gmean = slot[0];
  const double pmean = o.phi ? slot[1] : 0.0;
  double e2 = 0.0;
  for (int j = tid; j < Ng; j += NT) {
    const double gp = sb[j] - gmean;
    const double gm = sb[j == 0 ? Ng - 1 : j - 1] - gmean;
    const double E = -0.5 * (gp + gm);
    const double Et = o.ext ? E + o.ext[row + j] : E;
    if (o.E) o.E[row + j] = Et;
    e2 += Et * Et;
    if (o.phi) o.phi[row + j] = se[j] - pmean;
  }
  double S, K;
  block_sum2<NW>(e2, ke_share, ws, S, K);
  if (tid == 0) {
    const double pe = 0.5 * S * dx;
    if (o.PEr) o.PEr[env] = pe;
    if (o.PE) o.PE[env] = pe * N_over_L;
    if (o.KE) o.KE[env] = 0.5 * K;
    if (o.hist) {
      o.hist[env] = 0.5 * K;
      o.hist[o.num_envs + env] = pe * N_over_L;
      o.hist[2 * (size_t)o.num_envs + env] = pe;
    }
  }
  if (o.fb.M > 0) feedback_action<NW>(o.E + row, o.fb, env, Ng, ws, nullptr);   // (block_sum2 ended with a barrier: ws is free)
}

}  // namespace
