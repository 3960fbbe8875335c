// pic_sweep.h -- the push sweeps: in-prologue field solve, one particle through one Yoshida sub-stage, flush of the
// LDS mesh into the global fixed-point accumulators, and sweep_kernel<P, A, SHAPE, STAGE> itself.
#pragma once
#include "pic_device.h"
#include "pic_solve.h"

namespace {

// Field tile of a sweep workgroup, computed in its own prologue from the accumulator row the previous sweep
// filled: density -> b = n - n0 -> G = dx cumsum(b) - mean -> E_j = -(G_{j+1/2} + G_{j-1/2})/2 (+ E_ext).
// Every workgroup of an environment repeats the solve (2 KB of input at Ng = 256, read through L2); in exchange a
// step has no field-solve launch between its sweeps.  sb: Ng doubles of LDS scratch.
// ctl (pointers of THIS environment): the external field on the mesh, or the actuator coefficients it is built from here
// (xt: Ng doubles of LDS scratch for it) -- a controlled step has no actuator launch either.
// first: the deposit of node threadIdx.x, requested by the caller BEFORE its particle tile (the solve is what the workgroup
// waits for first, and a wave's loads return in the order they were issued); it is waited for here, behind the actuator product
template <typename T, int OFF>
__device__ __forceinline__ void prologue_field(const acc_t* __restrict__ acc_in, const AccRequest& first, int S, long long sub,
                                               const Control& ctl, double* __restrict__ ext_out, int Ng, double unit, double scale,
                                               double n0, double dx, double* __restrict__ sb, double* __restrict__ xt,
                                               double* __restrict__ slot, T* __restrict__ Es) {
  const int tid = threadIdx.x;
  if (ctl.act)
    for (int j = tid; j < Ng; j += BLOCK) {
      const double e = actuator_field(ctl.basis, ctl.basis + (size_t)Ng * ctl.M, ctl.act, j, ctl.M);   // actuator.py:54-63
      xt[j] = e;
      if (ext_out) ext_out[j] = e;
    }
  if (tid < Ng) sb[tid] = ((double)acc_row_finish(first, S) * unit) * scale - n0;           // interpolate.py:16-18, pic.py:116
  for (int j = tid + BLOCK; j < Ng; j += BLOCK)
    sb[j] = ((double)acc_row_sum(acc_in, j, S, sub) * unit) * scale - n0;
  __syncthreads();
  PIC_STAMP(3);
  scan_fields(sb, nullptr, Ng, dx, slot);
  __syncthreads();
  PIC_STAMP(4);
  const double gmean = slot[0];
  for (int i = tid; i < Ng + 2; i += BLOCK) {
    int node = i - OFF;
    node = node < 0 ? node + Ng : (node >= Ng ? node - Ng : node);
    const double gp = sb[node] - gmean;
    const double gm = sb[node == 0 ? Ng - 1 : node - 1] - gmean;
    double E = -0.5 * (gp + gm);
    if (ctl.act) E += xt[node];                                                           // util.py:102-103
    else if (ctl.ext) E += ctl.ext[node];
    Es[i] = (T)E;
  }
  __syncthreads();
  PIC_STAMP(5);
}

// One particle through one sub-stage.  Stages D / REFRESH also deposit the NEXT step's first drift
// position q1 = x' + (c1 p) dt into a second mesh (acc2), which is exactly what sweep A of the next
// step would deposit from the stored x', p -- so that sweep (a full read of x and v) is skipped.
template <typename P, typename A, int SHAPE, int STAGE>
__device__ __forceinline__ void push_one(typename P::X& xq, typename P::V& vp, const typename P::W* __restrict__ Es,
                                         A* __restrict__ acc, A* __restrict__ acc2, const Consts<P>& k, double& ke,
                                         unsigned& bad) {
  using T = typename P::W;
  T w[3];
  typename P::X xw;
  int j;
  unsigned frac;
  typename P::X q = xq;
  typename P::V p = vp;
  if (STAGE == ST_B2) {
    // The post-step deposit of the PREVIOUS step (pic.py:145): q is the x' that step's sweep D2 wrapped and stored, so cell and
    // weights are the ones its own deposit would have had (locate = wrap + locate_in_box, and the wrap of a wrapped position
    // is the position).  A value outside [0, L) cannot come from D2; should memory hold one, its index folds to node 0 below.
    if constexpr (P::kFixed) locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
    else locate_in_box<P, SHAPE>(q, k, j, w, frac);
    deposit<A, P, SHAPE>(acc2, j, w, frac, k.magic);
  }
  if (STAGE == ST_A) {
    q = drift<P>(q, p, k.c_cur, k, bad);                          // integration.py:42, c1
  } else if (STAGE == ST_B || STAGE == ST_B2 || STAGE == ST_C || STAGE == ST_D || STAGE == ST_D2) {
    if (STAGE == ST_B || STAGE == ST_B2) q = drift<P>(q, p, k.c_prev, k, bad);      // q1 again (it is never stored)
    locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
    const T E = gather_field<T, SHAPE>(Es, j, w);                 // util.py:105 / pic.py:120
    p = p + (typename P::V)((k.d_cur * (-E)) * k.dt);             // integration.py:32, pic.py:127
    q = drift<P>(q, p, k.c_cur, k, bad);                          // integration.py:42
  }
  if (STAGE == ST_D2) {                                           // the wrap alone: the deposit of x' is the next sweep B2's
    if constexpr (P::kFixed) xw = q;
    else xw = wrap_periodic(q, k.L, bad);
  } else {
    locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
    deposit<A, P, SHAPE>(acc, j, w, frac, k.magic);
  }
  if (STAGE == ST_D || STAGE == ST_D2 || STAGE == ST_REFRESH) {
    q = xw;                                                       // pic.py:139 (+ util.py:51)
    ke += (double)p * (double)p;
    const typename P::X qn = drift<P>(q, p, k.c_next, k, bad);    // next step's q1 (integration.py:42, c1)
    typename P::X xn;
    locate<P, SHAPE>(qn, k, xn, j, w, frac, bad);
    deposit<A, P, SHAPE>(acc2, j, w, frac, k.magic);
  }
  xq = q;
  vp = p;
}

// Total of one mesh node over the R LDS replicas of a workgroup, periodic ghost slots folded in, as an integer
// in units of 2^-fg (the unit of the global accumulators).
template <typename A, int SHAPE>
__device__ __forceinline__ acc_t mesh_node_sum(const A* __restrict__ acc_all, int R, int stride, int Ng, int fg, int c) {
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  if constexpr (std::is_same<A, fix_t>::value) {
    const unsigned long long* a = reinterpret_cast<const unsigned long long*>(acc_all);
    const int cm = c == 0 ? Ng - 1 : c - 1;
    unsigned long long own = 0ull, left = 0ull;
    for (int r = 0; r < R; ++r) {
      own += a[(size_t)r * stride + c];
      left += a[(size_t)r * stride + cm];
    }
    const long long mask = (1ll << FX_LOW) - 1;
    const long long q = ((long long)(own >> FX_LOW) << FX_FRAC) - ((long long)own & mask) + ((long long)left & mask);
    return q << (fg - FX_FRAC);                                   // 2^-24 units -> 2^-fg units
  } else {
    A s = A(0);
    for (int r = 0; r < R; ++r) {
      const A* ar = acc_all + (size_t)r * stride;
      A t = ar[c + OFF];
      if (SHAPE == PIC_CIC) {
        if (c == 0) t += ar[Ng];
      } else {
        if (c == 0) t += ar[Ng + 1];
        if (c == Ng - 1) t += ar[0];
      }
      s += t;
    }
    if constexpr (std::is_same<A, acc_t>::value) return s;
    else return __double2ll_rn(ldexp((double)s, fg));
  }
}

// Add a workgroup's LDS mesh to the environment's global accumulator row with 64-bit integer atomics (executed
// at the memory side, order-independent; 16.8 MB per sweep at config 2, hidden under the streaming of the other
// resident workgroups: profiles/experiments_r2.md).
template <typename A, int SHAPE>
__device__ __forceinline__ void flush_mesh(const A* __restrict__ acc_all, int R, int stride, int Ng, int fg,
                                           acc_t* __restrict__ row) {
  unsigned long long* out = reinterpret_cast<unsigned long long*>(row);
  for (int c = threadIdx.x; c < Ng; c += BLOCK) {
    const acc_t q = mesh_node_sum<A, SHAPE>(acc_all, R, stride, Ng, fg, c);
    if (q) atomicAdd(out + c, (unsigned long long)q);
  }
}

constexpr size_t kSweepStaticLds = (2 * WAVES + 2) * sizeof(double);     // sweep_kernel's static __shared__ arrays

// what a sweep reads its field from, where its deposits go, which retired accumulator rows it clears
struct SweepIO {
  // accumulator rows are [S][env][Ng] (S sub-rows, SweepArgs::S; pic_device.h: acc_row_sum)
  const acc_t* acc_in;     // deposit the field of this sweep's gather is solved from (gather stages)
  Control ctl;             // external field of the force evaluation: on the mesh or as actuator coefficients (environment 0's pointers)
  // The actuator's field is built ONCE per environment and step where the schedule allows it, not in every workgroup of every
  // sweep: ext_out [env][Ng] (or null) receives, from workgroup 0 of every environment, the field this sweep built from ctl.act
  // (sweep B of a call's first step: it cannot wait for anyone) or -- with ctl.ext and next_act -- the field of the NEXT step's
  // coefficients (sweep D inside a rollout).  The sweeps that follow read it as a mesh field (host: run_stages).
  double* ext_out;
  const double* next_act;  // [env][2M] coefficients of the next step (environment 0), or null
  acc_t* acc_out;          // receives this sweep's deposit (zero on entry; null in sweep D2, which makes none)
  acc_t* acc_out2;         // receives the next step's q1 deposit (D, D2, REFRESH) / the previous step's final positions (B2)
  acc_t* zero0;            // accumulators no kernel reads any more: cleared for a later sweep
  acc_t* zero1;
  double* ke_part;         // [env][nblk] sum of p^2 per workgroup (dual stages)
  unsigned long long* bad; // [1] count of non-finite / unrepresentable positions
  // Sweep C inside a multi-step pic_step call: the post-step refresh of the PREVIOUS step (pic.py:145-146) -- nothing this
  // step reads -- is done by one extra workgroup per environment (blockIdx.x == nblk) instead of a launch of its own, from the
  // deposit this step's sweep B2 made of the positions it read (rounds 2-3: sweep D deposited them and sweep B carried the solve).
  // post.acc == null: no such workgroup.
  SolveIO post;
};

constexpr size_t kSweepInlineOffset = 2 * sizeof(void*) + sizeof(SweepIO) + sizeof(SweepArgs);   // sweep_kernel(x, v, io, a, act_inline)
static_assert(sizeof(SweepIO) % 8 == 0 && sizeof(SweepArgs) % 8 == 0, "arguments lie back to back");

template <typename P, typename A, int SHAPE, int STAGE>
__global__ __launch_bounds__(BLOCK) void sweep_kernel(typename P::X* __restrict__ x, typename P::V* __restrict__ v,
                                                      SweepIO io, SweepArgs a, InlineDoubles act_inline) {
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  constexpr int VEC = P::VEC;
  using T = typename P::W;
  using XV = typename P::XV;
  using VV = typename P::VV;
  constexpr bool kIsB = (STAGE == ST_B || STAGE == ST_B2), kIsD = (STAGE == ST_D || STAGE == ST_D2);
  constexpr bool kGather = (kIsB || STAGE == ST_C || kIsD);
  constexpr bool kStore = (kGather || STAGE == ST_REFRESH);
  constexpr bool kStoreV = kGather;
  constexpr bool kReadV = (STAGE != ST_PROBE);
  constexpr bool kFirst = (STAGE != ST_D2);                                    // deposits into the first LDS mesh (-> acc_out)
  constexpr bool kDual = (kIsD || STAGE == ST_REFRESH || STAGE == ST_B2);      // ... into the second one (-> acc_out2)
  constexpr bool kEnergy = (kIsD || STAGE == ST_REFRESH);                      // sum of p^2 per workgroup

  // LDS: [R meshes: acc][R meshes: acc2 (dual stages)][field tile Es]; the mesh region is the scratch of the
  // prologue solve first
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int Ng = a.Ng;
  const int stride = Ng + 2;
  A* acc_all = reinterpret_cast<A*>(smem_raw);
  A* acc2_all = acc_all + (size_t)a.R * stride;
  T* Es = reinterpret_cast<T*>(smem_raw + (size_t)2 * a.R * stride * sizeof(A));
  __shared__ double red[2 * WAVES];
  __shared__ double slot[2];       // mean of the prologue solve's gradient (16 B: keeps the dynamic LDS base aligned)
  static_assert(sizeof(red) + sizeof(slot) == kSweepStaticLds, "pic_create adds the static LDS to the dynamic part it sizes");
  PIC_STAMP(0);

  if (STAGE == ST_C && blockIdx.x == (unsigned)a.nblk) {      // the extra workgroup of its environment (host: only with io.post.acc)
    // its arguments are read from the kernel-argument segment here, by this workgroup alone: held in scalar registers from the
    // kernel's entry on they cost every other workgroup of the sweep a wave of occupancy (pic_device.h: kernarg_at)
    SolveIO post = kernarg_at<SolveIO>(2 * sizeof(void*) + offsetof(SweepIO, post));   // x, v, io, a
    post.out.fb = Feedback{};                                  // (the feedback law's solves are launches of their own)
    solve_environment(post, blockIdx.y, Ng, a.nblk, a.fg, a.S, a.sub, a.scale, a.n0, a.dx, a.N_over_L, smem_raw, red, slot);
    return;
  }

  const int tid = threadIdx.x;
  // Consecutive sweeps walk memory in opposite directions: what the previous sweep wrote last (still
  // in the 256 MB Infinity Cache) is what this one reads first.
  const int env = a.reverse ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int blk = a.reverse ? a.nblk - 1 - (int)blockIdx.x : (int)blockIdx.x;

  typename P::X* xe = x + (size_t)env * a.ld;
  typename P::V* ve = v + (size_t)env * a.ld;
  const long long step = (long long)BLOCK * VEC;
  // A workgroup owns one contiguous run of `chunk` particles of its environment; a lane streams 16 B of x and
  // 16 B of v per iteration and updates them in place.  The first tile is requested before the prologue, so that
  // its latency runs under the field solve -- but behind the accumulator row the solve starts from.
  const long long begin = (long long)blk * a.chunk;
  long long end = begin + a.chunk;
  if (end > a.N) end = a.N;
  long long i = begin + (long long)tid * VEC;
  AccRequest first_node{};            // the accumulator row goes first: the prologue waits for it, the tile is not needed before the push
  if (kGather) first_node = acc_row_request(io.acc_in + (size_t)env * Ng, tid < Ng ? tid : Ng - 1, a.S, a.sub);   // (no branch: see acc_row_request)
  // (unconditional, from a clamped address where the lane has no first tile: the row is padded to 64 elements.  A load under
  // a branch costs the exact vmcnt bookkeeping, and the prologue would wait for the tile as well as for the row)
  const long long i_first = (i + VEC <= end) ? i : 0;
  XV xv = stream_load(reinterpret_cast<const XV*>(xe + i_first));
  VV vv = {};
  if (kReadV) vv = stream_load(reinterpret_cast<const VV*>(ve + i_first));
  PIC_STAMP(2);

  // LDS of the prologue: b / G in the SECOND mesh's place; the first mesh is cleared while the accumulator row is on its way
  // (with an actuator its field takes that place first).  A sweep that deposits one mesh and has no actuator field to build
  // goes from the field tile's barrier straight to its particles.
  bool clear_first = true;
  if (kGather) {
    Control ctl = io.ctl;
    if (ctl.ext) ctl.ext += (size_t)env * Ng;
    // (a.act_inline: the call's actuator coefficients [num_envs][2M] are the kernel's last argument -- read through a pointer
    // into the argument segment -- instead of an array a copy or a launch in front of the step would have had to fill)
    if (a.act_inline) ctl.act = kernarg_ptr<double>(kSweepInlineOffset);
    if (ctl.act) ctl.act += (size_t)env * 2 * ctl.M;
    else {
      for (int c = tid; c < a.R * stride; c += BLOCK) acc_all[c] = A{};
      clear_first = false;
    }
    prologue_field<T, OFF>(io.acc_in + (size_t)env * Ng, first_node, a.S, a.sub, ctl,
                           (io.ext_out && blk == 0) ? io.ext_out + (size_t)env * Ng : nullptr, Ng, ldexp(1.0, -a.fg), a.scale, a.n0, a.dx,
                           reinterpret_cast<double*>(acc2_all), reinterpret_cast<double*>(smem_raw), slot, Es);
  }
  if (kIsD && io.next_act && io.ext_out && blk == 0) {   // (inside a rollout: one workgroup per environment)
    const Control ctl = io.ctl;
    for (int j = tid; j < Ng; j += BLOCK)
      io.ext_out[(size_t)env * Ng + j] = actuator_field(ctl.basis, ctl.basis + (size_t)Ng * ctl.M, io.next_act + (size_t)env * 2 * ctl.M, j, ctl.M);
  }
  if (clear_first) for (int c = tid; c < a.R * stride; c += BLOCK) acc_all[c] = A{};
  if (kDual) for (int c = tid; c < a.R * stride; c += BLOCK) acc2_all[c] = A{};
  if (blk < a.S) {      // workgroup s of an environment clears sub-row s of the retired accumulator rows
    const size_t z = (size_t)blk * a.sub + (size_t)env * Ng;
    if (io.zero0) for (int c = tid; c < Ng; c += BLOCK) io.zero0[z + c] = 0;
    if (io.zero1) for (int c = tid; c < Ng; c += BLOCK) io.zero1[z + c] = 0;
  }
  if (clear_first || kDual) __syncthreads();
  PIC_STAMP(6);

  const int rep = (tid >> 6) & (a.R - 1);
  A* acc = acc_all + (size_t)rep * stride;
  A* acc2 = acc2_all + (size_t)rep * stride;
  const Consts<P> k(a);

  double ke = 0.0;
  unsigned bad = 0u;
  [[maybe_unused]] int tile = 0;
  const StreamOut xout(xe + begin), vout(ve + begin);
  while (i + VEC <= end) {
    PIC_STAMP_LOADS(8 + 2 * tile);
    typename P::X* xs = reinterpret_cast<typename P::X*>(&xv);
    typename P::V* vs = reinterpret_cast<typename P::V*>(&vv);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      typename P::V pv = kReadV ? vs[c] : typename P::V(0);
      push_one<P, A, SHAPE, STAGE>(xs[c], pv, Es, acc, acc2, k, ke, bad);
      if (kReadV) vs[c] = pv;
    }
    if (kStore) {
      xout.store((int)((i - begin) * (long long)sizeof(typename P::X)), xv);
      if (kStoreV) vout.store((int)((i - begin) * (long long)sizeof(typename P::V)), vv);
    }
    PIC_STAMP(9 + 2 * tile);
    ++tile;
    i += step;
    if (i + VEC <= end) {
      xv = stream_load(reinterpret_cast<const XV*>(xe + i));
      if (kReadV) vv = stream_load(reinterpret_cast<const VV*>(ve + i));
    }
  }
  for (long long c = i; c < end; ++c) {       // ragged tail (fewer than VEC particles left for this lane)
    typename P::X xq = xe[c];
    typename P::V pv = kReadV ? ve[c] : typename P::V(0);
    push_one<P, A, SHAPE, STAGE>(xq, pv, Es, acc, acc2, k, ke, bad);
    if (kStore) {
      xe[c] = xq;
      if (kStoreV) ve[c] = pv;
    }
  }
  __syncthreads();
  PIC_STAMP(24);

  const size_t sub_row = (size_t)(blk % a.S) * a.sub + (size_t)env * Ng;
  if (kFirst) flush_mesh<A, SHAPE>(acc_all, a.R, stride, Ng, a.fg, io.acc_out + sub_row);
  if (kDual) flush_mesh<A, SHAPE>(acc2_all, a.R, stride, Ng, a.fg, io.acc_out2 + sub_row);
  PIC_STAMP(25);

  if (kEnergy) {
    double w = wave_sum(ke);
    if ((tid & 63) == 0) red[tid >> 6] = w;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int c = 0; c < WAVES; ++c) s += red[c];
      io.ke_part[(size_t)env * a.nblk + blk] = s;
    }
  }
  if (bad) atomicAdd(io.bad, (unsigned long long)bad);
  PIC_STAMP_LOADS(26);
}

}  // namespace
