// pic_resident.h -- resident_kernel<P, A, SHAPE, PPT, NW>: whole environment steps inside ONE workgroup.
//
// The reference's own workload is many small environments (N = 5000, Ng = 250: run_wo_oc.py:33-34, the RL
// trainers).  There a step of the streaming schedule is four dependent launches of a few microseconds each and
// most of that is latency (accumulator round trips, kernel boundaries).  When an environment's particles fit
// the registers of one workgroup (N <= 64 NW PPT, up to 8192), this kernel keeps them there for all `nsteps` steps
// of a pic_step call: the meshes live in LDS, the field solves are workgroup scans, nothing crosses a kernel
// boundary and nothing but the final particles, fields and per-step energies goes to HBM.
//
// Arithmetic is the streaming path's, helper by helper (locate, gather_field, drift, deposit, mesh_node_sum,
// scan_fields; where registers allow, the gather re-uses the cell and weights the previous deposit located instead of
// locating the same position again), and the deposits are the same integer sums, so particles and fields come out bit for bit as
// from the sweeps (tests/test_gpu_resident.py).  Only KE, a float64 sum whose order follows the launch geometry,
// may differ in its last bits.
#pragma once
#include "pic_device.h"
#include "pic_sweep.h"

namespace {

constexpr size_t kResidentStaticLds = (4 * 8 + 4 + 2 * kMaxFeedbackModes) * sizeof(double);   // resident_kernel's static __shared__ arrays (NW = 8)

// The kernel's argument block, in groups that are read from the kernel-argument segment WHERE THEY ARE USED (pic_device.h:
// kernarg_at).  Passed by value and left to the compiler the whole block is loaded into scalar registers at entry and held to
// its last use -- through every particle phase: the first cut of round 3 paid 400 more v_readlane per step than round 2 and
// 2.5-3 us of every 15 us step for it (profiles/experiments_r3.md 4).
struct ResidentCtl {       // what drives the external field: read at the top of a step, and only in the modes that change it
  Control ctl;             // external field of the force evaluations (environment 0, first step of the call), or nulls
  long long ext_step;      // elements from one step's ctl.ext to the next step's (0: the same field for all steps of the call)
  long long act_step;      // the same for ctl.act (a new action every step: PIC.simulate's E_external_traj, ddpg.py:421-468)
  Feedback fb;             // fb.M > 0: the action of every step is the feedback law's, computed from the field the step before
                           // left (step 0: from E as it stands in memory); ctl.basis / ctl.M describe the actuator, ctl.act is unused
};
struct ResidentOut {       // where a post-step refresh writes: read inside the field phase that carries it
  double *n, *E, *phi;     // [env][Ng] (pic.py:145-146)
  double *KE, *PE, *PEr;   // [env]
  double* hist;            // [nsteps][3][env] KE, PE, PE_reward after every step, or null
};
struct ResidentEdge {      // entry and exit of a launch
  // The LDS mesh holding the deposit of the NEXT step's first drift position crosses from one launch to the next as it
  // stands (R (Ng + 2) words per environment): a call that follows another call neither deposits q1 again nor waits for its
  // particles before the first field phase.
  const unsigned long long* q1_in;    // [env][R (Ng + 2)] left by the previous launch, or null: deposit from the particles
  unsigned long long* q1_out;         // [env][R (Ng + 2)]
  // ... and, for a handful of environments, so do the cell and weights every particle's q1 was located with (kernels that carry
  // them, up to 10 particles per lane): [env] blocks of 64 NW PPT cells (int) followed by 2 (TSC: 3) arrays of as many weights.
  // Without them the first particle phase behind a mesh taken over locates every q1 again (1.8 us of a 22 us launch at N = 5000).
  const void* carry_in;               // valid together with q1_in, or null
  void* carry_out;                    // or null
  unsigned long long* bad;
  void* snap;              // [nsteps][2][env][N] positions (as floats of the particle dtype) and velocities after every step, or null
};
enum : int { RM_EXT = 1,       // the force evaluations have an external field (rc.xt)
             RM_PER_STEP = 2,  // ... a new one every step (ext_step / act_step)
             RM_FEEDBACK = 4,  // ... computed by the feedback law
             RM_SNAP = 8,      // particle snapshots are recorded
             RM_ACT_INLINE = 32, // the (held) actuator coefficients of all environments are in ResidentIO::act_inline, not behind ctl.act
             RM_RECORD = 16 }; // something reads the fields and energies BETWEEN the steps of the call (an energy history, the feedback
                               // law): every step's post-step refresh is made.  Otherwise only the last step's, which is all that
                               // can be seen afterwards (pic.py:145-146 overwrites them step by step)
struct ResidentIO {
  int nsteps;
  int num_envs;
  int mode;                // RM_* bits: all the particle phases and the plain step loop need to know of the groups below
  double c1, c2, d1, d2;   // Yoshida-4 (integration.py:62-69): c = (c1, c2, c2, c1), d = (0, d1, d2, d1)
  ResidentCtl c;
  ResidentOut o;
  ResidentEdge e;
};
constexpr size_t kResidentIoOffset = 2 * sizeof(void*);      // resident_kernel(x, v, io, a, act_inline): io's byte offset in the argument list
// ... and that of the last argument, [num_envs][2M] actuator coefficients with RM_ACT_INLINE (read through a pointer into the
// argument segment, never as a value; behind everything else, so that a launch without it touches none of its lines)
constexpr size_t kResidentInlineOffset = kResidentIoOffset + sizeof(ResidentIO) + sizeof(SweepArgs);
static_assert(sizeof(ResidentIO) % 8 == 0 && sizeof(SweepArgs) % 8 == 0 && alignof(InlineDoubles) == 8, "arguments lie back to back");
#define RESIDENT_ARG(group, member) kernarg_at<group>(kResidentIoOffset + offsetof(ResidentIO, member))

// What the post-step refresh of a step needs besides the meshes (pic.py:145-146): where its results go, and the LDS
// scratch of the second scanning wave.
struct RefreshCtx {
  double* s2;              // Ng doubles: b = n - n0 of the refreshed density, then its G
  double* se;              // Ng doubles: phi
  double* slot2;           // 2 doubles
  double* ws;              // 2 NW doubles
  double* wsf;             // 2 NW doubles more for the feedback law's mode sums
  double* a_lds;           // 2 M doubles: the feedback action
  double* xt;              // Ng doubles: the external field of this environment in LDS
  size_t row;              // env * Ng
  int env;
  double N_over_L;
};

// Field tile Es (gather layout: Ng + 2 slots, OFF for TSC) from the LDS mesh `acc_all`; sb: Ng doubles of scratch;
// has_ext: rc.xt holds the external field of this environment.
// While the scanning wave works, the others clear the meshes the coming particle phase deposits into (`z0`, and
// `z1` = the mesh just read, in sub-stage D), so that the phase needs no barrier of its own for that.
//
// kRefresh (sub-stage B of every step but the first of a launch): the post-step refresh of the PREVIOUS step -- density
// from the mesh `z0` still holds (sub-stage D's deposit of the final positions), E, phi and the three energies -- rides
// along: its sums go with the force field's sums, its two scans run in wave 1 next to wave 0's, its outputs with the
// field tile; it costs no barrier of its own.  Nothing in the step that follows reads what it produces.
template <typename T, typename A, int SHAPE, int NW, bool kRefresh>
__device__ __forceinline__ void resident_field(const A* __restrict__ acc_all, int R, int stride, bool has_ext,
                                               int Ng, int fg, double scale, double n0, double dx, double* __restrict__ sb,
                                               double* __restrict__ slot, T* __restrict__ Es, A* __restrict__ z0,
                                               A* __restrict__ z1, int mode, int num_envs, const RefreshCtx& rc,
                                               double ke_prev, int prev_step) {
  constexpr int NT = NW * 64;
  constexpr int OFF = (SHAPE == PIC_TSC) ? 1 : 0;
  const int tid = threadIdx.x;
  const double* xt = has_ext ? rc.xt : nullptr;      // (the feedback law below rewrites it: no __restrict__)
  const double unit = ldexp(1.0, -fg);
  for (int j = tid; j < Ng; j += NT)
    sb[j] = ((double)mesh_node_sum<A, SHAPE>(acc_all, R, stride, Ng, fg, j) * unit) * scale - n0;
  ResidentOut out{};
  if (kRefresh) {
    out = RESIDENT_ARG(ResidentOut, o);
    for (int j = tid; j < Ng; j += NT) {
      const double nj = ((double)mesh_node_sum<A, SHAPE>(z0, R, stride, Ng, fg, j) * unit) * scale;
      out.n[rc.row + j] = nj;
      rc.s2[j] = nj - n0;
    }
  }
  __syncthreads();                                   // the meshes have been read: they may be cleared now
  scan_fields(sb, nullptr, Ng, dx, slot, 0);
  if (kRefresh) scan_fields(rc.s2, rc.se, Ng, dx, rc.slot2, 1);
  for (int i = tid; i < R * stride; i += NT) z0[i] = A{};
  if (z1) for (int i = tid; i < R * stride; i += NT) z1[i] = A{};
  __syncthreads();
  // The refresh's outputs (solve_block's, reduction order included).  They follow the field tile -- the particle phase waits
  // for that one -- unless the feedback law is on: then the field just refreshed decides the external field of the tile.
  double e2 = 0.0;
  auto refresh_outputs = [&]() {
    const double g2 = rc.slot2[0], pmean = rc.slot2[1];
    for (int j = tid; j < Ng; j += NT) {
      const double gp = rc.s2[j] - g2;
      const double gm = rc.s2[j == 0 ? Ng - 1 : j - 1] - g2;
      const double E = -0.5 * (gp + gm);
      out.E[rc.row + j] = E;
      e2 += E * E;
      out.phi[rc.row + j] = rc.se[j] - pmean;
    }
  };
  const bool fb_on = kRefresh && (mode & RM_FEEDBACK);
  if (fb_on) {                                       // the step about to start is driven by the field just refreshed
    refresh_outputs();
    const ResidentCtl c = RESIDENT_ARG(ResidentCtl, c);
    Feedback fb = c.fb;
    if (fb.act_hist) fb.act_hist += (size_t)(prev_step + 1) * num_envs * 2 * fb.M;
    feedback_action<NW>(out.E + rc.row, fb, rc.env, Ng, rc.wsf, rc.a_lds);
    __syncthreads();
    int t0 = tid;
    asm volatile("" : "+v"(t0));
    for (int j = t0; j < Ng; j += NT)
      rc.xt[j] = actuator_field(c.ctl.basis, c.ctl.basis + (size_t)Ng * c.ctl.M, rc.a_lds, j, c.ctl.M);
    __syncthreads();
  }
  const double gmean = slot[0];
  for (int i = tid; i < Ng + 2; i += NT) {
    int node = i - OFF;
    node = node < 0 ? node + Ng : (node >= Ng ? node - Ng : node);
    const double gp = sb[node] - gmean;
    const double gm = sb[node == 0 ? Ng - 1 : node - 1] - gmean;
    double E = -0.5 * (gp + gm);
    if (xt) E += xt[node];
    Es[i] = (T)E;
  }
  if (kRefresh) {
    if (!fb_on) refresh_outputs();
    const double wa = wave_sum(e2), wb = wave_sum(ke_prev);
    if ((tid & 63) == 0) {
      rc.ws[tid >> 6] = wa;
      rc.ws[NW + (tid >> 6)] = wb;
    }
  }
  __syncthreads();
  if (kRefresh && tid == 0) {                        // ws is not written again before the next field phase's barriers
    double S = 0.0, K = 0.0;
    for (int i = 0; i < NW; ++i) {
      S += rc.ws[i];
      K += rc.ws[NW + i];
    }
    const double pe = 0.5 * S * dx;
    out.PEr[rc.env] = pe;
    out.PE[rc.env] = pe * rc.N_over_L;
    out.KE[rc.env] = 0.5 * K;
    if (out.hist) {
      double* h3 = out.hist + (size_t)prev_step * 3 * num_envs;
      h3[rc.env] = 0.5 * K;
      h3[num_envs + rc.env] = pe * rc.N_over_L;
      h3[2 * (size_t)num_envs + rc.env] = pe;
    }
  }
}

template <typename P, typename A, int SHAPE, int PPT, int NW, bool kCarry>
__global__ __launch_bounds__(NW * 64) void resident_kernel(typename P::X* __restrict__ x, typename P::V* __restrict__ v,
                                                           ResidentIO io, SweepArgs a, InlineDoubles act_inline) {
  constexpr int NT = NW * 64;
  using T = typename P::W;
  using X = typename P::X;
  using V = typename P::V;
  // Of the argument block only these scalars live in registers; the groups io.c / io.o / io.e are read where they are used.
  const int nsteps = io.nsteps, num_envs = io.num_envs, mode = io.mode;

  // LDS: [R meshes: accA][R meshes: accB][sb][se][s2][xt: Ng doubles each][field tile Es]
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int Ng = a.Ng;
  const int stride = Ng + 2;
  const int R = a.R;
  A* accA = reinterpret_cast<A*>(smem_raw);
  A* accB = accA + (size_t)R * stride;
  double* sb = reinterpret_cast<double*>(accB + (size_t)R * stride);
  double* se = sb + Ng;
  double* s2 = se + Ng;
  double* xt_lds = s2 + Ng;
  T* Es = reinterpret_cast<T*>(xt_lds + Ng);
  __shared__ double ws[2 * NW], wsf[2 * NW];
  __shared__ double slot[2], slot2[2];
  __shared__ double a_lds[2 * kMaxFeedbackModes];
  static_assert(sizeof(ws) + sizeof(wsf) + sizeof(slot) + sizeof(slot2) + sizeof(a_lds) == kResidentStaticLds,
                "pic_create adds the static LDS to the dynamic part it sizes");

  PIC_STAMP(0);
  kernarg_warm<kResidentIoOffset + sizeof(ResidentIO) + sizeof(SweepArgs)>();
  const int tid = threadIdx.x;
  const int env = blockIdx.x;
  const int rep = (tid >> 6) & (R - 1);
  const Consts<P> k(a);
  const T c1 = (T)io.c1, c2 = (T)io.c2, d1 = (T)io.d1, d2 = (T)io.d2;
  X* xe = x + (size_t)env * a.ld;
  V* ve = v + (size_t)env * a.ld;
  const size_t row = (size_t)env * Ng;
  const int words = R * stride;                      // one LDS mesh, as it crosses launches
  static_assert(sizeof(A) == sizeof(unsigned long long), "LDS mesh entries are 8 bytes in every accumulator format");
  unsigned long long* accA_w = reinterpret_cast<unsigned long long*>(accA);
  RefreshCtx rc{s2, se, slot2, ws, wsf, a_lds, xt_lds, row, env, a.N_over_L};
  const bool has_ext = mode & RM_EXT;
  const unsigned long long* q1_in = io.e.q1_in;      // (by value here: used at entry only)
  const bool took_over = q1_in != nullptr;           // the q1 mesh comes from the previous launch
  constexpr int NWT = SHAPE == PIC_TSC ? 3 : 2;
  constexpr bool kHandCarry = kCarry && PPT <= 10 && sizeof(T) == 8;   // (16 particles per lane leave no registers for it, and the
                                                                        // float32 kernels spill with it: they locate q1 again)
  constexpr size_t kCarryBlock = (size_t)NT * PPT * (sizeof(int) + (NWT == 2 ? 2 : 4) * sizeof(T));   // bytes per environment: cell + weights (TSC: padded to 4)
  const char* carry_in = kHandCarry && took_over ? static_cast<const char*>(io.e.carry_in) : nullptr;

  // The q1 mesh of the previous launch is requested first and the particles after it: the first field phase needs the mesh
  // only, so that the particles' latency runs under it.
  constexpr int QW = 4;                              // words per thread: R (Ng + 2) <= 4 NT (pic_create sees to it)
  unsigned long long qw[QW];
  if (took_over) {
#pragma unroll
    for (int c = 0; c < QW; ++c) {
      const int i = tid + c * NT;
      qw[c] = i < words ? q1_in[(size_t)env * words + i] : 0ull;
    }
  }
  // A lane holds PPT / 2 pairs of neighbouring particles: slots 2g and 2g + 1 are particles 2 (g NT + tid) and the one after it,
  // so that the state enters and leaves in 16-byte (float64) or 8-byte accesses -- half as many memory instructions as one
  // particle per access (issuing them was 1.6 us at either end of a launch).  Rows are padded to an even length with zeros
  // (ld), so the partner of the last particle of an odd N is loaded, never pushed, and stored back as it came.
  static_assert(PPT % 2 == 0 && PPT <= 32, "particle slots come in pairs; one validity bit each");
  auto slot_index = [&](int s) { return 2 * ((long long)(s >> 1) * NT + tid) + (s & 1); };
  typedef X XPair __attribute__((ext_vector_type(2)));
  typedef V VPair __attribute__((ext_vector_type(2)));
  X xs[PPT];
  V vs[PPT];
  unsigned live = 0u;                                // bit s: slot s holds a particle (index < N)
#pragma unroll
  for (int g = 0; g < PPT / 2; ++g) {
    const long long i = slot_index(2 * g);
    XPair xp = {X(0), X(0)};
    VPair vp = {V(0), V(0)};
    if (i < a.N) {
      xp = *reinterpret_cast<const XPair*>(xe + i);
      vp = *reinterpret_cast<const VPair*>(ve + i);
      live |= (i + 1 < a.N ? 3u : 1u) << (2 * g);
    }
    xs[2 * g] = xp.x; xs[2 * g + 1] = xp.y;
    vs[2 * g] = vp.x; vs[2 * g + 1] = vp.y;
  }
  unsigned bad = 0u;
  PIC_STAMP(2);

  // Where a sub-stage deposits a particle is where the next one gathers its field.  kCarry keeps cell and weights of
  // that position in registers from one sub-stage to the next (one locate per sub-stage instead of two, -15 % per
  // step); without it the gather locates again and the kernel needs ~50 registers fewer, so that two workgroups
  // share a CU -- the better trade once there are more environments than CUs (host: launch_resident).
  constexpr int NCAR = kCarry ? PPT : 1;
  int js[NCAR] = {};
  T wgt[NCAR][SHAPE == PIC_TSC ? 3 : 2] = {};

  // the external field of a call that holds it for all its steps: one copy in LDS, made here
  if (has_ext && !(mode & (RM_PER_STEP | RM_FEEDBACK))) {
    const Control ctl = io.c.ctl;                    // (by value: dead after this block)
    const double* act = (mode & RM_ACT_INLINE) ? kernarg_ptr<double>(kResidentInlineOffset) : ctl.act;
    if (act)
      for (int j = tid; j < Ng; j += NT)
        xt_lds[j] = actuator_field(ctl.basis, ctl.basis + (size_t)Ng * ctl.M, act + (size_t)env * 2 * ctl.M, j, ctl.M);
    else
      for (int j = tid; j < Ng; j += NT) xt_lds[j] = ctl.ext[row + j];
  }
  for (int i = tid; i < words; i += NT) accB[i] = A{};
  if (took_over) {
#pragma unroll
    for (int c = 0; c < QW; ++c) {
      const int i = tid + c * NT;
      if (i < words) accA_w[i] = qw[c];
    }
  } else {
    // deposit of the first drift position q1 = x + (c1 v) dt (integration.py:42 with d1 = 0) into accA
    for (int i = tid; i < words; i += NT) accA[i] = A{};
    __syncthreads();
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
      if (kCarry) js[kCarry ? s : 0] = 0;
      if (live & (1u << s)) {
        T w[3];
        X xw;
        int j;
        unsigned frac;
        const X q = drift<P>(xs[s], vs[s], c1, k, bad);
        locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
        deposit<A, P, SHAPE>(accA + (size_t)rep * stride, j, w, frac, k.magic);
        if (kCarry) {
          js[kCarry ? s : 0] = j;
          wgt[kCarry ? s : 0][0] = w[0]; wgt[kCarry ? s : 0][1] = w[1];
          if (SHAPE == PIC_TSC) wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0] = w[2];
        }
      }
    }
  }
  __syncthreads();
  PIC_STAMP(3);
  if (kHandCarry && carry_in) {
    // Requested HERE, behind the entry's barrier: the first field phase (no memory traffic of its own) runs over their latency,
    // and the wait for the q1 mesh above does not have to count them (requested with the particles they held the entry up by
    // 1.2 us: the compiler drains every load in flight there).  Every slot: the buffer holds 64 NW PPT entries whatever N is.
    const int* cj = reinterpret_cast<const int*>(carry_in + (size_t)env * kCarryBlock);
    typedef T TW __attribute__((ext_vector_type(NWT == 2 ? 2 : 4)));
    const TW* cw = reinterpret_cast<const TW*>(cj + NT * PPT);
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
      const int e = s * NT + tid;
      js[kCarry ? s : 0] = cj[e];
      const TW w = cw[e];
      wgt[kCarry ? s : 0][0] = w.x; wgt[kCarry ? s : 0][1] = w.y;
      if (SHAPE == PIC_TSC) wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0] = w[2];
    }
  }

  double ke = 0.0;               // sum of p^2 of this lane's particles after the step just made
  for (int step = 0; step < nsteps; ++step) {
    const double ke_prev = ke;
    ke = 0.0;
    if (step > 0 && step == nsteps - 1) PIC_STAMP(20);
    // The external field of this step's three force evaluations where it changes from step to step, one copy in LDS.  Written
    // here, read in the field phases behind their barriers; the previous step's last read lies before the barrier that ended
    // its particle phase.  (The feedback law's field of a later step is made inside sub-stage B's field phase.)
    if (mode & RM_FEEDBACK) {
      if (step == 0) {           // from the field as it stands in memory
        const ResidentCtl c = RESIDENT_ARG(ResidentCtl, c);
        const ResidentOut out = RESIDENT_ARG(ResidentOut, o);
        feedback_action<NW>(out.E + row, c.fb, env, Ng, wsf, a_lds);
        __syncthreads();
        for (int j = tid; j < Ng; j += NT) xt_lds[j] = actuator_field(c.ctl.basis, c.ctl.basis + (size_t)Ng * c.ctl.M, a_lds, j, c.ctl.M);
      }
    } else if (mode & RM_PER_STEP) {
      const ResidentCtl c = RESIDENT_ARG(ResidentCtl, c);
      if (c.ctl.act)
        for (int j = tid; j < Ng; j += NT)
          xt_lds[j] = actuator_field(c.ctl.basis, c.ctl.basis + (size_t)Ng * c.ctl.M,
                                     c.ctl.act + (size_t)step * c.act_step + (size_t)env * 2 * c.ctl.M, j, c.ctl.M);
      else
        for (int j = tid; j < Ng; j += NT) xt_lds[j] = c.ctl.ext[(size_t)step * c.ext_step + row + j];
    }
    for (int st = ST_B; st <= ST_D; ++st) {
      // sub-stage st reads the field of the deposit in `in` and deposits into `out` (D: also the next q1 into `in`)
      A* in = (st == ST_C) ? accB : accA;
      A* out = (st == ST_C) ? accA : accB;
      // (the previous step's post-step refresh goes with sub-stage B's field phase; the last step's follows the loop)
      if (st == ST_B && step > 0 && (mode & RM_RECORD))
        resident_field<T, A, SHAPE, NW, true>(in, R, stride, has_ext, Ng, a.fg, a.scale, a.n0, a.dx, sb, slot, Es, out, nullptr,
                                              mode, num_envs, rc, ke_prev, step - 1);
      else
        resident_field<T, A, SHAPE, NW, false>(in, R, stride, has_ext, Ng, a.fg, a.scale, a.n0, a.dx, sb, slot, Es, out,
                                               st == ST_D ? in : nullptr, mode, num_envs, rc, 0.0, 0);
      if (step == 0) PIC_STAMP(8 + 2 * (st - ST_B));
      else if (step == nsteps - 1) PIC_STAMP(14 + 2 * (st - ST_B));
      // Yoshida coefficients of this sub-stage (integration.py:62-69): (c, d) = (c2, d1), (c3, d2), (c4, d3)
      const T c_cur = (st == ST_D) ? c1 : c2;
      const T d_cur = (st == ST_C) ? d2 : d1;
      A* acc = out + (size_t)rep * stride;
      A* acc2 = in + (size_t)rep * stride;
      // Behind a mesh taken over from the last launch nothing has located q1 for this launch's registers yet: one pass of
      // locates before the first particle phase.  (In the float32 kernels with 16 particles per lane or three TSC weights each that
      // pass costs the register allocator its footing -- 0.1-1.8 KB of scratch per lane, 114 instead of 16 us per step at N = 8000 --
      // and the locate goes into the phase's own loop there: tests/test_host_cpu.py holds every launchable kernel to zero scratch.)
      constexpr bool kRelocateInLoop = kCarry && sizeof(T) == 4 && (SHAPE == PIC_TSC || (PPT == 16 && !P::kFixed));
      const bool relocate = kRelocateInLoop && took_over && step == 0 && st == ST_B;
      if (kCarry && !kRelocateInLoop && took_over && !carry_in && step == 0 && st == ST_B) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
          js[kCarry ? s : 0] = 0;
          if (live & (1u << s)) {
            T w[3];
            X xw;
            int j;
            unsigned frac;
            locate<P, SHAPE>(drift<P>(xs[s], vs[s], c1, k, bad), k, xw, j, w, frac, bad);
            js[kCarry ? s : 0] = j;
            wgt[kCarry ? s : 0][0] = w[0]; wgt[kCarry ? s : 0][1] = w[1];
            if (SHAPE == PIC_TSC) wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0] = w[2];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < PPT; ++s) {
        if (live & (1u << s)) {
          T w[3];
          X xw;
          int j;
          unsigned frac;
          X q = xs[s];
          V p = vs[s];
          if (st == ST_B) q = drift<P>(q, p, c1, k, bad);         // q1 again (it is never stored)
          if (kCarry && !relocate) {                              // cell and weights of q, located by the previous deposit
            j = js[kCarry ? s : 0];
            w[0] = wgt[kCarry ? s : 0][0]; w[1] = wgt[kCarry ? s : 0][1];
            w[2] = (SHAPE == PIC_TSC) ? wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0] : T(0);
          } else {
            locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
          }
          const T E = gather_field<T, SHAPE>(Es, j, w);           // util.py:105 / pic.py:120
          p = p + (V)((d_cur * (-E)) * k.dt);                     // integration.py:32, pic.py:127
          q = drift<P>(q, p, c_cur, k, bad);                      // integration.py:42
          locate<P, SHAPE>(q, k, xw, j, w, frac, bad);
          deposit<A, P, SHAPE>(acc, j, w, frac, k.magic);
          if (st == ST_D) {
            q = xw;                                               // pic.py:139 (+ util.py:51)
            ke += (double)p * (double)p;
            const X qn = drift<P>(q, p, c1, k, bad);              // next step's q1
            X xn;
            locate<P, SHAPE>(qn, k, xn, j, w, frac, bad);
            deposit<A, P, SHAPE>(acc2, j, w, frac, k.magic);
          }
          if (kCarry) {
            js[kCarry ? s : 0] = j;
            wgt[kCarry ? s : 0][0] = w[0]; wgt[kCarry ? s : 0][1] = w[1];
            if (SHAPE == PIC_TSC) wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0] = w[2];
          }
          xs[s] = q;
          vs[s] = p;
        }
      }
      __syncthreads();
      if (step == 0) PIC_STAMP(9 + 2 * (st - ST_B));
      else if (step == nsteps - 1) PIC_STAMP(15 + 2 * (st - ST_B));
    }

    if (mode & RM_SNAP) {             // PIC.simulate's particle snapshots (pic.py:175-223), written from the registers
      using F = typename P::V;        // positions leave as floats of the particle dtype whatever their format
      const ResidentEdge e = RESIDENT_ARG(ResidentEdge, e);
      F* sx = static_cast<F*>(e.snap) + ((size_t)step * 2 * num_envs + env) * (size_t)a.N;
      F* sv = sx + (size_t)num_envs * (size_t)a.N;
      // two neighbouring particles per store where the record's row is aligned for it (always with an even N): whole 16- or
      // 8-byte pieces per lane -- the record may lie in pinned host memory (pic_step_observe), where half-filled lines cost
      typedef F FPair __attribute__((ext_vector_type(2)));
      const bool paired = (((size_t)sx | (size_t)sv) & (sizeof(FPair) - 1)) == 0;
      auto to_length = [&](X q) {
        F xf = (F)pos_to_length<P>(q, a.L);
        if (P::kFixed && xf >= (F)a.L) xf = F(0);
        return xf;
      };
#pragma unroll
      for (int g = 0; g < PPT / 2; ++g) {
        const long long i = slot_index(2 * g);
        const unsigned both = (live >> (2 * g)) & 3u;
        if (both == 3u && paired) {
          *reinterpret_cast<FPair*>(sx + i) = FPair{to_length(xs[2 * g]), to_length(xs[2 * g + 1])};
          *reinterpret_cast<FPair*>(sv + i) = FPair{vs[2 * g], vs[2 * g + 1]};
        } else {
          if (both & 1u) { sx[i] = to_length(xs[2 * g]); sv[i] = vs[2 * g]; }
          if (both & 2u) { sx[i + 1] = to_length(xs[2 * g + 1]); sv[i + 1] = vs[2 * g + 1]; }
        }
      }
    }
  }
  PIC_STAMP(24);

  // The particles and the mesh of the next step's q1 leave first: their stores run under the refresh that follows.
  const ResidentEdge edge = RESIDENT_ARG(ResidentEdge, e);
#pragma unroll
  for (int g = 0; g < PPT / 2; ++g) {
    if (live & (1u << (2 * g))) {
      const long long i = slot_index(2 * g);
      *reinterpret_cast<XPair*>(xe + i) = XPair{xs[2 * g], xs[2 * g + 1]};
      *reinterpret_cast<VPair*>(ve + i) = VPair{vs[2 * g], vs[2 * g + 1]};
    }
  }
  if (edge.q1_out)
    for (int i = tid; i < words; i += NT) edge.q1_out[(size_t)env * words + i] = accA_w[i];
  if (kHandCarry && edge.carry_out && nsteps > 0) {  // (js / wgt: sub-stage D's last locate = the next step's q1; every slot)
    int* cj = reinterpret_cast<int*>(static_cast<char*>(edge.carry_out) + (size_t)env * kCarryBlock);
    typedef T TW __attribute__((ext_vector_type(NWT == 2 ? 2 : 4)));
    TW* cw = reinterpret_cast<TW*>(cj + NT * PPT);
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
      const int e = s * NT + tid;
      cj[e] = js[kCarry ? s : 0];
      TW w = {};
      w.x = wgt[kCarry ? s : 0][0]; w.y = wgt[kCarry ? s : 0][1];
      if (SHAPE == PIC_TSC) w[2] = wgt[kCarry ? s : 0][SHAPE == PIC_TSC ? 2 : 0];
      cw[e] = w;
    }
  }
  if (bad) atomicAdd(edge.bad, (unsigned long long)bad);
  PIC_STAMP(25);

  // post-step refresh of the last step (pic.py:145-146; no external field: pic.py:114-117) from the deposit in accB
  if (nsteps > 0) {
    const ResidentOut out = RESIDENT_ARG(ResidentOut, o);
    const double unit = ldexp(1.0, -a.fg);
    for (int j = tid; j < Ng; j += NT) {
      const double nj = ((double)mesh_node_sum<A, SHAPE>(accB, R, stride, Ng, a.fg, j) * unit) * a.scale;
      out.n[row + j] = nj;
      sb[j] = nj - a.n0;
    }
    __syncthreads();
    SolveOut o{};
    o.E = out.E; o.phi = out.phi; o.KE = out.KE; o.PE = out.PE; o.PEr = out.PEr;
    solve_block<NW>(o, env, Ng, a.dx, a.N_over_L, ke, sb, se, ws, slot);
    if (out.hist && tid == 0) {       // this thread wrote the three energies a moment ago
      double* h3 = out.hist + (size_t)(nsteps - 1) * 3 * num_envs;
      h3[env] = out.KE[env];
      h3[num_envs + env] = out.PE[env];
      h3[2 * (size_t)num_envs + env] = out.PEr[env];
    }
  }
  PIC_STAMP_LOADS(26);
}

#undef RESIDENT_ARG

}  // namespace
