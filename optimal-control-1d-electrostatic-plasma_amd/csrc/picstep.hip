// picstep.hip -- MI355X (gfx950 / CDNA4) 1-D electrostatic PIC stepper behind the C ABI of
// include/picstep.h.  Written for wave64, LDS-resident per-block mesh tiles and coalesced SoA
// particle streams; there is no other backend and no CPU fallback.
//
// One environment step = PIC.update_state of the reference (src/env/pic.py:131-146), i.e. the
// Yoshida-4 composition of src/env/integration.py:60-75 restated as kick/drift sub-stages:
//
//   [sweep A: q1 = x + (c1 v) dt ; deposit(q1)]   -- normally NOT run: the previous sweep D (or the
//                                                    reset sweep) has already deposited this q1
//   sweep B : E = field(deposit of q1) + E_ext ; p1 = v + (d1 (-E(q1))) dt ; q2 = q1 + (c2 p1) dt ; deposit(q2)
//   sweep C : E = field(deposit of q2) + E_ext ; p2 = p1 + (d2 (-E(q2))) dt ; q3 = q2 + (c3 p2) dt ; deposit(q3)
//   sweep D : E = field(deposit of q3) + E_ext ; p3 = p2 + (d3 (-E(q3))) dt ; q4 = q3 + (c4 p3) dt ;
//             x' = mod(q4, L) ; deposit(x') ; KE partials ; store x', p3 ;
//             deposit(next q1 = x' + (c1 p3) dt) into a second mesh
//   solve   : n, E_mesh (no E_ext), phi, KE, PE, PE_reward        (pic.py:145-146, util.py:119-147)
//
// 4 launches and 3 read+write passes over the particles per step (96 B per particle-step in fp64).  Inside a multi-step
// pic_step call every step but the last ends with sweep D2 = D without deposit(x'); the next step's sweep B2 = B + deposit of the
// positions it reads (that x') makes it instead -- D is the one sweep bound by VALU issue, B is bound by memory -- and the post-step
// solve of the step rides with that step's sweep C (one extra workgroup per environment): 3 launches per step there.  Every
// sweep workgroup solves the field it gathers from in its own prologue (pic_sweep.h: prologue_field), from
// the accumulator row the previous sweep filled.
//
// Arithmetic inside a sub-stage keeps the reference's operand order and is compiled with
// -ffp-contract=off so that fp64 results track NumPy to rounding (tests/ hold the bounds).
//
// Deposit: every workgroup owns an LDS copy of its environment's mesh (two in sweep D: final positions and the next step's
// first drift) and accumulates with integer LDS atomics (weights as 2^-fg fixed point, or one packed word per particle for
// single-precision CIC; pic_device.h), then adds its partial mesh to the environment's row [env][Ng] of a
// global 64-bit fixed-point accumulator with memory-side integer atomics.  Integer sums are order-independent:
// a step is bitwise reproducible and does not depend on the launch geometry.  Accumulator rows rotate through a
// small ring; a row whose readers are done is cleared by a later sweep (no memset launches).
//
// Files: pic_device.h (particle formats, per-particle helpers, scans), pic_sweep.h (push sweeps), pic_solve.h
// (field solve), pic_aux.h (kernels off the step path); this file holds the handle, the launch schedule and
// the C ABI.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "picstep.h"

#include "pic_device.h"
#include "pic_sweep.h"
#include "pic_solve.h"
#include "pic_resident.h"
#include "pic_aux.h"


// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
enum Format : int { FMT_F64 = 0, FMT_F32 = 1, FMT_U32 = 2 };     // PosF64 / PosF32 / PosU32
typedef pic_placement PlacementStats;
struct PlacementState {
  size_t pbytes = 0;                  // size of x and of v (0: no search on this handle)
  int legs = 0;                       // legs run so far
  size_t frontier = 0;                // blocks the longest leg has walked over: a later leg takes no readings before that
  double best_n = 0.0, worst_n = 0.0; // normalised readings of the block kept and of the slowest pair seen (0: none yet)
  bool found = false;
  bool ptrs_exposed = false;          // pic_device_ptrs has handed out the addresses of x and v: v stays where it is
  bool x_cleared = false;
};
constexpr int RING = 8;                                          // accumulator rows in rotation

struct pic_handle {
  pic_config cfg{};
  int fmt = FMT_F64;
  int acc_kind = PIC_ACC_FIX64;  // resolved accumulator (never PIC_ACC_AUTO)
  int vec = 2;
  size_t esz = 8;          // particle element size (positions and velocities have the same width in every format)
  long long ld = 0;
  long long chunk = 0;
  int nblk = 0;
  int R = 1;
  int fg = 42;             // fractional bits of the fixed-point accumulators
  int S = 1;               // sub-rows per accumulator row (pic_device.h: acc_row_sum)
  double magic = 0;
  size_t sweep_lds = 0, solve_lds = 0;
  // resident schedule (pic_resident.h): one workgroup of res_nw waves holds an environment, res_ppt particles per lane
  bool resident = false;
  int res_ppt = 0, res_nw = 0, res_R = 1;
  bool res_lean = false;          // resident kernel without carried cell / weights (two workgroups per CU)
  size_t res_lds = 0;
  double dx = 0, scale = 0;
  double cs[4]{}, ds[4]{};
  hipStream_t stream = nullptr;       // the stream every call works on (own_stream, or the caller's)
  hipStream_t own_stream = nullptr;   // created by pic_create, destroyed by pic_destroy
  bool v_separate = false;            // v is an allocation of its own (large states: alloc_particles)
  int post_slot = -1;                 // ring row whose post-step solve rides with the next sweep C (inside pic_step only)
  bool refresh_pending = false;       // the last sweep was a D2 (inside pic_step only): the next sweep B deposits the positions it reads
  bool light_inner_steps = false;     // inner steps of a call end with sweep D2 (pic_create: particle states of 256 MB and more)
  double* hist_row = nullptr;         // where the NEXT post-step solve also records its three energies (step_recording), or null
  double* post_hist_row = nullptr;    // the same for the solve that post_slot stands for
  PlacementStats place{};             // what the search for an (x, v) placement did, all legs together (pic_placement_stats)
  PlacementState place_state{};       // what a later leg of it needs to know (placement_leg, resume_placement)
  void* x = nullptr;
  void* v = nullptr;
  void* scratch = nullptr;        // [env][ld] positions of a probe (eval_field / compute_E)
  void* stage = nullptr;          // [env][N] float staging: fixed-point positions <-> the caller's floats
  // accumulator ring: rows [env][Ng] of 64-bit fixed-point weight sums
  acc_t* ring = nullptr;
  std::vector<int> clean, dirty;  // rows that are zero / rows whose readers have all been enqueued
  hipError_t ring_error = hipSuccess;   // a clearing memset of ring_take_clean that failed (reported by launch_status)
  int q_slot = -1;                // row holding the deposit of the NEXT step's q1 (sweep A is skipped while >= 0)
  int stage_slot = -1;            // pic_step_stage: row the next stage's field comes from
  int sweep_parity = 0;           // direction of the next push sweep
  acc_t* probe_acc = nullptr;     // accumulator row of the probes (their own: a probe never touches step state)
  double* probe_ext = nullptr;    // device copy of a probe's host E_ext
  double* ke_part = nullptr;      // [env][nblk]
  double* n = nullptr;
  double* E_mesh = nullptr;
  double* phi = nullptr;
  double* ext = nullptr;          // device copy of a host E_ext / the actuator's field of a step, built once per environment (run_stages)
  double* ext2 = nullptr;         // ... of the step after it (a rollout alternates between the two)
  int ext_turn = 0;               // which of the two holds the field of the step being launched
  double* basis = nullptr;        // [2][Ng][M] actuator tables (cos, sin)
  double* act = nullptr;          // [env][2M] actions: device copy of a host action / the feedback law's current action
  double* modes = nullptr;        // [2][env][M] Fourier modes (re, im)
  int act_modes = 0;
  int modes_cap = 0;
  double* tw = nullptr;           // [2][tw_rows][Ng] twiddles of modes 1..tw_rows (pic_aux.h: twiddle_kernel)
  int tw_rows = 0;
  void* traj = nullptr;           // device copy of a host trajectory of actions or fields (pic_step_*_traj), grown on demand
  size_t traj_bytes = 0;
  unsigned long long* res_q1 = nullptr;   // resident schedule: [env][R (Ng + 2)] LDS mesh of the next step's q1 deposit, launch to launch
  bool res_q1_valid = false;
  void* res_carry = nullptr;              // ... and the cell and weights of every particle's q1 (pic_resident.h: ResidentEdge), or null
  bool res_carry_valid = false;
  const InlineDoubles* inline_act = nullptr;   // streaming schedule, for the duration of a call: the held action rides in the sweeps' argument blocks
  Feedback fb{};                  // feedback outputs wanted from the NEXT post-step solve of the streaming schedule (fb.M = 0: none)
  double* aux_n = nullptr;        // probe outputs
  double* aux_E = nullptr;
  double* aux_pe = nullptr;
  double* h_probe_pe = nullptr;   // pinned host: the energy of a probe, written by the solve itself (pic_eval_field of a small host state)
  bool probe_row_clean = false;   // the probes' accumulator row is zero (the solve of the last probe cleared it behind its read)
  double* aux_phi = nullptr;
  int mid_stage = 0;              // pic_step_stage: force evaluations of the current step already done (0 = between steps)
  double* KE = nullptr;
  double* PE = nullptr;
  double* PEr = nullptr;
  double* h_scal = nullptr;       // pinned host staging for KE | PE | PE_reward
  void* h_part = nullptr;         // pinned host staging for x | v of states up to 64 MB (from pic_create on up to 4 MB, else on first use)
  bool h_part_refused = false;    // ... could not be had: do not ask again
  double* h_fields = nullptr;     // pinned host staging for n | E_mesh | phi (meshes up to 256 KB each in total), or null
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

// ---- accumulator ring -------------------------------------------------------------------------
size_t row_elems(const pic_handle* h) { return (size_t)h->S * h->cfg.num_envs * h->cfg.Ng; }       // one accumulator row: [S][env][Ng]
acc_t* ring_row(pic_handle* h, int slot) { return h->ring + (size_t)slot * row_elems(h); }

// a zeroed row for the deposit of the sweep about to be launched
int ring_take_clean(pic_handle* h) {
  if (h->clean.empty()) {          // not reached by the step schedule (every sweep clears two retired rows)
    const int s = h->dirty.front();
    h->dirty.erase(h->dirty.begin());
    // (a failure here must not pass silently as a dirty row: it is kept for the caller's next check, pic_* entry points end
    // with HIPCHK(h, ring_status(h)) through hipGetLastError's siblings below)
    const hipError_t e = hipMemsetAsync(ring_row(h, s), 0, row_elems(h) * sizeof(acc_t), h->stream);
    if (e != hipSuccess && h->ring_error == hipSuccess) h->ring_error = e;
    return s;
  }
  const int s = h->clean.back();
  h->clean.pop_back();
  return s;
}

// what the launches enqueued since the last check have left behind: the runtime's sticky error, or a failed row clearing
hipError_t launch_status(pic_handle* h) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) { e = h->ring_error; }
  h->ring_error = hipSuccess;
  return e;
}

// the last kernel reading `slot` has been enqueued: any later sweep may clear it
void ring_retire(pic_handle* h, int slot) {
  if (slot >= 0) h->dirty.push_back(slot);
}

template <typename P, typename A, int SHAPE, int STAGE>
void launch_sweep_t(pic_handle* h, const SweepIO& io, void* x, void* v, const SweepArgs& a) {
  dim3 grid(h->nblk + ((STAGE == ST_C && io.post.acc) ? 1 : 0), h->cfg.num_envs);
  static const InlineDoubles none{};
  hipLaunchKernelGGL((sweep_kernel<P, A, SHAPE, STAGE>), grid, dim3(BLOCK), h->sweep_lds, h->stream,
                     static_cast<typename P::X*>(x), static_cast<typename P::V*>(v), io, a, a.act_inline ? *h->inline_act : none);
}

template <typename P, typename A, int SHAPE>
void launch_sweep_s(pic_handle* h, const SweepIO& io, int stage, void* x, void* v, const SweepArgs& a) {
  switch (stage) {
    case ST_A: launch_sweep_t<P, A, SHAPE, ST_A>(h, io, x, v, a); break;
    case ST_B: launch_sweep_t<P, A, SHAPE, ST_B>(h, io, x, v, a); break;
    case ST_C: launch_sweep_t<P, A, SHAPE, ST_C>(h, io, x, v, a); break;
    case ST_D: launch_sweep_t<P, A, SHAPE, ST_D>(h, io, x, v, a); break;
    case ST_REFRESH: launch_sweep_t<P, A, SHAPE, ST_REFRESH>(h, io, x, v, a); break;
    case ST_B2: launch_sweep_t<P, A, SHAPE, ST_B2>(h, io, x, v, a); break;
    case ST_D2: launch_sweep_t<P, A, SHAPE, ST_D2>(h, io, x, v, a); break;
    default: launch_sweep_t<P, A, SHAPE, ST_PROBE>(h, io, x, v, a); break;
  }
}

template <typename P>
void launch_sweep_p(pic_handle* h, const SweepIO& io, int stage, void* x, void* v, const SweepArgs& a) {
  const bool tsc = h->cfg.interpol == PIC_TSC;
  if constexpr (std::is_same<P, PosF64>::value) {
    if (h->acc_kind == PIC_ACC_F64) {
      if (tsc) launch_sweep_s<P, double, PIC_TSC>(h, io, stage, x, v, a);
      else launch_sweep_s<P, double, PIC_CIC>(h, io, stage, x, v, a);
      return;
    }
  } else {
    if (h->acc_kind == PIC_ACC_PACKED) { launch_sweep_s<P, fix_t, PIC_CIC>(h, io, stage, x, v, a); return; }
  }
  if (tsc) launch_sweep_s<P, acc_t, PIC_TSC>(h, io, stage, x, v, a);
  else launch_sweep_s<P, acc_t, PIC_CIC>(h, io, stage, x, v, a);
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
void prof_begin(pic_handle* h, int kind) {
  if (!h->prof) return;
  if (h->ev_kind.size() >= 16384) prof_drain(h);
  const size_t i = h->ev_kind.size();
  prof_reserve(h, i + 1);
  hipEventRecord(h->ev[2 * i], h->stream);
  h->ev_kind.push_back(kind);
}
void prof_end(pic_handle* h) {
  if (!h->prof) return;
  hipEventRecord(h->ev[2 * (h->ev_kind.size() - 1) + 1], h->stream);
}

// One sweep over all environments.  in_slot: ring row the gather field is solved from (gather stages);
// out / out2: rows (or the probe accumulator) receiving the deposits.  The sweep also clears up to two
// retired ring rows for later use.
void launch_sweep(pic_handle* h, int stage, void* x, void* v, double c_prev, double c_cur, double d_cur,
                  int in_slot, const Control& ctl, acc_t* out, acc_t* out2, int post_slot = -1, double* ext_out = nullptr,
                  const double* next_act = nullptr) {
  SweepArgs a;
  a.N = h->cfg.N; a.ld = h->ld; a.chunk = h->chunk; a.Ng = h->cfg.Ng; a.nblk = h->nblk; a.R = h->R;
  a.act_inline = (ctl.act && h->inline_act) ? 1 : 0;
  const bool push = stage <= ST_D || stage == ST_B2 || stage == ST_D2;
  a.reverse = push ? (h->sweep_parity ^= 1) : 0;
  a.fg = h->fg; a.magic = h->magic;
  a.S = h->S; a.sub = (long long)h->cfg.num_envs * h->cfg.Ng;
  a.L = h->cfg.L; a.dx = h->dx; a.dt = h->cfg.dt;
  a.rdx = h->fmt == FMT_F64 ? 1.0 / h->dx : (double)(1.0f / (float)h->dx);
  a.c_prev = c_prev; a.c_cur = c_cur; a.d_cur = d_cur; a.c_next = h->cs[0];
  a.scale = h->scale; a.n0 = h->cfg.n0;
  a.to_units = 4294967296.0 / h->cfg.L;
  a.N_over_L = (double)h->cfg.N / h->cfg.L;
  SweepIO io{};
  io.acc_in = in_slot >= 0 ? ring_row(h, in_slot) : nullptr;
  io.ctl = ctl;
  io.ext_out = (ctl.act || next_act) ? ext_out : nullptr;
  io.next_act = next_act;
  io.acc_out = out;
  io.acc_out2 = out2;
  int z[2] = {-1, -1};
  for (int k = 0; k < 2 && !h->dirty.empty(); ++k) {
    z[k] = h->dirty.front();
    h->dirty.erase(h->dirty.begin());
  }
  io.zero0 = z[0] >= 0 ? ring_row(h, z[0]) : nullptr;
  io.zero1 = z[1] >= 0 ? ring_row(h, z[1]) : nullptr;
  io.ke_part = h->ke_part;
  io.bad = h->bad;
  if (post_slot >= 0) {            // sweep C also carries the previous step's post-step refresh (pic_sweep.h: SweepIO::post)
    io.post.acc = ring_row(h, post_slot);
    io.post.ke_part = h->ke_part; io.post.n = h->n; io.post.out.E = h->E_mesh; io.post.out.phi = h->phi;
    io.post.out.KE = h->KE; io.post.out.PE = h->PE; io.post.out.PEr = h->PEr;
    io.post.out.hist = h->post_hist_row; io.post.out.num_envs = h->cfg.num_envs;
  }
  prof_begin(h, stage <= ST_D ? stage : (stage == ST_B2 ? (int)ST_B : (stage == ST_D2 ? (int)ST_D : 5)));
  if (h->fmt == FMT_F64) launch_sweep_p<PosF64>(h, io, stage, x, v, a);
  else if (h->fmt == FMT_F32) launch_sweep_p<PosF32>(h, io, stage, x, v, a);
  else launch_sweep_p<PosU32>(h, io, stage, x, v, a);
  prof_end(h);
  for (int k = 0; k < 2; ++k)
    if (z[k] >= 0) h->clean.push_back(z[k]);      // zero for every LATER launch of this stream
}

template <typename P, typename A, int SHAPE, int PPT, int NW>
void launch_resident_t(pic_handle* h, const ResidentIO& io, const SweepArgs& a, const InlineDoubles& act) {
  // More environments than CUs and a slot count whose lean kernel fits 128 registers: two workgroups per CU beat
  // the 15 % the carried cell / weights save per workgroup (profiles/experiments_r2.md 5).
  // (carrying three TSC weights for 16 particles per lane would need more than 256 registers: that kernel is not even compiled,
  // pic_create sets res_lean for the shape)
  constexpr bool kCarryFits = !(SHAPE == PIC_TSC && PPT == 16);
  if constexpr (kCarryFits) {
    if (!h->res_lean) {
      hipLaunchKernelGGL((resident_kernel<P, A, SHAPE, PPT, NW, true>), dim3(h->cfg.num_envs), dim3(NW * 64), h->res_lds,
                         h->stream, static_cast<typename P::X*>(h->x), static_cast<typename P::V*>(h->v), io, a, act);
      return;
    }
  }
  hipLaunchKernelGGL((resident_kernel<P, A, SHAPE, PPT, NW, false>), dim3(h->cfg.num_envs), dim3(NW * 64), h->res_lds,
                     h->stream, static_cast<typename P::X*>(h->x), static_cast<typename P::V*>(h->v), io, a, act);
}

template <typename P, typename A, int SHAPE>
void launch_resident_s(pic_handle* h, const ResidentIO& io, const SweepArgs& a, const InlineDoubles& act) {
  switch (h->res_nw * 100 + h->res_ppt) {
    case 804: launch_resident_t<P, A, SHAPE, 4, 8>(h, io, a, act); break;
    case 808: launch_resident_t<P, A, SHAPE, 8, 8>(h, io, a, act); break;
    case 810: launch_resident_t<P, A, SHAPE, 10, 8>(h, io, a, act); break;
    default: launch_resident_t<P, A, SHAPE, 16, 8>(h, io, a, act); break;
  }
}

template <typename P>
void launch_resident_p(pic_handle* h, const ResidentIO& io, const SweepArgs& a, const InlineDoubles& act) {
  const bool tsc = h->cfg.interpol == PIC_TSC;
  if constexpr (!std::is_same<P, PosF64>::value) {
    if (h->acc_kind == PIC_ACC_PACKED) { launch_resident_s<P, fix_t, PIC_CIC>(h, io, a, act); return; }
  }
  if (tsc) launch_resident_s<P, acc_t, PIC_TSC>(h, io, a, act);
  else launch_resident_s<P, acc_t, PIC_CIC>(h, io, a, act);
}

// What drives the external field of the steps of one call (device pointers; see Control / Feedback in pic_device.h)
struct StepControl {
  Control ctl{};           // first step's field or action (environment 0)
  long long ext_step = 0;  // elements between consecutive steps' fields / actions (0: held for the whole call)
  long long act_step = 0;
  Feedback fb{};           // fb.M > 0: feedback law; fb.act_hist = device [nsteps][env][2M] record of the actions, or null
  // one held action of few coefficients, given on the host: inside the resident kernel's own argument block (no copy, no launch)
  int inline_n = 0;
  InlineDoubles inline_act{};
};

// nsteps environment steps in one launch of the resident schedule; hist: device [nsteps][3][env] or null
void launch_resident(pic_handle* h, const StepControl& sc, int nsteps, double* hist, void* snap = nullptr) {
  SweepArgs a{};
  a.N = h->cfg.N; a.ld = h->ld; a.Ng = h->cfg.Ng; a.R = h->res_R;
  a.fg = h->fg; a.magic = h->magic;
  a.L = h->cfg.L; a.dx = h->dx; a.dt = h->cfg.dt;
  a.rdx = h->fmt == FMT_F64 ? 1.0 / h->dx : (double)(1.0f / (float)h->dx);
  a.scale = h->scale; a.n0 = h->cfg.n0;
  a.to_units = 4294967296.0 / h->cfg.L;
  a.N_over_L = (double)h->cfg.N / h->cfg.L;
  ResidentIO io{};
  io.nsteps = nsteps; io.num_envs = h->cfg.num_envs;
  io.c1 = h->cs[0]; io.c2 = h->cs[1]; io.d1 = h->ds[1]; io.d2 = h->ds[2];
  io.c.ctl = sc.ctl; io.c.ext_step = sc.ext_step; io.c.act_step = sc.act_step; io.c.fb = sc.fb;
  io.o.n = h->n; io.o.E = h->E_mesh; io.o.phi = h->phi; io.o.KE = h->KE; io.o.PE = h->PE; io.o.PEr = h->PEr;
  io.o.hist = hist;
  io.e.snap = snap; io.e.bad = h->bad;
  // the LDS mesh with the next step's q1 deposit travels from launch to launch (pic_invalidate and every reload drop it)
  io.e.q1_in = h->res_q1_valid ? h->res_q1 : nullptr;
  io.e.q1_out = h->res_q1;
  io.e.carry_in = h->res_q1_valid && h->res_carry_valid ? h->res_carry : nullptr;
  io.e.carry_out = h->res_carry;
  io.mode = (sc.inline_n > 0 ? RM_ACT_INLINE : 0) | (sc.ctl.ext || sc.ctl.act || sc.fb.M > 0 ? RM_EXT : 0) | (sc.ext_step || sc.act_step ? RM_PER_STEP : 0) |
            (sc.fb.M > 0 ? RM_FEEDBACK : 0) | (snap ? RM_SNAP : 0) | (hist || sc.fb.M > 0 ? RM_RECORD : 0);
  prof_begin(h, 6);
  if (h->fmt == FMT_F64) launch_resident_p<PosF64>(h, io, a, sc.inline_act);
  else if (h->fmt == FMT_F32) launch_resident_p<PosF32>(h, io, a, sc.inline_act);
  else launch_resident_p<PosU32>(h, io, a, sc.inline_act);
  prof_end(h);
}

void launch_solve(pic_handle* h, const SolveIO& io) {
  SolveArgs a;
  a.N = h->cfg.N; a.Ng = h->cfg.Ng; a.nblk = h->nblk; a.fg = h->fg; a.L = h->cfg.L; a.dx = h->dx; a.n0 = h->cfg.n0;
  a.scale = h->scale; a.N_over_L = (double)h->cfg.N / h->cfg.L;
  a.S = io.acc ? h->S : 1; a.sub = (long long)h->cfg.num_envs * h->cfg.Ng;
  prof_begin(h, 4);
  hipLaunchKernelGGL(field_solve_kernel, dim3(h->cfg.num_envs), dim3(SBLOCK), h->solve_lds, h->stream, io, a);
  prof_end(h);
}

// the post-step refresh (pic.py:145-146, no external field: pic.py:114-117) from the deposit in ring row `slot`
void launch_final_solve(pic_handle* h, int slot) {
  SolveIO o{};
  o.acc = ring_row(h, slot);
  o.ke_part = h->ke_part; o.n = h->n; o.out.E = h->E_mesh; o.out.phi = h->phi;
  o.out.KE = h->KE; o.out.PE = h->PE; o.out.PEr = h->PEr;
  o.out.hist = h->hist_row; o.out.num_envs = h->cfg.num_envs;
  o.out.fb = h->fb;
  launch_solve(h, o);
  ring_retire(h, slot);
}

void drop_cached_deposits(pic_handle* h) {
  ring_retire(h, h->q_slot);
  ring_retire(h, h->stage_slot);
  h->q_slot = h->stage_slot = -1;
  h->mid_stage = 0;
  h->res_q1_valid = false;
}

int refresh_fields(pic_handle* h) {
  drop_cached_deposits(h);
  const int f = ring_take_clean(h), qn = ring_take_clean(h);
  launch_sweep(h, ST_REFRESH, h->x, h->v, 0, 0, 0, -1, Control{}, ring_row(h, f), ring_row(h, qn));
  launch_final_solve(h, f);
  HIPCHK(h, launch_status(h));
  h->q_slot = qn;   // ST_REFRESH also deposited the next step's q1
  return PIC_OK;
}

dim3 aux_grid(pic_handle* h, int nenv, long long cap = 1024) {
  long long gx = (h->cfg.N + BLOCK - 1) / BLOCK;
  if (gx > cap) gx = cap;
  return dim3((unsigned)gx, nenv);
}

int ensure_stage(pic_handle* h) {
  if (h->stage) return PIC_OK;
  HIPCHK(h, hipMalloc(&h->stage, (size_t)h->cfg.num_envs * h->cfg.N * sizeof(float)));
  return PIC_OK;
}

// copy a dense [env][N] caller array of velocities (or float positions) into a padded [env][ld] device array
int upload(pic_handle* h, void* dst_padded, const void* src, int mem_kind) {
  const size_t row = (size_t)h->cfg.N * h->esz;
  HIPCHK(h, hipMemcpy2DAsync(dst_padded, (size_t)h->ld * h->esz, src, row, row, h->cfg.num_envs,
                             mem_kind == PIC_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
  return PIC_OK;
}

// positions arrive as floats of the particle dtype; the fixed-point format converts them on the device
int upload_positions(pic_handle* h, void* dst_padded, const void* src, int mem_kind) {
  if (h->fmt != FMT_U32) return upload(h, dst_padded, src, mem_kind);
  const float* dsrc = static_cast<const float*>(src);
  if (mem_kind == PIC_HOST) {
    int rc = ensure_stage(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->stage, src, (size_t)h->cfg.num_envs * h->cfg.N * sizeof(float), hipMemcpyHostToDevice, h->stream));
    dsrc = static_cast<const float*>(h->stage);
  }
  hipLaunchKernelGGL((positions_in_kernel<PosU32, float>), aux_grid(h, h->cfg.num_envs), dim3(BLOCK), 0, h->stream, dsrc,
                     static_cast<unsigned*>(dst_padded), h->cfg.N, h->ld, h->cfg.L, h->bad);
  HIPCHK(h, hipGetLastError());
  return PIC_OK;
}

int download(pic_handle* h, void* dst, const void* src_padded, int mem_kind) {
  const size_t row = (size_t)h->cfg.N * h->esz;
  HIPCHK(h, hipMemcpy2DAsync(dst, row, src_padded, (size_t)h->ld * h->esz, row, h->cfg.num_envs,
                             mem_kind == PIC_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream));
  return PIC_OK;
}

int download_positions(pic_handle* h, void* dst, const void* src_padded, int mem_kind) {
  if (h->fmt != FMT_U32) return download(h, dst, src_padded, mem_kind);
  float* ddst = static_cast<float*>(dst);
  if (mem_kind == PIC_HOST) {
    int rc = ensure_stage(h);
    if (rc) return rc;
    ddst = static_cast<float*>(h->stage);
  }
  hipLaunchKernelGGL((positions_out_kernel<PosU32, float>), aux_grid(h, h->cfg.num_envs), dim3(BLOCK), 0, h->stream,
                     static_cast<const unsigned*>(src_padded), ddst, h->cfg.N, h->ld, h->cfg.L);
  HIPCHK(h, hipGetLastError());
  if (mem_kind == PIC_HOST)
    HIPCHK(h, hipMemcpyAsync(dst, h->stage, (size_t)h->cfg.num_envs * h->cfg.N * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  return PIC_OK;
}

// mesh [env][Ng] gathered at the positions x [env][ld] with the handle's shape function -> out, dense [env][N]
template <typename P>
void launch_gather_p(pic_handle* h, const void* x, const double* mesh, void* out) {
  const dim3 grid = aux_grid(h, h->cfg.num_envs);
  const size_t lds = ((size_t)h->cfg.Ng + 2) * sizeof(typename P::W);
  if (h->cfg.interpol == PIC_TSC)
    hipLaunchKernelGGL((gather_E_kernel<P, PIC_TSC>), grid, dim3(BLOCK), lds, h->stream, (const typename P::X*)x, mesh,
                       (typename P::W*)out, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
  else
    hipLaunchKernelGGL((gather_E_kernel<P, PIC_CIC>), grid, dim3(BLOCK), lds, h->stream, (const typename P::X*)x, mesh,
                       (typename P::W*)out, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
}
void launch_gather(pic_handle* h, const void* x, const double* mesh, void* out) {
  if (h->fmt == FMT_F64) launch_gather_p<PosF64>(h, x, mesh, out);
  else if (h->fmt == FMT_F32) launch_gather_p<PosF32>(h, x, mesh, out);
  else launch_gather_p<PosU32>(h, x, mesh, out);
}

// indices and weights of `nenv` environments' worth of positions x [nenv][ld] -> idx, w [nenv][3][N]
template <typename P>
void launch_shape_query_p(pic_handle* h, const void* x, int nenv, int shape, long long* idx, double* w) {
  const dim3 grid = aux_grid(h, nenv);
  if (shape == PIC_TSC)
    hipLaunchKernelGGL((shape_query_kernel<P, PIC_TSC>), grid, dim3(BLOCK), 0, h->stream, (const typename P::X*)x, h->cfg.N,
                       h->ld, h->cfg.Ng, h->cfg.L, h->dx, idx, w);
  else
    hipLaunchKernelGGL((shape_query_kernel<P, PIC_CIC>), grid, dim3(BLOCK), 0, h->stream, (const typename P::X*)x, h->cfg.N,
                       h->ld, h->cfg.Ng, h->cfg.L, h->dx, idx, w);
}
void launch_shape_query(pic_handle* h, const void* x, int nenv, int shape, long long* idx, double* w) {
  if (h->fmt == FMT_F64) launch_shape_query_p<PosF64>(h, x, nenv, shape, idx, w);
  else if (h->fmt == FMT_F32) launch_shape_query_p<PosF32>(h, x, nenv, shape, idx, w);
  else launch_shape_query_p<PosU32>(h, x, nenv, shape, idx, w);
}

}  // namespace

extern "C" {

int pic_abi_version(void) { return PICSTEP_ABI_VERSION; }

const char* pic_last_error(pic_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

// Where x and v land in HBM decides how fast they stream together.  On an unfragmented MI355X the 288 GiB behave as nine regions
// of 32 GiB: a kernel that streams two arrays lying in the SAME region runs at 5.25 TB/s, with the arrays in two DIFFERENT
// regions at 6.05 TB/s, whichever regions and whatever the access pattern (profiles/window_probe.hip: one 120 GiB block, x fixed,
// v moved through it; profiles/experiments_r2.md 15).  A fresh device hands out neighbouring memory, so x and v of a default
// allocation share a region almost always; on a device whose memory has been through other processes a block is a mixture of
// pages from several regions (profiles/touch_probe.hip, experiments_r4.md 1: the class of a 64 MB window follows the window of x
// it is paired with, not the candidate), which is why the search times WHOLE blocks, never windows of them.
// For particle states that live in HBM (>= 256 MB) x and v are therefore two allocations: x first, then blocks of the same
// size one after the other (they are laid down in sequence), and every 3 GiB the pair (x, newest block) is timed with a streaming
// pass.  The search is a policy on RATIOS, not on this part's numbers: it ends sixteen readings after the best pair seen streams
// >= 10 % faster than the slowest one seen (the kinds have been told apart and we hold a fast one; the best of all is kept), after
// 42 GiB walked without an improvement (more than a region, all pairs alike: nothing to gain on this device), or when a third of
// the free memory is held; everything but x and v is freed before the call returns.
// pic_config.placement = PIC_PLACE_OFF skips it (x | v in one block).  Smaller states keep x | v in one block too (they sit in
// the Infinity Cache, and the one-copy read-back of pic_get_particles wants them adjacent).
//
// Where the time goes, and why the search comes in LEGS of at most 100 ms (round 4, profiles/experiments_r4.md 1).
// * Nothing is paid for the first touch of a block (touch_probe: first pass 330 us, later ones 347): a candidate is not cleared
//   here, a reading is one timed pass behind one untimed pass.
// * What costs is hipMalloc of memory the device hands out for the first time since it came up: the driver clears it, 1.3 ms per
//   512 MB block with the GPU otherwise idle and 3-6 ms under a streaming kernel (released memory is wiped in the background and
//   comes back in 20-70 us; a hipMalloc right behind the exit of a process that held tens of gigabytes can also sit and wait for
//   that wipe, 0.6-1.5 s seen -- nothing a caller of hipMalloc can bound).  A first create on such a device has x at the very start
//   of a region, 31 GiB -- 80 to 200 ms of allocations -- from the first block that pairs well with it.  No budget that a
//   constructor may take covers that.  But what one
//   leg has cleared and given back stays clean, so the NEXT leg walks through it in microseconds per block and spends its 100 ms
//   beyond: pic_create runs the first leg, and while it ends for lack of time pic_reset / pic_reset_sampled -- which replace the
//   particles anyway, so that moving v costs nothing -- run further ones (at most kMaxLegs, and only as long as pic_device_ptrs has
//   not handed the addresses to anybody).
// * The blocks are allocated by a thread of the call's own while the calling thread times; over never-used memory (slow mallocs) the
//   two take turns instead, because the clear and the timed stream slow each other down.
// * A device that has rested >= 3 ms runs its next 10-20 ms 4-13 % slow (early_steps3.py), and readings taken at different points
//   of that ramp show a "10 % faster" pair of the SAME kind.  Every reading is therefore a RATIO: the time of (x, candidate) over
//   the time of (x, the leg's first block) taken in the same breath (again whenever the stream has rested since), behind a filler.
struct BlockFeed {                                  // candidate blocks, allocated by a thread of their own (placement_leg)
  std::mutex m;
  std::condition_variable cv;
  std::vector<void*> blocks;                        // in allocation order; only ever grown by the feeder
  size_t taken = 0;                                 // blocks.size() when the timing thread last took one
  size_t lead = 1;                                  // the feeder stays at most this many blocks ahead of `taken`
  bool stop = false, done = false;
  bool timing = false;                              // a reading is being taken
  bool slow = false;                                // the last hipMalloc was of never-used memory (being cleared): take turns with the readings
  double malloc_seconds = 0.0;
};

constexpr int kMaxLegs = 4;

// One leg of the search.  On entry h->x is allocated; h->v is the block kept so far, or null (first leg).
void placement_leg(pic_handle* h, size_t pbytes) {
  // Everything the search allocates has to be given back, and the driver wipes released memory before it hands it out again
  // (asynchronously; whatever allocates next on the device may wait for that): an untouched 32 GiB spacer that carried the search out
  // of x's own region at once made the next pic_create of a create / destroy loop take 0.4-3 s (experiments_r3.md 18).  Blocks of
  // the state's own size, given back within the call, do not.
  constexpr size_t kLead = (size_t)3 << 30;         // distance between two readings
  constexpr double kGain = 1.10;                    // slowest / best (normalised) at which the search has found what it looks for
  constexpr size_t kPatience = (size_t)42 << 30;    // walked without an improvement before giving up: more than the 32 GiB a region
                                                    // spans (15 GiB gave up inside x's own region on some boxes: 1049 instead of 958 us)
  constexpr int kMore = 16;                         // readings beyond the first that passes kGain
  // per leg, the release of the blocks included: 100 ms, or what forty steps of the handle being placed take if that is more (a
  // step moves 12 x pbytes at ~6 TB/s: 1 ms at config 2, 4 ms at config 4's share, 10 ms at config 5's -- whose 2-5 GB blocks cost
  // 5-60 ms each to allocate on a device that hands them out for the first time)
  const double kMaxSeconds = h->cfg.placement_ms > 0 ? 1e-3 * h->cfg.placement_ms : std::max(0.100, 40.0 * 12.0 * (double)pbytes / 6.0e12);
  constexpr double kFreeSeconds = 0.0002;           // what giving one block back costs (hipFree: 25 ms for 110 blocks)
  constexpr double kSlowPerGiB = 0.0008;            // a hipMalloc slower than this per GiB is clearing never-used memory
  constexpr int kMaxBlocks = 192;
  PlacementStats& st = h->place;
  PlacementState& ps = h->place_state;
  const auto t_begin = std::chrono::steady_clock::now();
  auto seconds = [t_begin]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
  ps.legs += 1;
  size_t free_b = 0, total_b = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
  const size_t budget = free_b / 3;
  const long long n2 = (long long)(pbytes / sizeof(double2));
  long long nb = n2 / ((long long)BLOCK * 8);
  if (nb < 256) nb = 256;
  const long long chunk2 = (n2 + nb - 1) / nb;
  const long long nbh = (nb + 1) / 2, chunk2h = (n2 / 2 + nbh - 1) / nbh;
  double2* xa = static_cast<double2*>(h->x);
  // filler: the two halves of x streamed against each other (the same kernel at half the size), ~0.1 ms per GB of state
  auto filler = [&](int passes) {
    for (int r = 0; r < passes; ++r)
      hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)nbh), dim3(BLOCK), 0, h->stream, xa, xa + n2 / 2, n2 / 2, chunk2h, 1.0, r & 1);
  };
  const double pass_ms_guess = 2.0 * (double)pbytes / 5.0e9;           // one filler pass moves 2 x pbytes at ~5 TB/s
  const int fill_1ms = std::max(1, (int)std::ceil(1.0 / pass_ms_guess));
  auto pair_ms = [&](void* vb, float* ms) {                            // one untimed pass over (x, block), one timed
    double2* b = static_cast<double2*>(vb);
    hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, h->stream, xa, b, n2, chunk2, 1.0, 0);
    bool good = hipGetLastError() == hipSuccess && hipEventRecord(e0, h->stream) == hipSuccess;
    hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, h->stream, xa, b, n2, chunk2, 1.0, 0);
    return good && hipGetLastError() == hipSuccess && hipEventRecord(e1, h->stream) == hipSuccess &&
           hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(ms, e0, e1) == hipSuccess;
  };
  const double gb_per_ms = 4.0 * (double)pbytes / 1e6;                // one pass, 2 arrays read and written: GB/s = this / ms

  BlockFeed feed;
  const size_t lead_blocks = std::max<size_t>(1, kLead / pbytes);
  const int device = h->cfg.device_id;
  std::thread feeder;
  if (ok && !ps.x_cleared) {
    ok = hipMemsetAsync(h->x, 0, pbytes, h->stream) == hipSuccess;    // (x holds particles in a later leg: the passes scale by 1.0)
    ps.x_cleared = true;
  }
  if (ok) {
    feed.lead = lead_blocks;
    try {
    feeder = std::thread([&feed, seconds, pbytes, budget, device, kMaxSeconds]() {
      const bool dev_ok = hipSetDevice(device) == hipSuccess;
      for (;;) {
        {
          std::unique_lock<std::mutex> lk(feed.m);
          feed.cv.wait(lk, [&] { return feed.stop || (feed.blocks.size() < feed.taken + feed.lead && !(feed.slow && feed.timing)); });
          if (feed.stop || !dev_ok || (int)feed.blocks.size() >= kMaxBlocks || (feed.blocks.size() + 2) * pbytes > budget ||
              seconds() + kFreeSeconds * (double)feed.blocks.size() > kMaxSeconds)
            break;
        }
        void* b = nullptr;
        const double tm = seconds();
        const bool got = hipMalloc(&b, pbytes) == hipSuccess;
        const double dt = seconds() - tm;
        std::lock_guard<std::mutex> lk(feed.m);
        feed.malloc_seconds += dt;
        feed.slow = dt > kSlowPerGiB * ((double)pbytes / (double)(1ull << 30));
        if (!got) { (void)hipGetLastError(); break; }
        feed.blocks.push_back(b);
        feed.cv.notify_all();
      }
      std::lock_guard<std::mutex> lk(feed.m);
      feed.done = true;
      feed.cv.notify_all();
    });
    } catch (...) {                                                   // no thread to be had: no search
      ok = false;
    }
  }
  // normalised readings: time of (x, block) / time of (x, the leg's first block) taken in the same breath
  void* ref = nullptr;
  float ref_ms = 0.f;
  double last_reading_at = -1.0;                                      // seconds() when the stream last finished a reading
  auto reading = [&](void* b, double* norm, float* raw_ms) {
    const double t0 = seconds();
    {
      std::lock_guard<std::mutex> lk(feed.m);
      feed.timing = true;
    }
    bool good = true;
    const bool rested = last_reading_at < 0.0 || t0 - last_reading_at > 0.0005;
    if (rested) {                                                     // the reference again, behind a filler: same point of the ramp
      filler(last_reading_at < 0.0 ? 4 * fill_1ms : fill_1ms);
      good = pair_ms(ref, &ref_ms);
    }
    if (good && b != ref) good = pair_ms(b, raw_ms); else *raw_ms = ref_ms;
    last_reading_at = seconds();
    {
      std::lock_guard<std::mutex> lk(feed.m);
      feed.timing = false;
      feed.cv.notify_all();
    }
    *norm = (double)*raw_ms / (double)ref_ms;
    st.timing_seconds += seconds() - t0;
    return good;
  };
  void* best = h->v;                                                  // the block kept by earlier legs, or null
  double best_n = ps.best_n, worst_n = ps.worst_n;                    // normalised; 0 = none yet
  float best_raw = 0.f, worst_raw = 0.f;
  int timed = 0, found_at = 0;
  size_t last = 0, best_at = 0;                                       // blocks.size() at the last / at the best reading
  int outcome = PIC_PLACED_MEMORY;                                    // (the feeder ran into the block or memory limit, or hipMalloc failed)
  while (ok) {
    void* b = nullptr;
    {
      std::unique_lock<std::mutex> lk(feed.m);
      // the next reading is due `lead` blocks further on (or on what the feeder managed before it stopped)
      feed.cv.wait(lk, [&] { return feed.done || feed.blocks.size() >= last + feed.lead; });
      if (feed.blocks.size() == last) break;                          // the feeder has stopped and every block it made has been looked at
      last = feed.taken = feed.blocks.size();
      b = feed.blocks.back();
      if (found_at > 0) feed.lead = 1;                                // (past the first find every block is looked at: fewer to give back)
      if (!ref) ref = feed.blocks.front();
      feed.cv.notify_all();
    }
    if (seconds() + kFreeSeconds * (double)last > kMaxSeconds) {      // (the 100 ms include giving the blocks back)
      outcome = PIC_PLACED_TIMEOUT;
      break;
    }
    if (last <= ps.frontier) continue;                                // an earlier leg has been here: nothing new to learn
    double n = 0.0;
    float raw = 0.f;
    if (timed == 0) {
      // first reading of a leg: the reference itself (first leg: it is a candidate like any other, n = 1), or where the block
      // kept by the earlier legs stands today
      void* first = h->v ? h->v : ref;
      ok = reading(first, &n, &raw);
      if (!ok) break;
      best = first; best_n = n; best_raw = raw; best_at = last;
      if (worst_n < n) { worst_n = n; worst_raw = raw; }
      ++timed;
      if (b == first) continue;
    }
    ok = reading(b, &n, &raw);
    if (!ok) break;
    ++timed;
    if (best_n == 0.0 || n < best_n) { best = b; best_n = n; best_raw = raw; best_at = last; }
    if (n > worst_n) { worst_n = n; worst_raw = raw; }
    if (worst_n >= kGain * best_n) {                                  // a fast pair, known to be one ...
      // ... but there are more than two kinds (5.0-5.3 / 5.6-5.75 / 5.85-6.0 TB/s read on used devices, 0.983 / 0.970 / 0.963 ms per
      // step at config 2), and the first pair 10 % above the slowest is often of the middle one: a reading costs 0.7 ms, so
      // kMore further blocks are looked at and the best of all is kept
      if (found_at == 0) found_at = timed;
      if (timed - found_at >= kMore) { outcome = PIC_PLACED_FOUND; break; }
      continue;
    }
    // (for the 2-5 GB blocks of configs 4 and 5 that is sixteen blocks at least: nine alike have been followed by a fast one)
    if ((last - best_at) * pbytes >= kPatience && last - best_at >= 16) { outcome = PIC_PLACED_PATIENCE; break; }
  }
  if (feeder.joinable()) {
    {
      std::lock_guard<std::mutex> lk(feed.m);
      feed.stop = true;
      feed.cv.notify_all();
    }
    feeder.join();
  }
  if (outcome == PIC_PLACED_MEMORY && seconds() + kFreeSeconds * (double)feed.blocks.size() > kMaxSeconds)
    outcome = PIC_PLACED_TIMEOUT;                                     // (the feeder's own clock check)
  if (found_at > 0) outcome = PIC_PLACED_FOUND;                       // a fast pair is in hand: no further leg for the rest of the sixteen
  (void)hipStreamSynchronize(h->stream);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  (void)hipGetLastError();
  std::vector<void*>& blocks = feed.blocks;
  if (!best && !blocks.empty()) best = blocks.front();                // nothing could be timed: any block will do
  const double tf = seconds();
  for (void* b : blocks)
    if (b != best) hipFree(b);
  if (h->v && best != h->v) hipFree(h->v);                            // a later leg found a better block than the one kept
  st.free_seconds += seconds() - tf;
  h->v = best;
  ps.best_n = best_n; ps.worst_n = worst_n;
  ps.found = found_at > 0;
  ps.frontier = std::max(ps.frontier, blocks.size());
  st.blocks += (int)blocks.size();
  st.pairs_timed += timed;
  st.malloc_seconds += feed.malloc_seconds;
  st.outcome = outcome;
  if (best_raw > 0.f) st.kept_gbytes_per_s = gb_per_ms / best_raw;
  if (worst_raw > 0.f && (st.slowest_gbytes_per_s == 0.0 || gb_per_ms / worst_raw < st.slowest_gbytes_per_s))
    st.slowest_gbytes_per_s = gb_per_ms / worst_raw;
  st.seconds += seconds();
  st.legs = ps.legs;
}

hipError_t alloc_particles(pic_handle* h, size_t pbytes) {
  constexpr size_t kMinBytes = (size_t)256 << 20;
  h->place = PlacementStats{};
  h->place_state = PlacementState{};
  if (2 * pbytes < kMinBytes || h->cfg.placement == PIC_PLACE_OFF) {
    const hipError_t e = hipMalloc(&h->x, 2 * pbytes);
    h->v = static_cast<char*>(h->x) + pbytes;
    return e;
  }
  hipError_t e = hipMalloc(&h->x, pbytes);
  if (e != hipSuccess) return e;
  h->v_separate = true;
  h->place_state.pbytes = pbytes;
  placement_leg(h, pbytes);
  if (!h->v) return hipMalloc(&h->v, pbytes);                          // no candidate at all (no memory to search in): plain allocation
  return hipSuccess;
}

// A reset replaces the particles: while the search has only ended for lack of time, and nobody outside has been given the arrays'
// addresses, it may run another leg and move v for nothing.
void resume_placement(pic_handle* h) {
  PlacementState& ps = h->place_state;
  if (!h->v_separate || ps.pbytes == 0 || ps.ptrs_exposed || ps.legs >= kMaxLegs || h->place.outcome != PIC_PLACED_TIMEOUT) return;
  (void)hipStreamSynchronize(h->stream);
  placement_leg(h, ps.pbytes);
}

int pic_create(const pic_config* cfg, pic_handle** out) {
  if (!cfg || !out) return fail(nullptr, PIC_EINVAL, "pic_create: null argument");
  *out = nullptr;
  if (cfg->N < 1 || cfg->Ng < 4 || cfg->num_envs < 1 || !(cfg->L > 0) || !(cfg->dt > 0) || !(cfg->n0 > 0))
    return fail(nullptr, PIC_EINVAL, "pic_create: need N>=1, Ng>=4, num_envs>=1, L>0, dt>0, n0>0");
  if (cfg->N > (1ll << 36)) return fail(nullptr, PIC_EINVAL, "pic_create: N > 2^36");
  if (cfg->num_envs > 65535) return fail(nullptr, PIC_EINVAL, "pic_create: num_envs > 65535");
  if (cfg->env_index_base < 0) return fail(nullptr, PIC_EINVAL, "pic_create: env_index_base < 0");
  if (cfg->particle_dtype != PIC_F64 && cfg->particle_dtype != PIC_F32)
    return fail(nullptr, PIC_EINVAL, "pic_create: particle_dtype must be PIC_F64 or PIC_F32");
  if (cfg->position_dtype != PIC_POS_FLOAT && cfg->position_dtype != PIC_POS_FIXED32)
    return fail(nullptr, PIC_EINVAL, "pic_create: position_dtype must be PIC_POS_FLOAT or PIC_POS_FIXED32");
  if (cfg->position_dtype == PIC_POS_FIXED32 && cfg->particle_dtype != PIC_F32)
    return fail(nullptr, PIC_EINVAL, "pic_create: 32-bit fixed-point positions go with float32 particles");
  if (cfg->accum_dtype < PIC_ACC_AUTO || cfg->accum_dtype > PIC_ACC_F64)
    return fail(nullptr, PIC_EINVAL, "pic_create: accum_dtype must be PIC_ACC_AUTO, _FIX64, _PACKED or _F64");
  if (cfg->interpol != PIC_CIC && cfg->interpol != PIC_TSC)
    return fail(nullptr, PIC_EINVAL, "pic_create: interpol must be PIC_CIC or PIC_TSC");
  if (cfg->accum_dtype == PIC_ACC_PACKED && cfg->particle_dtype != PIC_F32)
    return fail(nullptr, PIC_EINVAL, "pic_create: the packed accumulator needs float32 particles");
  if (cfg->accum_dtype == PIC_ACC_PACKED && cfg->interpol != PIC_CIC)
    return fail(nullptr, PIC_EINVAL, "pic_create: the packed accumulator is CIC only");
  if (cfg->placement != PIC_PLACE_AUTO && cfg->placement != PIC_PLACE_OFF)
    return fail(nullptr, PIC_EINVAL, "pic_create: placement must be PIC_PLACE_AUTO or PIC_PLACE_OFF");
  if (cfg->placement_ms < 0) return fail(nullptr, PIC_EINVAL, "pic_create: placement_ms < 0");
  if (cfg->accum_dtype == PIC_ACC_F64 && cfg->particle_dtype != PIC_F64)
    return fail(nullptr, PIC_EINVAL, "pic_create: the float64 accumulator needs float64 particles");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(nullptr, PIC_EHIP, "pic_create: no HIP device visible (this library has no CPU path)");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, PIC_EINVAL, "pic_create: bad device_id");
  {
    // the library holds gfx950 code objects only: say so here, not as hipErrorNoBinaryForGpu at the first launch
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess)
      return fail(nullptr, PIC_EHIP, "pic_create: hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(nullptr, PIC_EHIP, std::string("pic_create: device ") + std::to_string(cfg->device_id) + " is " + prop.gcnArchName +
                                         ": libpicstep.so is built for gfx950 (MI355X) only");
  }

  pic_handle* h = new (std::nothrow) pic_handle();
  if (!h) return fail(nullptr, PIC_ENOMEM, "pic_create: out of host memory");
  h->cfg = *cfg;
  h->fmt = cfg->particle_dtype == PIC_F64 ? FMT_F64 : (cfg->position_dtype == PIC_POS_FIXED32 ? FMT_U32 : FMT_F32);
  h->acc_kind = cfg->accum_dtype;
  if (h->acc_kind == PIC_ACC_AUTO)
    h->acc_kind = (cfg->particle_dtype == PIC_F32 && cfg->interpol == PIC_CIC) ? PIC_ACC_PACKED : PIC_ACC_FIX64;
  h->esz = cfg->particle_dtype == PIC_F64 ? 8 : 4;
  h->vec = cfg->particle_dtype == PIC_F64 ? 2 : 4;
  h->dx = cfg->L / cfg->Ng;                                   // pic.py:36
  h->scale = cfg->n0 * cfg->L / (double)cfg->N / h->dx;       // interpolate.py:18
  yoshida_coefficients(h->cs, h->ds);
  h->ld = (cfg->N + 63) / 64 * 64;
  // fixed-point accumulators: the weights of all N particles on one node must fit in 63 bits
  int lg = 0;
  while ((1ll << lg) < cfg->N + 1) ++lg;
  h->fg = 62 - lg > 50 ? 50 : 62 - lg;
  h->magic = std::ldexp(1.5, 52 - h->fg);

  // workgroups per environment: enough in total to fill 256 CUs several times, at least one
  // BLOCK*VEC tile each
  const long long tile = (long long)BLOCK * h->vec;
  long long nblk = cfg->blocks_per_env;      // workgroups per environment of the streaming sweeps (resets and probes always use them)
  if (nblk <= 0) {
    const long long target_total = 8192;      // ~128 workgroups per env at 64 envs (profiles/experiments_r1.md)
    nblk = (target_total + cfg->num_envs - 1) / cfg->num_envs;
    // Large environments: >= 8 tiles per workgroup (amortises the prologue and the flush).  Small ones in small
    // ensembles are latency-bound (profiles/experiments_r2.md: N = 1e5 13 -> 98 workgroups 47 -> 28 us/step; N = 1e6
    // is best at 122 whatever the number of environments): one tile per workgroup, at most 64 workgroups per environment.
    const bool small = (double)cfg->N * cfg->num_envs <= 4.0e6 && cfg->N <= 131072;
    long long tiles_min = small ? 1 : 8;
    // one or two large environments: 4 tiles per workgroup, so that there is a workgroup for every CU (N = 1e6, one
    // environment: 122 workgroups 64 us/step, 245 56 us, 489 63 us)
    if (!small && (cfg->N + 8 * tile - 1) / (8 * tile) * cfg->num_envs < 256) tiles_min = 4;
    long long max_by_work = (cfg->N + tiles_min * tile - 1) / (tiles_min * tile);
    if (small && max_by_work > 64) max_by_work = 64;
    if (nblk > max_by_work) nblk = max_by_work;
    if (nblk < 1) nblk = 1;
    // ... and at most ~10 tiles (80 KB of x, 80 KB of v) per workgroup: a total of 8192 workgroups is 156 000 particles each at
    // config 5's share -- 340 us per workgroup, and a last partial round of workgroups that long at the end of every sweep.  Scans
    // of both large shares (profiles/bpe_big.sh): N = 4e6 x 64 float64 128 -> 384 workgroups per environment 4078 -> 3997 us per
    // step (512: 4029), N = 1e7 x 128 float32 64 -> 512: 10223 -> 9913 (768: 9928, 1024: 10058); config 2 (122) is not touched.
    const long long by10 = (cfg->N + 10 * tile - 1) / (10 * tile);
    if (!small && nblk < by10) nblk = by10;
    // A handful of large environments run as one to six workgroups per CU: a total that fills the CUs unevenly leaves some
    // with one workgroup more than others for the whole sweep (3 x 1e6: 3 x 123 = 369 workgroups on 256 CUs 63.3 us/step, 3 x 163
    // = 489 58.9).  Take the workgroups per environment from the smallest k >= 2 workgroups per CU that keeps >= 8 tiles' worth
    // ... per workgroup where it can (k ncu / E, at most the 4-tile count); 1, 2, 4, 6, 8, 12 environments keep what they had.
    if (!small) {
      int ncu = 256;
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device_id);
      const long long by4 = (cfg->N + 4 * tile - 1) / (4 * tile);
      if (nblk * cfg->num_envs < 6ll * ncu) {
        for (long long k = 2; k <= 6; ++k) {
          long long c = k * ncu / cfg->num_envs;
          if (c > by4) c = by4;
          if (c >= nblk || c == by4) { nblk = c; break; }
        }
      }
    }
  }
  {                                          // a workgroup's chunk of x or v is addressed with 31-bit byte offsets (StreamOut)
    const long long cap = (1ll << 27) - tile;
    if (nblk < (cfg->N + cap - 1) / cap) nblk = (cfg->N + cap - 1) / cap;
  }
  if (h->acc_kind == PIC_ACC_PACKED) {       // count field of the packed accumulator: < 2^20 particles per workgroup
    const long long cap = (1ll << 20) - tile;
    if (nblk < (cfg->N + cap - 1) / cap) nblk = (cfg->N + cap - 1) / cap;
  }
  long long chunk = (cfg->N + nblk - 1) / nblk;
  chunk = (chunk + tile - 1) / tile * tile;
  nblk = (cfg->N + chunk - 1) / chunk;
  if (nblk > 65535) { delete h; return fail(nullptr, PIC_EINVAL, "pic_create: blocks_per_env too large"); }
  h->chunk = chunk;
  h->nblk = (int)nblk;
  // Sub-rows of an accumulator row (pic_device.h: acc_row_sum): with few environments all workgroups of an environment flush
  // at about the same time, and their atomics on one 8 Ng-byte row are serialised at the memory side.  At most 4 sub-rows (1 / 2 /
  // 3 / 4 environments of 1e6: 31.2 / 45.0 / 64.5 / 74.0 us per step with 4, 31.6 / 45.3 / 65.5 / 75.1 with 8, 33.3 / 47.8 /
  // 65.2 / 75.4 with 16: every reader sums them), at least 8 workgroups per sub-row; with 16 environments or more the rows
  // themselves spread the traffic (and the flushes hide under the streaming of the other workgroups).
  // Inner steps of a multi-step call leave the deposit of their final positions to the next step's sweep B2 (run_stages) where the
  // sweeps are bound by HBM -- a particle state that does not fit the 256 MB Infinity Cache: there sweep B has the issue slots that
  // sweep D lacks (config 2 969 -> 948 us per step).  States that live in the cache, or are bound by the latency of each launch,
  // gain nothing or lose (all measured on one box, round 3's tree against this one: 8 x 1e6 float64 131.2 -> 132.6, 64 x 20000
  // float32 TSC 26.2 -> 27.2, config 1 14.5 -> 14.7, one environment of N = 1e5 19.1 -> 19.4): they keep the full sweep D.
  h->light_inner_steps = 2.0 * (double)cfg->num_envs * (double)h->ld * (double)h->esz >= 256.0 * 1048576.0;
  h->S = 1;
  while (h->S < 4 && nblk / (2 * h->S) >= 8 && (long long)cfg->num_envs * 2 * h->S <= 32) h->S *= 2;

  const size_t stride = (size_t)cfg->Ng + 2;
  // LDS: 2 R meshes (sweep D deposits two) + the field tile.  R = 1: one mesh for the eight waves of a workgroup.  Copies per
  // wave pair (R = 4, rounds 1-2) bought nothing at config 2 and cost 2-4 % where a step is short (more to sum and clear per
  // workgroup); even with every particle in ONE cell a sweep is only 16 % slower, with 1 copy as with 4
  // (profiles/experiments_r2.md 17, profiles/clustered.py)
  h->R = 1;
  h->sweep_lds = 2 * h->R * stride * 8 + stride * h->esz;
  h->solve_lds = 2 * (size_t)cfg->Ng * sizeof(double);
  // a workgroup may use 64 KB of LDS: the dynamic part sized here plus the kernels' static arrays (kSweepStaticLds, kResidentStaticLds)
  constexpr size_t kLdsLimit = 64 * 1024;
  if (h->sweep_lds + kSweepStaticLds > kLdsLimit) {
    const long long max_ng = (long long)((kLdsLimit - kSweepStaticLds) / (2 * h->R * 8 + h->esz)) - 2;
    delete h;
    return fail(nullptr, PIC_EINVAL, "pic_create: Ng too large for the LDS-resident mesh (at most " + std::to_string(max_ng) +
                                     " cells with this particle dtype)");
  }
  // Resident schedule (pic_resident.h): environments whose particles fit one workgroup's registers are stepped by
  // one launch per pic_step call.  blocks_per_env: 0 = use it where it applies, > 0 = streaming sweeps with that many
  // workgroups, -1 = resident or fail.
  {
    // 512-thread workgroups holding 4, 8, 10 or 16 particles per lane (1024 threads leave 128 registers per lane:
    // the 10- and 16-particle bodies spill there, so larger environments stay with the sweeps)
    static const int shapes[4][3] = {{8, 4, 2048}, {8, 8, 4096}, {8, 10, 5120}, {8, 16, 8192}};
    for (const auto& sh : shapes)
      if (cfg->N <= sh[2]) { h->res_nw = sh[0]; h->res_ppt = sh[1]; break; }
    h->res_R = 1;   // one mesh per workgroup: replicas cost more in node sums and clearing than they save in LDS atomic contention (experiments_r2.md 17)
    auto need = [&](int R) { return (size_t)2 * R * stride * 8 + 4 * (size_t)cfg->Ng * 8 + stride * h->esz; };
    while (h->res_R > 1 && need(h->res_R) > 48 * 1024) h->res_R >>= 1;
    h->res_lds = need(h->res_R);
    const bool possible = h->res_nw != 0 && h->res_lds + kResidentStaticLds <= kLdsLimit && h->acc_kind != PIC_ACC_F64;
    if (cfg->blocks_per_env < 0 && !possible) {
      const long long max_ng = (long long)((kLdsLimit - kResidentStaticLds - 2 * (2 * 8 + h->esz)) / (2 * 8 + h->esz + 4 * 8));
      delete h;
      return fail(nullptr, PIC_EINVAL, "pic_create: the resident schedule needs N <= 8192, Ng <= " + std::to_string(max_ng) +
                                       " (this particle dtype) and an integer accumulator");
    }
    // Measured (profiles/experiments_r2.md): one workgroup steps 5000 float64 particles in ~17 us whatever the number of
    // environments, the sweeps need 22 us for one environment of 8000 and 35-110 us for 64-1024 of 5000.  A lone
    // large-ish environment is therefore left to the sweeps (they spread it over many CUs).
    const bool worth = cfg->N <= 5120 || cfg->num_envs >= 32;
    h->resident = possible && (cfg->blocks_per_env < 0 || (cfg->blocks_per_env == 0 && worth));
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device_id);
    // (carrying three TSC weights for 16 particles per lane would need more than 256 registers)
    h->res_lean = (cfg->num_envs > ncu && h->res_ppt <= 10) || (cfg->interpol == PIC_TSC && h->res_ppt == 16);
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
  CREATE_CHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  const size_t pbytes = (size_t)cfg->num_envs * h->ld * h->esz;
  const size_t gbytes = (size_t)cfg->num_envs * cfg->Ng * sizeof(double);
  CREATE_CHK(alloc_particles(h, pbytes));            // x, v: [env][ld] each
  CREATE_CHK(hipMemsetAsync(h->x, 0, pbytes, h->stream));
  CREATE_CHK(hipMemsetAsync(h->v, 0, pbytes, h->stream));
  // small states (the reference's N = 5000) are read back every step by a Gym-style loop: one copy of x and v
  // together into pinned memory instead of two copies into pageable memory
  if (2 * (size_t)cfg->num_envs * cfg->N * h->esz <= ((size_t)4 << 20))
    CREATE_CHK(hipHostMalloc(&h->h_part, 2 * (size_t)cfg->num_envs * h->ld * h->esz, hipHostMallocDefault));
  CREATE_CHK(hipMalloc((void**)&h->ring, (size_t)(RING + 1) * h->S * gbytes));      // acc_t and double are both 8 bytes
  CREATE_CHK(hipMemsetAsync(h->ring, 0, (size_t)(RING + 1) * h->S * gbytes, h->stream));
  h->probe_acc = ring_row(h, RING);
  for (int s = 0; s < RING; ++s) h->clean.push_back(s);
  if (h->resident) {
    const size_t qbytes = (size_t)cfg->num_envs * h->res_R * stride * sizeof(unsigned long long);
    CREATE_CHK(hipMalloc((void**)&h->res_q1, qbytes));
    CREATE_CHK(hipMemsetAsync(h->res_q1, 0, qbytes, h->stream));
    // cells and weights of the q1 positions travel with it where a launch is latency, not traffic: a handful of environments
    // (20 bytes per particle each way: 256 environments would spend 8 us on them), kernels that carry them (pic_resident.h: kHandCarry)
    if (!h->res_lean && h->res_ppt <= 10 && h->esz == 8 && cfg->num_envs <= 32) {
      const size_t cbytes = (size_t)cfg->num_envs * h->res_nw * 64 * h->res_ppt * (sizeof(int) + (cfg->interpol == PIC_TSC ? 4 : 2) * h->esz);
      CREATE_CHK(hipMalloc(&h->res_carry, cbytes));
      CREATE_CHK(hipMemsetAsync(h->res_carry, 0, cbytes, h->stream));
    }
  }
  CREATE_CHK(hipMalloc((void**)&h->ke_part, (size_t)cfg->num_envs * h->nblk * sizeof(double)));
  CREATE_CHK(hipMemsetAsync(h->ke_part, 0, (size_t)cfg->num_envs * h->nblk * sizeof(double), h->stream));
  double** grids[] = {&h->n, &h->E_mesh, &h->phi, &h->ext, &h->ext2, &h->probe_ext, &h->aux_n, &h->aux_E, &h->aux_phi};
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
  if (gbytes <= ((size_t)256 << 10)) CREATE_CHK(hipHostMalloc((void**)&h->h_fields, 3 * gbytes, hipHostMallocDefault));
  CREATE_CHK(hipHostMalloc((void**)&h->h_probe_pe, sbytes, hipHostMallocDefault));
  CREATE_CHK(hipMalloc((void**)&h->aux_pe, sbytes));
  CREATE_CHK(hipMemsetAsync(h->aux_pe, 0, sbytes, h->stream));
  CREATE_CHK(hipMalloc((void**)&h->bad, sizeof(unsigned long long)));
  CREATE_CHK(hipMemsetAsync(h->bad, 0, sizeof(unsigned long long), h->stream));
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
  void* bufs[] = {h->x, h->scratch, h->stage, h->ring, h->ke_part, h->n, h->E_mesh, h->phi, h->ext, h->ext2, h->probe_ext,
                  h->basis, h->act, h->modes, h->aux_n, h->aux_E, h->aux_pe, h->aux_phi, h->KE, h->bad, h->tw, h->traj, h->res_q1, h->res_carry};
  for (void* b : bufs)
    if (b) hipFree(b);
  if (h->v_separate && h->v) hipFree(h->v);
  if (h->h_scal) hipHostFree(h->h_scal);
  if (h->h_part) hipHostFree(h->h_part);
  if (h->h_fields) hipHostFree(h->h_fields);
  if (h->h_probe_pe) hipHostFree(h->h_probe_pe);
  if (h->own_stream) hipStreamDestroy(h->own_stream);
  delete h;
  return PIC_OK;
}

static int switch_stream(pic_handle* h, hipStream_t next) {
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->stream));      // drain the old stream: later work must see its results
  prof_drain(h);
  h->stream = next;
  return PIC_OK;
}

int pic_set_stream(pic_handle* h, void* hip_stream) {
  if (!h) return PIC_EINVAL;
  return switch_stream(h, static_cast<hipStream_t>(hip_stream));      // NULL is a stream too: the device's default stream
}

int pic_own_stream(pic_handle* h) {
  if (!h) return PIC_EINVAL;
  return switch_stream(h, h->own_stream);
}

int pic_schedule(pic_handle* h) { return h ? (h->resident ? 1 : 0) : PIC_EINVAL; }

int pic_placement_info(pic_handle* h, int* candidates, double* kept_gbytes_per_s, double* slowest_gbytes_per_s,
                       double* seconds) {
  if (!h) return PIC_EINVAL;
  if (candidates) *candidates = h->place.pairs_timed > 0 ? h->place.pairs_timed : 1;
  if (kept_gbytes_per_s) *kept_gbytes_per_s = h->place.kept_gbytes_per_s;
  if (slowest_gbytes_per_s) *slowest_gbytes_per_s = h->place.slowest_gbytes_per_s;
  if (seconds) *seconds = h->place.seconds;
  return PIC_OK;
}

int pic_placement_stats(pic_handle* h, pic_placement* out) {
  if (!h || !out) return PIC_EINVAL;
  *out = h->place;
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
  drop_cached_deposits(h);      // first: an upload that fails half way has still changed x, and no cached deposit may outlive that
  int rc = upload_positions(h, h->x, x, mem_kind);
  if (rc) return rc;
  rc = upload(h, h->v, v, mem_kind);
  if (rc) return rc;
  if (mem_kind == PIC_HOST) HIPCHK(h, hipStreamSynchronize(h->stream));
  h->has_state = true;
  return PIC_OK;
}

int pic_invalidate(pic_handle* h) {
  if (!h) return PIC_EINVAL;
  drop_cached_deposits(h);
  return PIC_OK;
}

int pic_refresh(pic_handle* h) {
  if (!h) return PIC_EINVAL;
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_refresh: no particles loaded (call pic_reset first)");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  return refresh_fields(h);
}

int pic_reset(pic_handle* h, const void* x0, const void* v0, int mem_kind) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  resume_placement(h);
  HIPCHK(h, hipMemsetAsync(h->bad, 0, sizeof(unsigned long long), h->stream));
  int rc = pic_set_particles(h, x0, v0, mem_kind);
  if (rc) return rc;
  return refresh_fields(h);
}

// the sweeps of one environment step; `upto`: 1 = through sweep B, 2 = through C, 3 = whole step.  from: first
// stage to run (1, 2, 3).  Each force evaluation takes `ctl` (may differ per stage in the staged entry point).
// another_step_follows (the steps of one call but the last): this step ends with sweep D2, which leaves the deposit of its final
// positions to the next step's sweep B2, and its post-step solve is not launched: that step's sweep C carries it in one extra
// workgroup per environment (its results -- n, E_mesh, phi, the energies -- are read by nothing inside the call; the last step
// ends with the full sweep D and a solve launch of its own as ever).  Every refresh is made, from the same integer sums.
static void run_stages(pic_handle* h, int from, int upto, const Control& ctl, bool another_step_follows = false,
                       bool field_ready = false, const double* next_act = nullptr) {
  const double* c = h->cs;
  const double* d = h->ds;
  // A whole step under actuator coefficients builds the actuator's field once per environment, not in every workgroup of every
  // sweep (pic_sweep.h: SweepIO::ext_out).  h->ext / h->ext2 -- idle in such a call: they stage host fields -- take turns:
  // `mine` holds this step's field, written by sweep B of a call's first step (which builds it in every workgroup: it cannot
  // wait for anyone) or by the previous step's sweep D (field_ready); sweep D writes the next step's from next_act into the
  // other one.  Same doubles, same sums: config 3 as specified pays 0.5 % for its control instead of 2.6 %.
  const bool share_field = from == 1 && upto == 3 && ctl.act != nullptr;
  double* mine = h->ext_turn ? h->ext2 : h->ext;
  double* other = h->ext_turn ? h->ext : h->ext2;
  Control first = ctl, later = ctl;
  if (share_field) {
    later.act = nullptr; later.ext = mine;
    if (field_ready) first = later;
  }
  for (int st = from; st <= upto; ++st) {
    if (st == 1) {
      if (h->q_slot < 0) {          // particles were loaded without a refresh: deposit q1 = x + (c1 v) dt now
        h->q_slot = ring_take_clean(h);
        launch_sweep(h, ST_A, h->x, h->v, 0.0, c[0], 0.0, -1, Control{}, ring_row(h, h->q_slot), nullptr);
      }
      const int x1 = ring_take_clean(h);
      if (h->refresh_pending) {
        // the step before ended with sweep D2: this sweep B deposits the positions it reads (that step's x') into row r, and sweep C
        // carries the post-step solve of that step from it
        const int r = ring_take_clean(h);
        launch_sweep(h, ST_B2, h->x, h->v, c[0], c[1], d[1], h->q_slot, first, ring_row(h, x1), ring_row(h, r), -1,
                     share_field && !field_ready ? mine : nullptr);
        h->refresh_pending = false;
        h->post_slot = r;
      } else {
        launch_sweep(h, ST_B, h->x, h->v, c[0], c[1], d[1], h->q_slot, first, ring_row(h, x1), nullptr, -1,
                     share_field && !field_ready ? mine : nullptr);
      }
      ring_retire(h, h->q_slot);
      h->q_slot = -1;
      h->stage_slot = x1;
    } else if (st == 2) {
      const int x2 = ring_take_clean(h);
      launch_sweep(h, ST_C, h->x, h->v, 0.0, c[2], d[2], h->stage_slot, later, ring_row(h, x2), nullptr, h->post_slot);
      ring_retire(h, h->post_slot);
      h->post_slot = -1;
      ring_retire(h, h->stage_slot);
      h->stage_slot = x2;
    } else {
      const bool hand_on = share_field && next_act != nullptr;
      if (another_step_follows && h->light_inner_steps) {
        // an inner step of a call: nothing can see its post-step fields before the next step has started, so the deposit they come
        // from is left to that step's sweep B2 (sweep D is the one sweep bound by VALU issue: 336 -> 321 us at config 2, B 320 -> 322)
        const int qn = ring_take_clean(h);
        launch_sweep(h, ST_D2, h->x, h->v, 0.0, c[3], d[3], h->stage_slot, later, nullptr, ring_row(h, qn), -1,
                     hand_on ? other : nullptr, hand_on ? next_act : nullptr);
        h->refresh_pending = true;
        h->post_hist_row = h->hist_row;
        h->q_slot = qn;
      } else {
        const int f = ring_take_clean(h), qn = ring_take_clean(h);
        launch_sweep(h, ST_D, h->x, h->v, 0.0, c[3], d[3], h->stage_slot, later, ring_row(h, f), ring_row(h, qn), -1,
                     hand_on ? other : nullptr, hand_on ? next_act : nullptr);
        // (a small state's inner step: the full sweep D, its solve still not a launch -- it rides with the next step's sweep C)
        if (another_step_follows) { h->post_slot = f; h->post_hist_row = h->hist_row; }
        else launch_final_solve(h, f);
        h->q_slot = qn;
      }
      if (hand_on) h->ext_turn ^= 1;
      ring_retire(h, h->stage_slot);
      h->stage_slot = -1;
    }
  }
}

// host -> device staging of a call's inputs.  Small per-step inputs have buffers of their own (h->ext, h->act); whole
// trajectories go through h->traj, grown on demand (growing drains the stream first).
static int ensure_traj(pic_handle* h, size_t bytes) {
  if (bytes <= h->traj_bytes) return PIC_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->traj) { hipFree(h->traj); h->traj = nullptr; h->traj_bytes = 0; }
  if (hipMalloc(&h->traj, bytes) != hipSuccess) return fail(h, PIC_ENOMEM, "trajectory staging buffer");
  h->traj_bytes = bytes;
  return PIC_OK;
}

static int stage_ext(pic_handle* h, const double* E_ext, int mem_kind, const double** ext) {
  *ext = nullptr;
  if (!E_ext) return PIC_OK;
  *ext = E_ext;
  if (mem_kind == PIC_HOST) {
    HIPCHK(h, hipMemcpyAsync(h->ext, E_ext, (size_t)h->cfg.num_envs * h->cfg.Ng * sizeof(double), hipMemcpyHostToDevice,
                             h->stream));
    *ext = h->ext;
  }
  return PIC_OK;
}

static int ensure_twiddle(pic_handle* h, int rows) {
  if (rows <= h->tw_rows) return PIC_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->tw) { hipFree(h->tw); h->tw = nullptr; h->tw_rows = 0; }
  HIPCHK(h, hipMalloc((void**)&h->tw, (size_t)2 * rows * h->cfg.Ng * sizeof(double)));
  hipLaunchKernelGGL(twiddle_kernel, dim3((h->cfg.Ng + BLOCK - 1) / BLOCK, rows), dim3(BLOCK), 0, h->stream, h->tw, h->cfg.Ng, rows);
  HIPCHK(h, hipGetLastError());
  h->tw_rows = rows;
  return PIC_OK;
}

// nsteps x PIC.update_state under `sc`, all launches enqueued, no host synchronisation.  hist: device [nsteps][3][env] record of
// the energies, or null; snap (resident schedule only): device record of the particles.
static int advance(pic_handle* h, const StepControl& sc, int nsteps, double* hist, void* snap = nullptr) {
  if (nsteps <= 0) return PIC_OK;
  const int E = h->cfg.num_envs;
  if (h->resident) {
    launch_resident(h, sc, nsteps, hist, snap);
    ring_retire(h, h->q_slot);       // the ring's q1 deposit belongs to the particles before these steps
    ring_retire(h, h->stage_slot);
    h->q_slot = h->stage_slot = -1;
    // the q1 mesh and the carried cells the kernel leaves behind are valid only if the launch went out: after a failed one the
    // next call must deposit q1 itself instead of taking over an unwritten block
    const hipError_t e = launch_status(h);
    h->res_q1_valid = e == hipSuccess;
    h->res_carry_valid = e == hipSuccess && h->res_carry != nullptr;
    HIPCHK(h, e);
    return PIC_OK;
  }
  const size_t act_row = (size_t)E * 2 * sc.ctl.M;
  h->inline_act = sc.inline_n > 0 ? &sc.inline_act : nullptr;      // (launch_sweep: the held action inside the sweeps' arguments)
  for (int s = 0; s < nsteps; ++s) {
    Control ctl = sc.ctl;
    if (ctl.ext) ctl.ext += (size_t)s * sc.ext_step;
    if (ctl.act) ctl.act += (size_t)s * sc.act_step;
    h->hist_row = hist ? hist + (size_t)s * 3 * E : nullptr;
    bool rides = s + 1 < nsteps;     // the post-step solve rides with the next step's sweep B
    if (sc.fb.M > 0) {
      // The action of step s is the feedback law's on the field step s-1 left: the post-step solve is on the critical path
      // (a launch of its own that also computes the next action); before the first step a small kernel does it.
      Feedback fb = sc.fb;
      fb.act_out = h->act;
      if (s == 0) {
        hipLaunchKernelGGL(feedback_kernel, dim3(E), dim3(BLOCK), 0, h->stream, h->E_mesh, fb, h->cfg.Ng);
      }
      ctl.act = h->act;
      ctl.ext = nullptr;
      rides = false;
      h->fb = Feedback{};
      if (s + 1 < nsteps) {
        h->fb = fb;
        if (fb.act_hist) h->fb.act_hist = fb.act_hist + (size_t)(s + 1) * act_row;
      }
    }
    // (the actuator's field of step s: built by sweep B of step 0, after that by the previous step's sweep D -- or, under a
    // held action, still the one step 0 left; the feedback law's action exists only after the post-step solve: every step builds)
    const bool rollout = sc.fb.M == 0 && sc.ctl.act != nullptr;
    const bool held = rollout && sc.act_step == 0;
    const double* next_act = (rollout && !held && s + 1 < nsteps) ? ctl.act + sc.act_step : nullptr;
    run_stages(h, 1, 3, ctl, rides, rollout && s > 0, next_act);
  }
  h->hist_row = h->post_hist_row = nullptr;
  h->fb = Feedback{};
  h->inline_act = nullptr;
  HIPCHK(h, launch_status(h));
  return PIC_OK;
}

int pic_step_stage(pic_handle* h, int stage, const double* E_ext, int mem_kind) {
  if (!h) return PIC_EINVAL;
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_step_stage: call pic_reset first");
  if (stage < 1 || stage > 3 || stage != h->mid_stage + 1)
    return fail(h, PIC_ESTATE, "pic_step_stage: stages run in the order 1, 2, 3");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  Control ctl{};
  int rc = stage_ext(h, E_ext, mem_kind, &ctl.ext);
  if (rc) return rc;
  h->res_q1_valid = false;           // (a resident handle steps by sweeps here: its carried q1 mesh goes stale)
  run_stages(h, stage, stage, ctl);
  h->mid_stage = stage == 3 ? 0 : stage;
  HIPCHK(h, hipGetLastError());
  return PIC_OK;
}

static int check_steppable(pic_handle* h, int nsteps, const char* who) {
  if (!h->has_state) return fail(h, PIC_ESTATE, std::string(who) + ": call pic_reset first");
  if (nsteps < 0) return fail(h, PIC_EINVAL, std::string(who) + ": nsteps < 0");
  if (h->mid_stage) return fail(h, PIC_ESTATE, std::string(who) + ": a staged step is in progress (finish pic_step_stage 1..3)");
  return PIC_OK;
}

int pic_step(pic_handle* h, const double* E_ext, int mem_kind, int nsteps) {
  if (!h) return PIC_EINVAL;
  int rc = check_steppable(h, nsteps, "pic_step");
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  StepControl sc;
  rc = stage_ext(h, E_ext, mem_kind, &sc.ctl.ext);
  if (rc) return rc;
  return advance(h, sc, nsteps, nullptr);
}

// Runs `sc` for nsteps steps with the energies (hist, may be null) and / or the particles (snap, may be null) of every step
// kept on the device and read back once at the end; act_out (may be null): host [nsteps][env][2M] record of the feedback
// law's actions.  Returns after the read-backs, or -- nothing to read back -- without waiting for the device.
static int step_recording(pic_handle* h, StepControl sc, int nsteps, double* hist, void* snap, double* act_out, const char* who) {
  if (nsteps == 0) return PIC_OK;
  const int E = h->cfg.num_envs;
  const size_t hbytes = (size_t)nsteps * 3 * E * sizeof(double);
  const size_t sbytes = (size_t)nsteps * 2 * E * (size_t)h->cfg.N * h->esz;
  const size_t abytes = (size_t)nsteps * E * 2 * sc.fb.M * sizeof(double);
  double* dh = nullptr;
  double* da = nullptr;
  void* ds = nullptr;
  auto release = [&]() { if (dh) hipFree(dh); if (da) hipFree(da); if (ds) hipFree(ds); };
  if (hist && hipMalloc((void**)&dh, hbytes) != hipSuccess) return fail(h, PIC_ENOMEM, std::string(who) + ": history buffer");
  if (act_out && sc.fb.M > 0 && hipMalloc((void**)&da, abytes) != hipSuccess) { release(); return fail(h, PIC_ENOMEM, std::string(who) + ": action record"); }
  if (snap && hipMalloc(&ds, sbytes) != hipSuccess) {
    release();
    return fail(h, PIC_ENOMEM, std::string(who) + ": the snapshots of all steps do not fit on the device; record fewer steps per call");
  }
  sc.fb.act_hist = da;
  int rc = PIC_OK;
  if (!ds || h->resident) {
    rc = advance(h, sc, nsteps, dh, ds);      // energies: every post-step solve records its own entry; resident: the kernel records all
  } else {
    // particle snapshots on the streaming schedule: step by step, a copy kernel after each
    const dim3 grid = aux_grid(h, E);
    for (int s = 0; s < nsteps && rc == PIC_OK; ++s) {
      StepControl one = sc;
      if (one.ctl.ext) one.ctl.ext += (size_t)s * sc.ext_step;
      if (one.ctl.act) one.ctl.act += (size_t)s * sc.act_step;
      if (one.fb.act_hist) one.fb.act_hist += (size_t)s * E * 2 * sc.fb.M;
      rc = advance(h, one, 1, dh ? dh + (size_t)s * 3 * E : nullptr);
      if (rc != PIC_OK) break;
      if (h->fmt == FMT_F64)
        hipLaunchKernelGGL(record_particles_kernel<PosF64>, grid, dim3(BLOCK), 0, h->stream, (const double*)h->x,
                           (const double*)h->v, (double*)ds, s, h->cfg.N, h->ld, h->cfg.L);
      else if (h->fmt == FMT_F32)
        hipLaunchKernelGGL(record_particles_kernel<PosF32>, grid, dim3(BLOCK), 0, h->stream, (const float*)h->x,
                           (const float*)h->v, (float*)ds, s, h->cfg.N, h->ld, h->cfg.L);
      else
        hipLaunchKernelGGL(record_particles_kernel<PosU32>, grid, dim3(BLOCK), 0, h->stream, (const unsigned*)h->x,
                           (const float*)h->v, (float*)ds, s, h->cfg.N, h->ld, h->cfg.L);
    }
  }
  hipError_t e = hipGetLastError();
  if (!dh && !da && !ds) {
    if (rc != PIC_OK) return rc;
    if (e != hipSuccess) return fail(h, PIC_EHIP, std::string(who) + ": " + hipGetErrorString(e));
    return PIC_OK;
  }
  if (rc == PIC_OK && e == hipSuccess && dh) e = hipMemcpyAsync(hist, dh, hbytes, hipMemcpyDeviceToHost, h->stream);
  if (rc == PIC_OK && e == hipSuccess && da) e = hipMemcpyAsync(act_out, da, abytes, hipMemcpyDeviceToHost, h->stream);
  if (rc == PIC_OK && e == hipSuccess && ds) e = hipMemcpyAsync(snap, ds, sbytes, hipMemcpyDeviceToHost, h->stream);
  hipError_t e2 = hipStreamSynchronize(h->stream);
  release();
  if (rc != PIC_OK) return rc;
  if (e != hipSuccess || e2 != hipSuccess)
    return fail(h, PIC_EHIP, std::string(who) + ": " + hipGetErrorString(e != hipSuccess ? e : e2));
  return PIC_OK;
}

int pic_step_history(pic_handle* h, const double* E_ext, int mem_kind, int nsteps, double* hist) {
  if (!h || !hist) return fail(h, PIC_EINVAL, "pic_step_history: null argument");
  int rc = check_steppable(h, nsteps, "pic_step_history");
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  StepControl sc;
  rc = stage_ext(h, E_ext, mem_kind, &sc.ctl.ext);
  if (rc) return rc;
  return step_recording(h, sc, nsteps, hist, nullptr, nullptr, "pic_step_history");
}

int pic_step_snapshots(pic_handle* h, const double* E_ext, int mem_kind, int nsteps, void* snap, double* hist) {
  if (!h || !snap) return fail(h, PIC_EINVAL, "pic_step_snapshots: null argument");
  int rc = check_steppable(h, nsteps, "pic_step_snapshots");
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  StepControl sc;
  rc = stage_ext(h, E_ext, mem_kind, &sc.ctl.ext);
  if (rc) return rc;
  return step_recording(h, sc, nsteps, hist, snap, nullptr, "pic_step_snapshots");
}

// a per-step input trajectory [nsteps][row] (host or device) -> device pointer
static int stage_traj(pic_handle* h, const double* src, int mem_kind, size_t row_elems_, int nsteps, const double** dev) {
  *dev = src;
  if (mem_kind != PIC_HOST) return PIC_OK;
  const size_t bytes = (size_t)nsteps * row_elems_ * sizeof(double);
  int rc = ensure_traj(h, bytes);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->traj, src, bytes, hipMemcpyHostToDevice, h->stream));
  *dev = static_cast<const double*>(h->traj);
  return PIC_OK;
}

int pic_step_ext_traj(pic_handle* h, const double* E_ext_traj, int mem_kind, int nsteps, double* hist, void* snap) {
  if (!h || !E_ext_traj) return fail(h, PIC_EINVAL, "pic_step_ext_traj: null argument");
  int rc = check_steppable(h, nsteps, "pic_step_ext_traj");
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  StepControl sc;
  sc.ext_step = (long long)h->cfg.num_envs * h->cfg.Ng;
  rc = stage_traj(h, E_ext_traj, mem_kind, (size_t)sc.ext_step, nsteps, &sc.ctl.ext);
  if (rc) return rc;
  return step_recording(h, sc, nsteps, hist, snap, nullptr, "pic_step_ext_traj");
}

// x | v rows (and, with scalars, KE | PE | PE_reward) of float-position handles into the pinned staging buffers; the caller
// synchronises.  Up to kTinyState bytes a kernel of the handle's stream writes them (a copy command costs 5-7 us of launch
// latency behind a 20 us step, a kernel that stores to pinned memory 2.6: profiles/d2h_probe.hip); larger states go by copy
// commands (stores over PCIe run at ~9 GB/s from a kernel, the copy engine at ~50).  The staging for x | v exists from
// pic_create on for states up to 4 MB and is made here, once, for states up to 64 MB (a vectorised host-side loop over tens
// of environments: 64 x N = 5000 read back in 0.4 instead of 1.4 ms).
constexpr size_t kTinyState = (size_t)512 << 10;
static bool ensure_part_staging(pic_handle* h) {
  const size_t total = 2 * (size_t)h->cfg.num_envs * h->ld * h->esz;      // rows as they lie on the device (padded to ld)
  if (!h->h_part && !h->h_part_refused && total <= ((size_t)64 << 20)) {
    if (hipHostMalloc(&h->h_part, total, hipHostMallocDefault) != hipSuccess) {
      h->h_part = nullptr;
      h->h_part_refused = true;
      (void)hipGetLastError();
    }
  }
  return h->h_part != nullptr;
}
// *pitch: bytes from one environment's row to the next in the staging buffer (x rows, then v rows)
static int enqueue_observe(pic_handle* h, bool scalars, size_t* pitch) {
  const int E = h->cfg.num_envs;
  const size_t row = (size_t)h->cfg.N * h->esz;
  *pitch = row;
  if (2 * row * E > kTinyState || 2 * E > 65535) {      // (the kernel below numbers environments in grid.y: at most 65535 rows)
    // the arrays as they lie on the device, padding included, in plain copies (a strided copy command runs at a seventh of the rate)
    const size_t block = (size_t)E * h->ld * h->esz;
    *pitch = (size_t)h->ld * h->esz;
    if (!h->v_separate) {
      HIPCHK(h, hipMemcpyAsync(h->h_part, h->x, 2 * block, hipMemcpyDeviceToHost, h->stream));
    } else {
      HIPCHK(h, hipMemcpyAsync(h->h_part, h->x, block, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipMemcpyAsync(static_cast<char*>(h->h_part) + block, h->v, block, hipMemcpyDeviceToHost, h->stream));
    }
    if (scalars) HIPCHK(h, hipMemcpyAsync(h->h_scal, h->KE, 3 * (size_t)E * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return PIC_OK;
  }
  dim3 grid((unsigned)std::min<long long>((h->cfg.N + BLOCK - 1) / BLOCK, 64), 2 * E);
  if (h->esz == 8)
    hipLaunchKernelGGL(observe_kernel<double>, grid, dim3(BLOCK), 0, h->stream, static_cast<const double*>(h->x),
                       static_cast<const double*>(h->v), (long long)h->cfg.N, (long long)h->ld, E, static_cast<double*>(h->h_part),
                       h->KE, scalars ? h->h_scal : nullptr);
  else
    hipLaunchKernelGGL(observe_kernel<float>, grid, dim3(BLOCK), 0, h->stream, static_cast<const float*>(h->x),
                       static_cast<const float*>(h->v), (long long)h->cfg.N, (long long)h->ld, E, static_cast<float*>(h->h_part),
                       h->KE, scalars ? h->h_scal : nullptr);
  HIPCHK(h, hipGetLastError());
  return PIC_OK;
}
// staging -> the caller's [num_envs][N] arrays
static void unpack_part(pic_handle* h, void* x, void* v, size_t pitch) {
  const size_t E = (size_t)h->cfg.num_envs, row = (size_t)h->cfg.N * h->esz;
  const char* sx = static_cast<const char*>(h->h_part);
  const char* sv = sx + E * pitch;
  if (pitch == row) {
    if (x) std::memcpy(x, sx, E * row);
    if (v) std::memcpy(v, sv, E * row);
    return;
  }
  for (size_t e = 0; e < E; ++e) {
    if (x) std::memcpy(static_cast<char*>(x) + e * row, sx + e * pitch, row);
    if (v) std::memcpy(static_cast<char*>(v) + e * row, sv + e * pitch, row);
  }
}

// Actuator coefficients of one call, given on the host, to where the step reads them.  A handful (one environment's action:
// <= kInlineDoubles) rides in the argument blocks of the kernels that use it -- the resident kernel's, or the three sweeps' --
// where a pageable host-to-device copy command, or a launch of its own, costs 5-6 us on the stream in front of the step.
static int stage_actions(pic_handle* h, const double* actions, StepControl& sc) {
  const int n = h->cfg.num_envs * 2 * h->act_modes;
  sc.ctl.act = h->act;
  if (n <= kInlineDoubles) {
    std::memcpy(sc.inline_act.v, actions, (size_t)n * sizeof(double));
    sc.inline_n = n;
    return PIC_OK;
  }
  HIPCHK(h, hipMemcpyAsync(h->act, actions, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return PIC_OK;
}

int pic_get_particles(pic_handle* h, void* x, void* v, int mem_kind) {
  if (!h) return PIC_EINVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  if (x && v && mem_kind == PIC_HOST && h->fmt != FMT_U32 && ensure_part_staging(h)) {
    // states up to 64 MB go through pinned staging (enqueue_observe)
    size_t pitch = 0;
    const int rc = enqueue_observe(h, false, &pitch);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unpack_part(h, x, v, pitch);
    return PIC_OK;
  }
  int rc = PIC_OK;
  if (x) rc = download_positions(h, x, h->x, mem_kind);
  if (!rc && v) rc = download(h, v, h->v, mem_kind);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PIC_OK;
}

int pic_device_ptrs(pic_handle* h, void** x, void** v, int64_t* ld, double** n, double** E_mesh, double** phi,
                    double** KE, double** PE, double** PE_reward) {
  if (!h) return PIC_EINVAL;
  if (x || v) h->place_state.ptrs_exposed = true;       // (from here on the search for a placement may not move v: resume_placement)
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
  if (h->h_fields) {                                  // small meshes: one kernel writes all three into pinned memory, one wait
    const long long count = (long long)h->cfg.num_envs * h->cfg.Ng;
    hipLaunchKernelGGL(fields_out_kernel, dim3((unsigned)std::min<long long>((count + BLOCK - 1) / BLOCK, 64), 3), dim3(BLOCK), 0,
                       h->stream, h->n, h->E_mesh, h->phi, h->h_fields, count);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n) std::memcpy(n, h->h_fields, gbytes);
    if (E_mesh) std::memcpy(E_mesh, h->h_fields + count, gbytes);
    if (phi) std::memcpy(phi, h->h_fields + 2 * count, gbytes);
    return PIC_OK;
  }
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
  launch_gather(h, h->x, h->E_mesh, dst);
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
  HIPCHK(h, hipMalloc((void**)&dj, 3 * N * sizeof(long long)));
  if (hipMalloc((void**)&dw, 3 * N * sizeof(double)) != hipSuccess) { hipFree(dj); return fail(h, PIC_ENOMEM, "pic_get_cic: hipMalloc"); }
  const char* xe = (const char*)h->x + (size_t)env * h->ld * h->esz;
  launch_shape_query(h, xe, 1, PIC_CIC, dj, dw);
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

// deposit of the positions in h->scratch into the probe accumulator, then one solve with `ext` added
// Deposit + solve of the positions at `xs` ([env][ld], device-readable: h->scratch, or pinned host memory) into the aux meshes.
// pe_out: where the solve leaves 0.5 sum(E^2) dx per environment (h->aux_pe, or pinned host memory).
static int probe_solve(pic_handle* h, const double* E_ext, bool want_phi, void* xs = nullptr, double* pe_out = nullptr) {
  const size_t gbytes = (size_t)h->cfg.num_envs * h->cfg.Ng * sizeof(double);
  const double* ext = nullptr;
  if (E_ext) {
    HIPCHK(h, hipMemcpyAsync(h->probe_ext, E_ext, gbytes, hipMemcpyHostToDevice, h->stream));
    ext = h->probe_ext;
  }
  if (!xs) xs = h->scratch;
  // (the solve zeroes the row behind its read: one command less per probe; a probe that failed half way leaves it unknown)
  if (!h->probe_row_clean) HIPCHK(h, hipMemsetAsync(h->probe_acc, 0, row_elems(h) * sizeof(acc_t), h->stream));
  h->probe_row_clean = false;
  launch_sweep(h, ST_PROBE, xs, xs, 0, 0, 0, -1, Control{}, h->probe_acc, nullptr);
  SolveIO o{};
  o.acc = h->probe_acc; o.acc_clear = h->probe_acc; o.out.ext = ext; o.n = h->aux_n; o.out.E = h->aux_E;
  o.out.PEr = pe_out ? pe_out : h->aux_pe;
  if (want_phi) o.out.phi = h->aux_phi;
  launch_solve(h, o);
  HIPCHK(h, hipGetLastError());
  h->probe_row_clean = true;
  return PIC_OK;
}

int pic_eval_field(pic_handle* h, const void* x, int mem_kind, const double* E_ext, double* n, double* E_mesh,
                   double* half_sum_E2_dx) {
  if (!h || !x) return fail(h, PIC_EINVAL, "pic_eval_field: null argument");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t gbytes = (size_t)h->cfg.num_envs * h->cfg.Ng * sizeof(double);
  const size_t E = (size_t)h->cfg.num_envs, row = (size_t)h->cfg.N * h->esz;
  // A small host state (Reward.compute_reward of a trainer that only changed its imports: reward.py:48-50, once per step): the
  // positions go into the pinned staging buffer with a host copy and the probe sweep reads them from there, the solve writes
  // the energy into pinned memory -- two kernels and one wait instead of seven commands (78 -> 3x us per call at N = 5000).
  if (mem_kind == PIC_HOST && h->fmt != FMT_U32 && E * row <= kTinyState / 2 && ensure_part_staging(h)) {
    const size_t pitch = (size_t)h->ld * h->esz;
    HIPCHK(h, hipStreamSynchronize(h->stream));          // (the staging buffer may still be the target of an earlier read-back)
    for (size_t e = 0; e < E; ++e) {
      char* dst = static_cast<char*>(h->h_part) + e * pitch;
      std::memcpy(dst, static_cast<const char*>(x) + e * row, row);
      std::memset(dst + row, 0, pitch - row);
    }
    int rc = probe_solve(h, E_ext, false, h->h_part, h->h_probe_pe);
    if (rc) return rc;
    if (n) HIPCHK(h, hipMemcpyAsync(n, h->aux_n, gbytes, hipMemcpyDeviceToHost, h->stream));
    if (E_mesh) HIPCHK(h, hipMemcpyAsync(E_mesh, h->aux_E, gbytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (half_sum_E2_dx) std::memcpy(half_sum_E2_dx, h->h_probe_pe, E * sizeof(double));
    return PIC_OK;
  }
  int rc = ensure_scratch(h);
  if (rc) return rc;
  rc = upload_positions(h, h->scratch, x, mem_kind);
  if (rc) return rc;
  rc = probe_solve(h, E_ext, false);
  if (rc) return rc;
  if (n) HIPCHK(h, hipMemcpyAsync(n, h->aux_n, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (E_mesh) HIPCHK(h, hipMemcpyAsync(E_mesh, h->aux_E, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (half_sum_E2_dx)
    HIPCHK(h, hipMemcpyAsync(half_sum_E2_dx, h->aux_pe, (size_t)h->cfg.num_envs * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return PIC_OK;
}

int pic_compute_E(pic_handle* h, const void* x, int mem_kind, const double* E_ext, void* E_part, void* phi_part,
                  double* n, double* E_mesh, double* phi_mesh, int64_t* idx, double* w) {
  if (!h || !x) return fail(h, PIC_EINVAL, "pic_compute_E: null argument");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int rc = ensure_scratch(h);
  if (rc) return rc;
  rc = upload_positions(h, h->scratch, x, mem_kind);
  if (rc) return rc;
  const int E_ = h->cfg.num_envs;
  const long long N = h->cfg.N;
  const size_t gbytes = (size_t)E_ * h->cfg.Ng * sizeof(double);
  rc = probe_solve(h, E_ext, true);
  if (rc) return rc;
  if (n) HIPCHK(h, hipMemcpyAsync(n, h->aux_n, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (E_mesh) HIPCHK(h, hipMemcpyAsync(E_mesh, h->aux_E, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (phi_mesh) HIPCHK(h, hipMemcpyAsync(phi_mesh, h->aux_phi, gbytes, hipMemcpyDeviceToHost, h->stream));

  // gathers at the particles and shape bookkeeping go through one temporary, sized for the larger of the two
  void* tmp = nullptr;
  const size_t part_bytes = (size_t)E_ * N * h->esz;
  const size_t shape_bytes = (size_t)E_ * 3 * N * 8;
  const bool want_shape = idx || w;
  if (E_part || phi_part || want_shape) {
    if (hipMalloc(&tmp, want_shape ? 2 * shape_bytes : part_bytes) != hipSuccess)
      return fail(h, PIC_ENOMEM, "pic_compute_E: hipMalloc of the gather buffer");
  }
  hipError_t e = hipSuccess;
  const double* meshes[2] = {h->aux_E, h->aux_phi};
  void* outs[2] = {E_part, phi_part};
  for (int k = 0; k < 2 && e == hipSuccess; ++k) {
    if (!outs[k]) continue;
    launch_gather(h, h->scratch, meshes[k], tmp);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(outs[k], tmp, part_bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);      // tmp is reused
  }
  if (want_shape && e == hipSuccess) {
    long long* dj = static_cast<long long*>(tmp);
    double* dw = reinterpret_cast<double*>(static_cast<char*>(tmp) + shape_bytes);
    launch_shape_query(h, h->scratch, E_, h->cfg.interpol, dj, dw);
    e = hipGetLastError();
    if (e == hipSuccess && idx) e = hipMemcpyAsync(idx, dj, shape_bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && w) e = hipMemcpyAsync(w, dw, shape_bytes, hipMemcpyDeviceToHost, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (tmp) hipFree(tmp);
  if (e != hipSuccess) return fail(h, PIC_EHIP, std::string("pic_compute_E: ") + hipGetErrorString(e));
  return PIC_OK;
}

int pic_solve_poisson(pic_handle* h, const double* rhs, double* phi, double* E_mesh) {
  if (!h || !rhs) return fail(h, PIC_EINVAL, "pic_solve_poisson: null argument");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t gbytes = (size_t)h->cfg.num_envs * h->cfg.Ng * sizeof(double);
  HIPCHK(h, hipMemcpyAsync(h->aux_n, rhs, gbytes, hipMemcpyHostToDevice, h->stream));
  SolveIO o{};
  o.rhs = h->aux_n; o.out.E = h->aux_E; o.out.phi = h->aux_phi;
  launch_solve(h, o);
  HIPCHK(h, hipGetLastError());
  if (phi) HIPCHK(h, hipMemcpyAsync(phi, h->aux_phi, gbytes, hipMemcpyDeviceToHost, h->stream));
  if (E_mesh) HIPCHK(h, hipMemcpyAsync(E_mesh, h->aux_E, gbytes, hipMemcpyDeviceToHost, h->stream));
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

static int actuator_control(pic_handle* h, StepControl& sc, const char* who) {
  if (!h->act_modes) return fail(h, PIC_ESTATE, std::string(who) + ": call pic_set_actuator first");
  sc.ctl.basis = h->basis;
  sc.ctl.M = h->act_modes;
  return PIC_OK;
}

int pic_step_actions(pic_handle* h, const double* actions, int mem_kind, int nsteps) {
  if (!h || !actions) return fail(h, PIC_EINVAL, "pic_step_actions: null argument");
  int rc = check_steppable(h, nsteps, "pic_step_actions");
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  StepControl sc;
  rc = actuator_control(h, sc, "pic_step_actions");
  if (rc) return rc;
  sc.ctl.act = actions;
  if (mem_kind == PIC_HOST) {
    rc = stage_actions(h, actions, sc);
    if (rc) return rc;
  }
  return advance(h, sc, nsteps, nullptr);
}

int pic_step_observe(pic_handle* h, const double* E_ext, const double* actions, int nsteps, void* x, void* v, double* KE,
                     double* PE, double* PE_reward) {
  if (!h) return PIC_EINVAL;
  if (E_ext && actions) return fail(h, PIC_EINVAL, "pic_step_observe: E_ext and actions are alternatives");
  int rc = check_steppable(h, nsteps, "pic_step_observe");
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  StepControl sc;
  if (actions) {
    rc = actuator_control(h, sc, "pic_step_observe");
    if (rc) return rc;
    rc = stage_actions(h, actions, sc);
    if (rc) return rc;
  } else {
    rc = stage_ext(h, E_ext, PIC_HOST, &sc.ctl.ext);
    if (rc) return rc;
  }
  // Everything the caller reads goes into pinned memory behind the step, then ONE wait.  Small states have pinned staging
  // (h_part).  A one-step call of the resident schedule needs nothing else: its kernel records the particles after the step
  // (the snapshot of PIC.simulate, positions in length units whatever their format) and the step's energies (the energy
  // history) -- both straight into the pinned buffers, whose layouts are those records' for one step.
  const size_t E = (size_t)h->cfg.num_envs, b = E * sizeof(double);
  const size_t row = (size_t)h->cfg.N * h->esz, half = row * E;
  const bool want_part = x || v;
  bool part_pinned = false;
  size_t pitch = row;
  if (want_part) ensure_part_staging(h);
  if (h->resident && nsteps == 1 && h->h_part && 2 * half <= kTinyState) {
    rc = advance(h, sc, 1, h->h_scal, want_part ? h->h_part : nullptr);
    if (rc) return rc;
    part_pinned = want_part;
  } else {
    rc = advance(h, sc, nsteps, nullptr);
    if (rc) return rc;
    if (want_part && h->h_part && h->fmt != FMT_U32) {
      rc = enqueue_observe(h, true, &pitch);
      if (rc) return rc;
      part_pinned = true;
    } else {
      HIPCHK(h, hipMemcpyAsync(h->h_scal, h->KE, 3 * b, hipMemcpyDeviceToHost, h->stream));
      if (x) rc = download_positions(h, x, h->x, PIC_HOST);
      if (!rc && v) rc = download(h, v, h->v, PIC_HOST);
      if (rc) return rc;
    }
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (part_pinned) unpack_part(h, x, v, pitch);
  if (KE) std::memcpy(KE, h->h_scal, b);
  if (PE) std::memcpy(PE, h->h_scal + E, b);
  if (PE_reward) std::memcpy(PE_reward, h->h_scal + 2 * E, b);
  return PIC_OK;
}

int pic_step_actions_traj(pic_handle* h, const double* actions, int mem_kind, int nsteps, double* hist) {
  if (!h || !actions) return fail(h, PIC_EINVAL, "pic_step_actions_traj: null argument");
  int rc = check_steppable(h, nsteps, "pic_step_actions_traj");
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  StepControl sc;
  rc = actuator_control(h, sc, "pic_step_actions_traj");
  if (rc) return rc;
  sc.act_step = (long long)h->cfg.num_envs * 2 * h->act_modes;
  rc = stage_traj(h, actions, mem_kind, (size_t)sc.act_step, nsteps, &sc.ctl.act);
  if (rc) return rc;
  return step_recording(h, sc, nsteps, hist, nullptr, nullptr, "pic_step_actions_traj");
}

int pic_step_feedback(pic_handle* h, int max_mode, int nsteps, double* actions_out, double* hist) {
  if (!h) return PIC_EINVAL;
  int rc = check_steppable(h, nsteps, "pic_step_feedback");
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  StepControl sc;
  rc = actuator_control(h, sc, "pic_step_feedback");
  if (rc) return rc;
  if (max_mode != h->act_modes || max_mode > kMaxFeedbackModes)
    return fail(h, PIC_EINVAL, "pic_step_feedback: max_mode must equal the actuator's (pic_set_actuator) and be at most 16");
  rc = ensure_twiddle(h, max_mode);
  if (rc) return rc;
  sc.fb.tw = h->tw; sc.fb.rows = h->tw_rows; sc.fb.M = max_mode;
  sc.fb.act_out = h->act;
  return step_recording(h, sc, nsteps, hist, nullptr, actions_out, "pic_step_feedback");
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
  int rc = ensure_twiddle(h, max_mode);
  if (rc) return rc;
  double* dre = h->modes;
  double* dim_ = h->modes + (size_t)h->cfg.num_envs * max_mode;
  hipLaunchKernelGGL(modes_kernel, dim3(max_mode, h->cfg.num_envs), dim3(BLOCK), 0, h->stream, h->E_mesh, h->tw, h->tw_rows,
                     dre, dim_, h->cfg.Ng, max_mode);
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
  resume_placement(h);
  const dim3 grid = aux_grid(h, h->cfg.num_envs, 2048);
  if (h->fmt == FMT_F64)
    hipLaunchKernelGGL(sample_kernel<PosF64>, grid, dim3(BLOCK), 0, h->stream, (double*)h->x, (double*)h->v, h->cfg.N,
                       h->ld, kind, a, v0, sigma, A, n_mode, h->cfg.L, (unsigned long long)seed, h->cfg.env_index_base);
  else if (h->fmt == FMT_F32)
    hipLaunchKernelGGL(sample_kernel<PosF32>, grid, dim3(BLOCK), 0, h->stream, (float*)h->x, (float*)h->v, h->cfg.N,
                       h->ld, kind, a, v0, sigma, A, n_mode, h->cfg.L, (unsigned long long)seed, h->cfg.env_index_base);
  else
    hipLaunchKernelGGL(sample_kernel<PosU32>, grid, dim3(BLOCK), 0, h->stream, (unsigned*)h->x, (float*)h->v, h->cfg.N,
                       h->ld, kind, a, v0, sigma, A, n_mode, h->cfg.L, (unsigned long long)seed, h->cfg.env_index_base);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemsetAsync(h->bad, 0, sizeof(unsigned long long), h->stream));
  h->has_state = true;
  return refresh_fields(h);      // a reset abandons an open staged step and any cached deposit
}

// counts[num_envs][nbins][nbins] of the current particles into a fresh device buffer (caller frees it)
static hipError_t phase_counts(pic_handle* h, int nbins, double vmin, double vmax, unsigned** out) {
  const size_t nb = (size_t)h->cfg.num_envs * nbins * nbins * sizeof(unsigned);
  unsigned* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, nb);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d, 0, nb, h->stream);
  const dim3 grid = aux_grid(h, h->cfg.num_envs, 2048);
  if (e == hipSuccess) {
    if (h->fmt == FMT_F64)
      hipLaunchKernelGGL(phase_hist_kernel<PosF64>, grid, dim3(BLOCK), 0, h->stream, (const double*)h->x,
                         (const double*)h->v, d, h->cfg.N, h->ld, nbins, h->cfg.L, vmin, vmax);
    else if (h->fmt == FMT_F32)
      hipLaunchKernelGGL(phase_hist_kernel<PosF32>, grid, dim3(BLOCK), 0, h->stream, (const float*)h->x,
                         (const float*)h->v, d, h->cfg.N, h->ld, nbins, h->cfg.L, vmin, vmax);
    else
      hipLaunchKernelGGL(phase_hist_kernel<PosU32>, grid, dim3(BLOCK), 0, h->stream, (const unsigned*)h->x,
                         (const float*)h->v, d, h->cfg.N, h->ld, nbins, h->cfg.L, vmin, vmax);
    e = hipGetLastError();
  }
  if (e != hipSuccess) { hipFree(d); return e; }
  *out = d;
  return hipSuccess;
}

int pic_phase_histogram(pic_handle* h, int nbins, double vmin, double vmax, uint32_t* counts) {
  if (!h || !counts || nbins < 1 || nbins > 4096 || !(vmax > vmin))
    return fail(h, PIC_EINVAL, "pic_phase_histogram: need counts, 1 <= nbins <= 4096, vmax > vmin");
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_phase_histogram: call pic_reset first");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t nb = (size_t)h->cfg.num_envs * nbins * nbins * sizeof(unsigned);
  unsigned* d = nullptr;
  hipError_t e = phase_counts(h, nbins, vmin, vmax, &d);
  if (e == hipSuccess) e = hipMemcpyAsync(counts, d, nb, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (d) hipFree(d);
  if (e != hipSuccess) return fail(h, PIC_EHIP, std::string("pic_phase_histogram: ") + hipGetErrorString(e));
  return PIC_OK;
}

int pic_phase_kl(pic_handle* h, int nbins, double vmin, double vmax, const double* feq, double* kl) {
  if (!h || !feq || !kl || nbins < 1 || nbins > 4096 || !(vmax > vmin))
    return fail(h, PIC_EINVAL, "pic_phase_kl: need feq, kl, 1 <= nbins <= 4096, vmax > vmin");
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_phase_kl: call pic_reset first");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int E = h->cfg.num_envs, nb2 = nbins * nbins;
  unsigned* d = nullptr;
  double* df = nullptr;
  hipError_t e = phase_counts(h, nbins, vmin, vmax, &d);
  if (e == hipSuccess) e = hipMalloc((void**)&df, ((size_t)nb2 + E) * sizeof(double));
  if (e == hipSuccess) e = hipMemcpyAsync(df, feq, (size_t)nb2 * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    const double dx = h->cfg.L / nbins, dv = (vmax - vmin) / nbins;
    const double norm = h->cfg.n0 / dx / dv / (double)h->cfg.N;                      // objective.py:12, left to right
    hipLaunchKernelGGL(phase_kl_kernel, dim3(E), dim3(BLOCK), 0, h->stream, d, df, df + nb2, nb2, norm, dx * dv);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(kl, df + nb2, (size_t)E * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (d) hipFree(d);
  if (df) hipFree(df);
  if (e != hipSuccess) return fail(h, PIC_EHIP, std::string("pic_phase_kl: ") + hipGetErrorString(e));
  return PIC_OK;
}

int pic_stream_probe(pic_handle* h, int repeats, double* gbytes_per_s) {
  if (!h || !gbytes_per_s || repeats < 1) return fail(h, PIC_EINVAL, "pic_stream_probe: bad argument");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const size_t pbytes = (size_t)h->cfg.num_envs * h->ld * h->esz;
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
  hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, h->stream, (double2*)a, (double2*)b, n2, chunk2, 1.0, 1);
  hipEventRecord(e0, h->stream);
  for (int r = 0; r < repeats; ++r)
    hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, h->stream, (double2*)a, (double2*)b, n2, chunk2, 1.0, r & 1);
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
  unsigned long long c = 0;
  HIPCHK(h, hipMemcpyAsync(&c, h->bad, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *count = (int64_t)c;
  return PIC_OK;
}

}  // extern "C"
