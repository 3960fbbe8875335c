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
// 7 launches for a lone step, 6 per step inside a multi-step call (the final solve shares a launch with the
// next step's first one), 4 for small problems (force solves folded into the sweep prologues); 3 read+write
// passes over the particles per step (96 B per particle-step in fp64).
//
// Arithmetic inside a sub-stage keeps the reference's operand order and is compiled with
// -ffp-contract=off so that fp64 results track NumPy to rounding (tests/ hold the bounds).
//
// Deposit: every workgroup owns LDS copies of its environment's mesh (one per wave, `R` copies, two
// sets in sweep D), accumulates with LDS atomics (ds_add_f64; for float32 particles one packed ds_add_u64
// per deposit, pic_device.h; ds_add_f32 is selectable but measured ~4x slower), then stores its partial mesh
// as one row of a slab [env][block][Ng] with plain
// coalesced stores.  The field-solve kernel sums the rows in a fixed order (no global atomics, no
// memset between sweeps), scales to a density and solves the periodic Poisson problem with two prefix
// scans.
//
// Compile-time switches (all off in the shipped build; results of each in profiles/experiments_r1.md):
// PIC_EXP_* are timing/diagnostic experiments, PIC_PIPE / PIC_TILES alternative loop forms.
//
// Files: pic_device.h (per-particle helpers, scans), pic_sweep.h (push sweeps), pic_solve.h (field solve),
// pic_aux.h (kernels off the step path); this file holds the handle, the launch schedule and the C ABI.

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

#include "pic_device.h"
#include "pic_sweep.h"
#include "pic_solve.h"
#include "pic_aux.h"


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
  double* aux_phi = nullptr;      // pic_compute_E / pic_solve_poisson: potential of the probe solve
  int mid_stage = 0;              // pic_step_stage: force evaluations of the current step already done (0 = between steps)
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
// rhs_rows > 0: the "slab" is a caller-supplied right-hand side of that many rows per environment, taken as it
// is (scale 1, n0 0) instead of a deposit to be turned into n - n0 (pic_solve_poisson)
void launch_solve(pic_handle* h, const Lane& ln, const SolveOut& o, const SolveOut* second = nullptr, int rhs_rows = 0) {
  SolveArgs a;
  a.env0 = ln.env0;
  a.N = h->cfg.N; a.Ng = h->cfg.Ng; a.nblk = h->nblk; a.L = h->cfg.L; a.dx = h->dx; a.n0 = h->cfg.n0;
  a.scale = h->scale; a.N_over_L = (double)h->cfg.N / h->cfg.L;
  if (rhs_rows > 0) { a.nblk = rhs_rows; a.scale = 1.0; a.n0 = 0.0; }
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

// mesh [env][Ng] gathered at the positions x [env][ld] with the handle's shape function -> out, dense [env][N]
void launch_gather(pic_handle* h, const void* x, const double* mesh, void* out) {
  long long gx = (h->cfg.N + BLOCK - 1) / BLOCK;
  if (gx > 1024) gx = 1024;
  dim3 grid((unsigned)gx, h->cfg.num_envs);
  const size_t lds = ((size_t)h->cfg.Ng + 2) * h->esz;
  const bool tsc = h->cfg.interpol == PIC_TSC;
  if (h->cfg.particle_dtype == PIC_F64) {
    if (tsc) hipLaunchKernelGGL((gather_E_kernel<double, PIC_TSC>), grid, dim3(BLOCK), lds, h->stream, (const double*)x, mesh, (double*)out, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
    else hipLaunchKernelGGL((gather_E_kernel<double, PIC_CIC>), grid, dim3(BLOCK), lds, h->stream, (const double*)x, mesh, (double*)out, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
  } else {
    if (tsc) hipLaunchKernelGGL((gather_E_kernel<float, PIC_TSC>), grid, dim3(BLOCK), lds, h->stream, (const float*)x, mesh, (float*)out, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
    else hipLaunchKernelGGL((gather_E_kernel<float, PIC_CIC>), grid, dim3(BLOCK), lds, h->stream, (const float*)x, mesh, (float*)out, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx);
  }
}

// indices and weights of `nenv` environments' worth of positions x [nenv][ld] -> idx, w [nenv][3][N]
void launch_shape_query(pic_handle* h, const void* x, int nenv, int shape, long long* idx, double* w) {
  long long gx = (h->cfg.N + BLOCK - 1) / BLOCK;
  if (gx > 1024) gx = 1024;
  dim3 grid((unsigned)gx, nenv);
  const bool tsc = shape == PIC_TSC;
  if (h->cfg.particle_dtype == PIC_F64) {
    if (tsc) hipLaunchKernelGGL((shape_query_kernel<double, PIC_TSC>), grid, dim3(BLOCK), 0, h->stream, (const double*)x, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx, idx, w);
    else hipLaunchKernelGGL((shape_query_kernel<double, PIC_CIC>), grid, dim3(BLOCK), 0, h->stream, (const double*)x, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx, idx, w);
  } else {
    if (tsc) hipLaunchKernelGGL((shape_query_kernel<float, PIC_TSC>), grid, dim3(BLOCK), 0, h->stream, (const float*)x, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx, idx, w);
    else hipLaunchKernelGGL((shape_query_kernel<float, PIC_CIC>), grid, dim3(BLOCK), 0, h->stream, (const float*)x, h->cfg.N, h->ld, h->cfg.Ng, h->cfg.L, h->dx, idx, w);
  }
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
  if (cfg->env_index_base < 0) return fail(nullptr, PIC_EINVAL, "pic_create: env_index_base < 0");
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
                  h->aux_n, h->aux_E, h->aux_pe, h->aux_phi, h->KE, h->bad};
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
  h->mid_stage = 0;
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

int pic_step_stage(pic_handle* h, int stage, const double* E_ext, int mem_kind) {
  if (!h) return PIC_EINVAL;
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_step_stage: call pic_reset first");
  if (stage < 1 || stage > 3 || stage != h->mid_stage + 1)
    return fail(h, PIC_ESTATE, "pic_step_stage: stages run in the order 1, 2, 3");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const double* ext = nullptr;
  if (E_ext) {
    ext = E_ext;
    if (mem_kind == PIC_HOST) {
      HIPCHK(h, hipMemcpyAsync(h->ext, E_ext, (size_t)h->cfg.num_envs * h->cfg.Ng * sizeof(double),
                               hipMemcpyHostToDevice, h->stream));
      ext = h->ext;
    }
  }
  const double* c = h->cs;
  const double* d = h->ds;
  Lane ln = whole(h);
  ln.parity = h->sweep_parity;
  SolveOut f;                      // force evaluation with THIS stage's external field
  f.ext = ext; f.Ef = h->Ef;
  if (stage == 1) {
    if (h->q1_ready) {
      f.slab = h->part2;           // q1 was deposited by the previous sweep D / reset
    } else {
      launch_sweep(h, ln, ST_A, h->x, h->v, 0.0, c[0], 0.0);
    }
    launch_solve(h, ln, f);
    launch_sweep(h, ln, ST_B, h->x, h->v, c[0], c[1], d[1]);
  } else if (stage == 2) {
    launch_solve(h, ln, f);
    launch_sweep(h, ln, ST_C, h->x, h->v, 0.0, c[2], d[2]);
  } else {
    launch_solve(h, ln, f);
    launch_sweep(h, ln, ST_D, h->x, h->v, 0.0, c[3], d[3]);
    SolveOut o;                    // post-step refresh: no external field (pic.py:114-117)
    o.ke_part = h->ke_part; o.n = h->n; o.E = h->E_mesh; o.phi = h->phi;
    o.KE = h->KE; o.PE = h->PE; o.PEr = h->PEr;
    launch_solve(h, ln, o);
    h->q1_ready = true;
  }
  h->sweep_parity = ln.parity;
  h->mid_stage = stage == 3 ? 0 : stage;
  HIPCHK(h, hipGetLastError());
  return PIC_OK;
}

int pic_step(pic_handle* h, const double* E_ext, int mem_kind, int nsteps) {
  if (!h) return PIC_EINVAL;
  if (!h->has_state) return fail(h, PIC_ESTATE, "pic_step: call pic_reset first");
  if (nsteps < 0) return fail(h, PIC_EINVAL, "pic_step: nsteps < 0");
  if (h->mid_stage) return fail(h, PIC_ESTATE, "pic_step: a staged step is in progress (finish pic_step_stage 1..3)");
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

int pic_step_history(pic_handle* h, const double* E_ext, int mem_kind, int nsteps, double* hist) {
  if (!h || !hist) return fail(h, PIC_EINVAL, "pic_step_history: null argument");
  if (nsteps < 0) return fail(h, PIC_EINVAL, "pic_step_history: nsteps < 0");
  if (nsteps == 0) return PIC_OK;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int E = h->cfg.num_envs;
  const size_t bytes = (size_t)nsteps * 3 * E * sizeof(double);
  double* dh = nullptr;
  HIPCHK(h, hipMalloc((void**)&dh, bytes));
  // the external field is uploaded once; the per-step calls then take it from the device
  const double* ext = E_ext;
  int kind = mem_kind;
  int rc = PIC_OK;
  if (E_ext && mem_kind == PIC_HOST) {
    hipError_t e = hipMemcpyAsync(h->ext, E_ext, (size_t)E * h->cfg.Ng * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { hipFree(dh); return fail(h, PIC_EHIP, std::string("pic_step_history: ") + hipGetErrorString(e)); }
    ext = h->ext;
    kind = PIC_DEVICE;
  }
  for (int s = 0; s < nsteps && rc == PIC_OK; ++s) {
    rc = pic_step(h, ext, kind, 1);
    if (rc == PIC_OK)
      hipLaunchKernelGGL(record_energies_kernel, dim3((E + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, h->stream, h->KE, h->PE,
                         h->PEr, dh, s, E);
  }
  hipError_t e = hipGetLastError();
  if (rc == PIC_OK && e == hipSuccess) e = hipMemcpyAsync(hist, dh, bytes, hipMemcpyDeviceToHost, h->stream);
  hipError_t e2 = hipStreamSynchronize(h->stream);
  hipFree(dh);
  if (rc != PIC_OK) return rc;
  if (e != hipSuccess || e2 != hipSuccess)
    return fail(h, PIC_EHIP, std::string("pic_step_history: ") + hipGetErrorString(e != hipSuccess ? e : e2));
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

int pic_compute_E(pic_handle* h, const void* x, int mem_kind, const double* E_ext, void* E_part, void* phi_part,
                  double* n, double* E_mesh, double* phi_mesh, int64_t* idx, double* w) {
  if (!h || !x) return fail(h, PIC_EINVAL, "pic_compute_E: null argument");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int rc = ensure_scratch(h);
  if (rc) return rc;
  rc = upload(h, h->scratch, x, mem_kind);
  if (rc) return rc;
  const int E_ = h->cfg.num_envs;
  const long long N = h->cfg.N;
  const size_t gbytes = (size_t)E_ * h->cfg.Ng * sizeof(double);
  if (!h->aux_phi) HIPCHK(h, hipMalloc((void**)&h->aux_phi, gbytes));
  const double* ext = nullptr;
  if (E_ext) {
    HIPCHK(h, hipMemcpyAsync(h->ext, E_ext, gbytes, hipMemcpyHostToDevice, h->stream));
    ext = h->ext;
  }
  Lane ln = whole(h);
  launch_sweep(h, ln, ST_PROBE, h->scratch, h->scratch, 0, 0, 0);
  SolveOut o;
  o.ext = ext; o.n = h->aux_n; o.E = h->aux_E; o.phi = h->aux_phi; o.PEr = h->aux_pe;
  launch_solve(h, ln, o);
  HIPCHK(h, hipGetLastError());
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
  if (!h->aux_phi) HIPCHK(h, hipMalloc((void**)&h->aux_phi, gbytes));
  // aux_n doubles as the one-row "slab" holding the right-hand side; the solve reads it before it writes n
  HIPCHK(h, hipMemcpyAsync(h->aux_n, rhs, gbytes, hipMemcpyHostToDevice, h->stream));
  Lane ln = whole(h);
  SolveOut o;
  o.slab = h->aux_n; o.E = h->aux_E; o.phi = h->aux_phi;
  launch_solve(h, ln, o, nullptr, 1);
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
                       h->ld, kind, a, v0, sigma, A, n_mode, h->cfg.L, (unsigned long long)seed, h->cfg.env_index_base);
  else
    hipLaunchKernelGGL(sample_kernel<float>, grid, dim3(BLOCK), 0, h->stream, (float*)h->x, (float*)h->v, h->cfg.N,
                       h->ld, kind, a, v0, sigma, A, n_mode, h->cfg.L, (unsigned long long)seed, h->cfg.env_index_base);
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
