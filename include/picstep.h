/*
 * picstep.h -- C ABI of libpicstep.so, the MI355X (gfx950) 1-D electrostatic PIC stepper.
 *
 * The reference (ZINZINBIN/Optimal-Control-1D-Electrostatic-Plasma) has no FFI layer: its
 * boundary is the Python duck type `PIC` (src/env/pic.py:11-223).  Each entry point below names
 * the reference interface it stands in for; the ctypes binding lives in
 * optimal-control-1d-electrostatic-plasma_amd/_abi.py and INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions: every function returns 0 on success or a negative PIC_E* code (text through
 * pic_last_error); no exception crosses the ABI; the library owns all device memory; host
 * buffers are caller-owned and copied.  One handle = one device = one HIP stream; calls on one
 * handle must be serialised by the caller, different handles are independent (one per GPU when
 * environments are sharded).  All entry points except pic_step / pic_reset (device inputs)
 * return after the stream has drained; pic_step is asynchronous -- call pic_sync or any getter.
 *
 * Particle arrays are [num_envs][ld] with ld >= N (ld from pic_device_ptrs); host copies are
 * dense [num_envs][N].  Mesh arrays are dense [num_envs][Ng] float64.
 */
#ifndef PICSTEP_H
#define PICSTEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PICSTEP_ABI_VERSION 4

enum { PIC_F64 = 0, PIC_F32 = 1 };           /* particle dtype (velocities; positions too unless fixed point) */
enum { PIC_POS_FLOAT = 0,                    /* positions stored in the particle dtype                         */
       PIC_POS_FIXED32 = 1 };                 /* positions as 32-bit fixed point, x = u L / 2^32 (float32 particles) */
enum { PIC_ACC_AUTO = 0,                     /* deposit accumulator: library's choice (PACKED for float32 CIC, else FIX64) */
       PIC_ACC_FIX64 = 1,                     /* 64-bit integers, weights rounded to 2^-fg: order-independent sums    */
       PIC_ACC_PACKED = 2,                    /* float32 particles, CIC: (count, sum of w_r) per cell in one word     */
       PIC_ACC_F64 = 3 };                     /* float64 particles: float64 running sums in LDS (ds_add_f64)          */
enum { PIC_CIC = 0, PIC_TSC = 1 };           /* src/env/interpolate.py:4 (CIC), :22 (TSC)     */
enum { PIC_PLACE_AUTO = 0,                   /* large states: look for x and v in two different regions of HBM (pic_placement_info) */
       PIC_PLACE_OFF = 1 };                   /* no search: x | v in one allocation                                                 */
enum { PIC_HOST = 0, PIC_DEVICE = 1 };       /* where a caller buffer lives                   */
enum { PIC_PLACED_NONE = 0,                  /* pic_placement.outcome: no search (small state or PIC_PLACE_OFF)                    */
       PIC_PLACED_FOUND = 1,                 /* the pair kept streams >= 10 % faster than the slowest pair seen                    */
       PIC_PLACED_PATIENCE = 2,              /* 42 GiB walked without an improvement: all alike, the best of them kept             */
       PIC_PLACED_TIMEOUT = 3,               /* the leg's 100 ms spent: the best pair seen so far kept; a reset may run another leg */
       PIC_PLACED_MEMORY = 4 };              /* a third of the free memory held (or an allocation failed): the best pair seen kept */

enum {
  PIC_OK = 0,
  PIC_EINVAL = -1,     /* bad argument / unsupported configuration   */
  PIC_EHIP = -2,       /* a HIP runtime call failed                  */
  PIC_ESTATE = -3,     /* call order (e.g. step before reset)        */
  PIC_ENOMEM = -4
};

/* Constructor arguments of PIC (src/env/pic.py:13-27) that matter to the step, plus batching.
 * dt is the value AFTER the CFL clamp of pic.py:71-73 (the host wrapper applies the clamp).
 * gamma is accepted for signature parity only: E_mesh does not depend on it. */
typedef struct pic_config {
  int64_t N;               /* particles per environment                                   */
  int32_t Ng;              /* mesh cells (N_mesh)                                         */
  int32_t num_envs;        /* independent environments batched on this device             */
  double  L;               /* box length                                                  */
  double  n0;              /* mean density                                                */
  double  dt;              /* time step (post-clamp)                                      */
  double  gamma;           /* unused by the scan solver                                   */
  int32_t particle_dtype;  /* PIC_F64 | PIC_F32                                           */
  int32_t accum_dtype;     /* PIC_ACC_*: how a workgroup accumulates its deposit in LDS.  Every choice ends in the
                              same global 64-bit fixed-point accumulators; FIX64 and PACKED are integer sums all
                              the way and make a step bitwise reproducible                        */
  int32_t interpol;        /* PIC_CIC | PIC_TSC                                           */
  int32_t device_id;       /* HIP device ordinal                                          */
  int32_t blocks_per_env;  /* 0 = choose the schedule: environments of up to 8192 particles are stepped RESIDENT (all
                              sub-stages and steps of a pic_step call inside one workgroup and one launch), larger ones
                              by streaming sweeps over a chosen number of workgroups; > 0 = streaming sweeps with this
                              many workgroups per environment; -1 = resident, or EINVAL where it does not apply.
                              Particles and fields do not depend on the choice, bit for bit                       */
  int32_t env_index_base;  /* global index of environment 0 of this handle (0 for a single handle): keys the device
                              sampler, so that a sharded ensemble does not depend on the number of ranks           */
  int32_t position_dtype;  /* PIC_POS_FLOAT | PIC_POS_FIXED32 (needs particle_dtype PIC_F32).  Fixed-point positions
                              are handed over and returned as float32 like any float32 particle array; on the device
                              they are uint32 (pic_device_ptrs' x)                                                  */
  int32_t placement;       /* PIC_PLACE_AUTO | PIC_PLACE_OFF: see pic_placement_info                                */
  int32_t placement_ms;    /* upper bound, in milliseconds, on one leg of the search for an (x, v) placement; 0 = the
                              default (100 ms, or forty steps' worth of the handle if that is more)                  */
} pic_config;

typedef struct pic_handle pic_handle;

/* PIC.__init__ (pic.py:13-61) minus sampling: allocates state for num_envs environments. */
int pic_create(const pic_config* cfg, pic_handle** out);
int pic_destroy(pic_handle* h);

/* PIC.initialize / reinit (pic.py:63-91) after the host has drawn x0, v0 and applied the velocity
 * perturbation: stores the particles and does update_density + update_E_field (pic.py:93-123).
 * x0, v0: [num_envs][N] of the particle dtype, host or device. */
int pic_reset(pic_handle* h, const void* x0, const void* v0, int mem_kind);

/* PIC.reinit with the sample drawn ON THE DEVICE: the distributions of src/env/dist.py (kind 0 =
 * TwoStream :27-102, halves at +v0 / -v0 with spread sigma; kind 1 = BumpOnTail :104-194, int(N/(1+a))
 * bulk particles from N(0,1) followed by the beam from N(v0, sigma) -- the index order high_indx relies
 * on), x uniform on [0, L), v truncated to [-10, 10], then the velocity perturbation
 * v *= 1 + A sin(2 pi n_mode x / L) (pic.py:68) and update_density + update_E_field.  Counter-based
 * Philox generator keyed by (seed, environment): reproducible, but NOT the reference's NumPy stream --
 * the host samplers of the Python layer keep that. */
int pic_reset_sampled(pic_handle* h, int kind, double a, double v0, double sigma, double A, int n_mode,
                      uint64_t seed);

/* nsteps x PIC.update_state(E_external) (pic.py:131-146): Yoshida-4 push
 * (src/env/integration.py:60-75), final wrap, density/field refresh, KE/PE reductions.
 * E_ext: NULL or [num_envs][Ng] float64 (held constant over the nsteps), host or device.
 * Asynchronous on the handle's stream. */
int pic_step(pic_handle* h, const double* E_ext, int mem_kind, int nsteps);

/* PIC.x / PIC.v / get_state (pic.py:165-167): copies of the particle arrays, dense [num_envs][N]. */
int pic_get_particles(pic_handle* h, void* x, void* v, int mem_kind);
/* Overwrite the particles without touching fields (checkpoint restore); follow with pic_refresh. */
int pic_set_particles(pic_handle* h, const void* x, const void* v, int mem_kind);
/* update_density + update_E_field on the current particles (pic.py:93-123). */
int pic_refresh(pic_handle* h);
/* The caller has written x or v through the device views of pic_device_ptrs (e.g. re-seeded finished
 * environments on the device).  The handle caches the next step's first deposit (taken by the last sweep
 * of the previous step); that cache no longer matches such particles.  pic_invalidate drops it, so that the
 * next pic_step re-deposits from the stored particles (one extra read of x, v); pic_refresh does the same and
 * also recomputes n / E_mesh / phi / energies.  One of the two MUST follow every external write. */
int pic_invalidate(pic_handle* h);

/* Zero-copy device views for torch: any pointer argument may be NULL. ld = leading dimension
 * (elements) of x and v; mesh arrays are dense. Valid until pic_destroy.  The views are writable; a write to
 * x or v must be followed by pic_invalidate or pic_refresh (see there). */
int pic_device_ptrs(pic_handle* h, void** x, void** v, int64_t* ld, double** n, double** E_mesh, double** phi,
                    double** KE, double** PE, double** PE_reward);

/* PIC.n, PIC.E_mesh, PIC.phi_mesh after a step (pic.py:103,116-117). phi is returned in the
 * mean-zero gauge (the reference's gauge is round-off). Host buffers, any may be NULL. */
int pic_get_fields(pic_handle* h, double* n, double* E_mesh, double* phi);

/* Per-environment energies of the current state: KE = 0.5*sum v^2 (src/env/util.py:144),
 * PE = 0.5*sum(E_mesh^2)*dx*N/L (util.py:129-130), PE_reward = 0.5*sum(E_mesh^2)*dx
 * (src/control/objective.py:33, the reward reduction). Host buffers [num_envs], any may be NULL. */
int pic_get_energies(pic_handle* h, double* KE, double* PE, double* PE_reward);

/* PIC.E (pic.py:120): E_mesh gathered at the particles, dense [num_envs][N], particle dtype. */
int pic_gather_E(pic_handle* h, void* E_particles, int mem_kind);

/* PIC.indx_l/indx_r/weight_l/weight_r (pic.py:104-107) of one environment: host buffers [N]
 * (int64 indices, float64 weights), any may be NULL. */
int pic_get_cic(pic_handle* h, int env, int64_t* indx_l, int64_t* indx_r, double* weight_l, double* weight_r);

/* compute_E on arbitrary positions (src/env/util.py:73-116), used by compute_electric_energy
 * (util.py:119-131) and estimate_electric_energy (objective.py:20-35): deposit x -> solve ->
 * E_mesh (+E_ext).  Does not modify the environments' state (probes have their own deposit accumulator and
 * E_ext staging, so they may also run between pic_step_stage calls).  x: [num_envs][N] particle dtype;
 * E_ext: NULL or [num_envs][Ng] host float64; outputs (host, [num_envs][Ng] / [num_envs], any
 * may be NULL): n, E_mesh (with E_ext added), half_sum_E2_dx = 0.5*sum(E_mesh^2)*dx. */
int pic_eval_field(pic_handle* h, const void* x, int mem_kind, const double* E_ext,
                   double* n, double* E_mesh, double* half_sum_E2_dx);

/* One environment step in three calls, each with its own external field: PIC.update_state_w_input_func
 * (pic.py:148-163), where the field is a function of the sub-stage state.  Stage k (1, 2, 3, in this order)
 * evaluates the force at q_k with E_ext (as in pic_step; NULL = none) and runs the sweep that kicks to p_k and
 * drifts to q_{k+1}; stage 3 also wraps x and refreshes n / E_mesh / phi / energies.  Between the calls
 * pic_get_particles returns the sub-stage state (q_{k+1} unwrapped, p_k) the caller's input function needs
 * (before stage 1: q_1 = x + (c_1 v) dt with c_1 = 0.5 / (2 - 2^(1/3)), formed by the caller).  pic_step and
 * pic_step_stage(1) are refused while a staged step is open; pic_reset / pic_set_particles abandon it. */
int pic_step_stage(pic_handle* h, int stage, const double* E_ext, int mem_kind);

/* nsteps x PIC.update_state with the energies of every step kept, i.e. the E / PE traces PIC.simulate
 * returns (pic.py:175-223) without its particle snapshots: hist, host [nsteps][3][num_envs] float64 =
 * KE, PE, PE_reward after each step (total energy = KE + PE).  E_ext as in pic_step, constant over the steps.
 * No host synchronisation between the steps; returns when the history has arrived. */
int pic_step_history(pic_handle* h, const double* E_ext, int mem_kind, int nsteps, double* hist);

/* The same with the particle snapshots PIC.simulate returns as well (pic.py:175-223): snap, host
 * [nsteps][2][num_envs][N] of the particle dtype = positions (wrapped, in length units whatever the position format)
 * and velocities after each step; hist as in pic_step_history, or NULL.  The snapshots stay on the device until the
 * end of the call (PIC_ENOMEM if nsteps of them do not fit: record in several calls); in the resident schedule the
 * kernel writes them from its registers, one launch for all steps. */
int pic_step_snapshots(pic_handle* h, const double* E_ext, int mem_kind, int nsteps, void* snap, double* hist);

/* compute_E with everything it can return (src/env/util.py:73-116, return_all=True) and the shape-function
 * bookkeeping of compute_n / CIC / TSC (util.py:48-70, src/env/interpolate.py:4-44), on arbitrary positions.
 * x: [num_envs][N] particle dtype (host or device); E_ext: NULL or host [num_envs][Ng] float64.  Host outputs,
 * any may be NULL: E_part, phi_part [num_envs][N] (particle dtype) = E_mesh (incl. E_ext) and phi_mesh gathered
 * at the particles; n, E_mesh, phi_mesh [num_envs][Ng] float64 (phi in the mean-zero gauge); idx (int64) and
 * w (float64) [num_envs][3][N]: rows indx_l, indx_r, 0 / weight_l, weight_r, 0 for CIC and l, m, r for TSC.
 * Does not modify the environments' state. */
int pic_compute_E(pic_handle* h, const void* x, int mem_kind, const double* E_ext, void* E_part, void* phi_part,
                  double* n, double* E_mesh, double* phi_mesh, int64_t* idx, double* w);

/* Gaussian_Elimination_Periodic on the 3-point periodic Laplacian (src/env/solve.py:27-53 as called from
 * util.py:99) + E_mesh = -grad @ phi (util.py:100): rhs host [num_envs][Ng] float64, summing to zero per
 * environment as n - n0 does (otherwise the periodic problem has no solution) -> phi (mean zero), E_mesh; host,
 * either may be NULL. */
int pic_solve_poisson(pic_handle* h, const double* rhs, double* phi, double* E_mesh);

/* Device-side actuator, E_field (src/control/actuator.py:4-63).  pic_set_actuator uploads the host
 * mirror's basis tables, basis_cos / basis_sin [Ng][max_mode] float64 (they carry the reference's
 * linspace(0, L, Ng) mesh).  pic_step_actions runs nsteps x update_state under E_ext = basis_cos @ a[:M] +
 * basis_sin @ a[M:] (actuator.py:54-63), a = actions [num_envs][2*max_mode] float64 (host or device), held for the
 * nsteps.  The field is built inside the field phase of the kernels that use it: a controlled step costs no launch and
 * no mesh-sized copy more than an uncontrolled one. */
int pic_set_actuator(pic_handle* h, int max_mode, const double* basis_cos, const double* basis_sin);
int pic_step_actions(pic_handle* h, const double* actions, int mem_kind, int nsteps);

/* A rollout with a NEW action every step in one call -- the inner loop of the trainers (src/control/rl/ddpg.py:421-468,
 * ppo.py:307-372, sac.py:328-396) once the actions are known, and PIC.simulate with an action trajectory:
 * actions [nsteps][num_envs][2*max_mode] float64 (host or device); step s runs under actions[s].  Resident schedule: one launch
 * for all steps; streaming: three launches per step, as pic_step(nsteps).  hist: NULL, or host [nsteps][3][num_envs] = KE, PE,
 * PE_reward after each step (the call then returns when it has arrived; with NULL it is asynchronous like pic_step). */
int pic_step_actions_traj(pic_handle* h, const double* actions, int mem_kind, int nsteps, double* hist);

/* The same with the fields given on the mesh: PIC.simulate(E_external_traj) (pic.py:175-223, E_external_traj[i] of step i).
 * E_ext_traj [nsteps][num_envs][Ng] float64 (host or device); hist as above; snap: NULL, or the particle snapshots of
 * pic_step_snapshots. */
int pic_step_ext_traj(pic_handle* h, const double* E_ext_traj, int mem_kind, int nsteps, double* hist, void* snap);

/* nsteps of the linear feedback loop of run_feedback.py:130-168 on the device: before every step the actuator coefficients are
 * set to (-Re E_m, +Im E_m), m = 1..max_mode, of the Fourier modes (src/interpret/spectrum.py:16) of the mesh field the
 * previous step left (the current E_mesh for the first step) -- compute_E_k_spectrum, E_field.update_E, E_field.compute_E and
 * PIC.update_state of one loop iteration without leaving the device.  max_mode must be the actuator's (pic_set_actuator) and
 * at most 16.  actions_out: NULL, or host [nsteps][num_envs][2*max_mode] = the coefficients each step ran under (cos half, sin
 * half: the reference's coeff_cos / coeff_sin lists); hist as above.  Bit for bit what the host loop pic_get_modes ->
 * pic_step_actions gives. */
int pic_step_feedback(pic_handle* h, int max_mode, int nsteps, double* actions_out, double* hist);

/* One iteration of a Gym-style loop in ONE call with ONE synchronisation (src/control/rl/ddpg.py:421-468, ppo.py, sac.py:
 * env.update_state(E) -> next_state = env.get_state() -> reward from the new state's electric energy): nsteps steps under
 * the given control -- E_ext [num_envs][Ng] on the mesh (util.py:102-103) or actions [num_envs][2*max_mode] actuator
 * coefficients (pic_step_actions), both host float64, at most one non-NULL -- then the particles (x, v: host [num_envs][N] in the
 * particle dtype, positions in length units) and the three energies (host [num_envs] each) of the state after them.  Any
 * output may be NULL.  Same results as pic_step / pic_step_actions followed by pic_get_particles and pic_get_energies,
 * which cost a synchronisation each (13-23 us of a 63 us iteration at the reference's N = 5000). */
int pic_step_observe(pic_handle* h, const double* E_ext, const double* actions, int nsteps, void* x, void* v,
                     double* KE, double* PE, double* PE_reward);

/* Rows 1..max_mode of compute_E_k_spectrum (src/interpret/spectrum.py:16) for the current E_mesh:
 * Ek[m] = fft(E_mesh)[m] / Ng * 2, re / im [num_envs][max_mode] float64 (host or device, any may be
 * NULL).  The feedback / behaviour-cloning action of run_feedback.py:133-135 and
 * src/control/rl/ddpg.py:369-371 is (-re, +im). */
int pic_get_modes(pic_handle* h, int max_mode, double* re, double* im, int mem_kind);

/* Phase-space histogram behind the KL diagnostic, estimate_f (src/control/objective.py:8-14):
 * counts[num_envs][nbins][nbins] (host, uint32) of the current particles on
 * np.histogram2d's bins for range [[0, L], [vmin, vmax]] -- same edge rules (edges lo + i*step,
 * last edge inclusive, out-of-range values dropped).  f = counts * n0 / dx / dv / N on the host. */
int pic_phase_histogram(pic_handle* h, int nbins, double vmin, double vmax, uint32_t* counts);

/* The KL cost built on it, Reward.compute_kl_divergence (src/control/rl/reward.py:43-46 -> estimate_KL_divergence,
 * objective.py:16-18), for every environment at once: kl[e] = sum_ij rel_entr(f_e[i][j], feq[i][j] + 1e-12) dx dv with
 * f_e = counts_e n0 / dx / dv / N, dx = L / nbins, dv = (vmax - vmin) / nbins; feq host [nbins][nbins] float64 (the target
 * density, estimate_f of the initial state), kl host [num_envs].  Histogram and reduction stay on the device. */
int pic_phase_kl(pic_handle* h, int nbins, double vmin, double vmax, const double* feq, double* kl);

/* Per-kernel timing with HIP events on the handle's stream (bench.py's roofline leg).
 * kinds: 0..3 = sweeps A..D, 4 = field solve. ms_sum / launches are arrays of 8. */
int pic_profile(pic_handle* h, int enable);
int pic_profile_read(pic_handle* h, double* ms_sum, int64_t* launches);

/* Streaming ceiling of this device for the sweeps' access shape (read x and v, write x and v, same
 * grid, no arithmetic): bytes moved per second in GB/s, averaged over `repeats` launches on
 * scratch arrays of the handle's particle footprint.  bench.py reports it next to the 8 TB/s spec. */
int pic_stream_probe(pic_handle* h, int repeats, double* gbytes_per_s);

/* Run all further work of this handle on the caller's HIP stream (a hipStream_t, e.g. torch's current stream; NULL
 * is the device's default stream, which is what torch uses unless told otherwise) instead of the handle's own;
 * pic_own_stream switches back.  The previous stream is drained first.  With device-pointer inputs/outputs (pic_step,
 * pic_step_actions, pic_get_modes, pic_device_ptrs views) a control loop then stays stream-ordered with the caller's
 * kernels and needs no host synchronisation. */
int pic_set_stream(pic_handle* h, void* hip_stream);
int pic_own_stream(pic_handle* h);

/* 1 if pic_step runs the resident schedule on this handle, 0 for streaming sweeps (see blocks_per_env). */
int pic_schedule(pic_handle* h);

/* Particle states of 256 MB and more: x and v are two allocations, and pic_create times a streaming pass over (x, candidate
 * block for v) for a series of candidate blocks: on MI355X two arrays stream together at 6.05 TB/s when they lie in different
 * 32 GiB regions of HBM and at 5.25 TB/s when they share one.  The search stops sixteen readings after the best pair
 * seen is 10 % faster than the slowest seen (keeping the best of all), after 42 GiB and sixteen blocks walked without an improvement
 * (more than a 32 GiB region), or when a third of the device's free memory is held; no absolute rate enters.  It runs in LEGS of at most 100 ms
 * (or forty steps' worth of the handle being placed, if that is more: 160 ms at N = 4e6 x 64, 410 ms at N = 1e7 x 128 float32):
 * pic_create runs one; while the search has ended only for lack of time (outcome PIC_PLACED_TIMEOUT: on a device whose memory is
 * handed out for the first time hipMalloc clears it at 1.3-6 ms per 512 MB block, and the first block that pairs well with x can be
 * 31 GiB away), pic_reset / pic_reset_sampled -- which replace the particles anyway -- run another leg each, up to four in all, and
 * may move v.  What an earlier leg has cleared and released comes back in microseconds, so every leg gets further.  Once
 * pic_device_ptrs has handed out the addresses of x and v, v stays where it is and no further leg runs.  Candidate blocks are
 * allocated by a thread of the call's own (joined before the call returns) while the calling thread times them.
 * What a co-resident allocator (torch's caching allocator, another handle on another thread or rank of the same device) sees:
 * while pic_create runs, blocks of the state's size are allocated one after the other and up to a third of the free memory is
 * held; all but x and v are freed before it returns.  An allocation made by someone else in that window can fail for lack of
 * memory although it would fit a moment later: create large handles before filling the device, serialise creates per device, or
 * set pic_config.placement = PIC_PLACE_OFF (x | v in one block, no search, nothing held; config 2 then steps ~10 % slower when
 * the block falls into one region).  Results do not depend on the placement.
 * -> how many pairs were timed (1 with rates 0 = small state or search off, nothing timed), the read+write rate of the pair kept
 * and of the slowest pair seen, in GB/s, and the wall time the search took; any pointer may be NULL. */
int pic_placement_info(pic_handle* h, int* candidates, double* kept_gbytes_per_s, double* slowest_gbytes_per_s, double* seconds);

/* The same report with how the search ended and where its time went, all legs together (ABI 4).  malloc_seconds is the part that
 * memory handed out for the first time since the device came up makes expensive (the driver clears it inside hipMalloc: 1.3-6 ms per
 * 512 MB block against 20-150 us for memory that has been allocated and released before); timing_seconds the streaming passes
 * (filler, the leg's reference pair, one untimed + one timed pass per candidate); free_seconds the release of the blocks not kept. */
typedef struct {
  int32_t pairs_timed;             /* 0: no search */
  int32_t blocks;                  /* candidate blocks allocated in all (timed or walked over) */
  int32_t outcome;                 /* PIC_PLACED_* */
  int32_t legs;                    /* legs of the search run so far: pic_create runs one, resets may run more (see above) */
  double kept_gbytes_per_s;        /* read + write rate of the bare stream over (x, v kept) */
  double slowest_gbytes_per_s;     /* ... over the slowest pair timed */
  double seconds;                  /* wall time of the search, all legs */
  double malloc_seconds, timing_seconds, free_seconds;
} pic_placement;
int pic_placement_stats(pic_handle* h, pic_placement* out);

int pic_sync(pic_handle* h);
/* Number of particle positions found non-finite or out of range by the last sweeps (0 = healthy). */
int pic_bad_count(pic_handle* h, int64_t* count);
const char* pic_last_error(pic_handle* h);   /* h may be NULL: error of the last failed pic_create */
int pic_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PICSTEP_H */
