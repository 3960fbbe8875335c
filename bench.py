#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s of the PIC step (PIC.update_state of the reference) on
BASELINE config 2 -- bump-on-tail, N = 1e6 particles, Ng = 256, 64 environments per GPU, fp64.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no WORLD_SIZE in the environment: starts those N ranks itself, as child
                                           processes made BEFORE this process touches the GPU, and relays their line)

One rank per GPU; environments are sharded (64 per rank, weak scaling), no collective inside the
step; the only RCCL traffic is the all-gather of per-environment returns after the K steps.
Rank 0 prints ONE JSON line (contract in DESIGN.md "Measurement").

Order of the regions (profiles/experiments_r4.md 2): a device that has rested for >= 3 ms runs its next ~60 ms of
sweeps 13 % -> 1 % slow, whatever ran on it before and whichever handle steps (the CPU leg in front of this script's GPU part
is such a rest).  The default order therefore times the K headline steps LAST, behind the regions this script measures anyway:
    cold_start (the first K steps of the handle, on the rested device: reported, round 3's headline) -> the event-bracketed pass
    (K steps, per-kernel durations for `roofline`) -> steady_state (~2 s of steps) -> W warm-up steps -> K timed steps = `value`.
`--order cold` puts W + K first, as rounds 1-3 had it.  `value` is K steps behind W warm-up steps either way.
"""
import argparse
import json
import os
import sys
import time

# the CPU baseline is a one-thread-per-environment figure: keep BLAS / OpenMP from fanning out
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# SURVEY 8d's algorithmic figure: 14 words per particle-step (4 reads + 3 writes of x and v: four deposits need four
# passes).  The schedule that ships moves 12 (sweep A's read is folded into the previous sweep D): 3 reads + 3 writes.
ALGORITHMIC_BYTES_PER_PARTICLE_STEP = {"float64": 112, "float32": 56}
MOVED_BYTES_PER_PARTICLE_STEP = {"float64": 96, "float32": 48}
# algorithmic particle-array words moved by one launch of each sweep (reads + writes of x and v)
SWEEP_WORDS = {"sweep_A": 2, "sweep_B": 4, "sweep_C": 4, "sweep_D": 4}


def synth_bump_on_tail_device(torch, num_envs, N, L, dtype, device, seed, a=0.2, vb=3.0, vth=1.0, A=0.1, n_mode=2):
    """SURVEY 8d synthetic ensemble, generated on the device: x ~ U[0,L), v a 1/(1+a) : a/(1+a)
    mixture of N(0,1) and N(vb, vth^2), then v *= 1 + A sin(2 pi n_mode x / L) (pic.py:68)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.rand((num_envs, N), generator=g, device=device, dtype=torch.float64) * L
    x = torch.clamp(x, max=float(np.nextafter(L, 0.0)))
    v = torch.randn((num_envs, N), generator=g, device=device, dtype=torch.float64)
    n1 = int(N * (1 / (1 + a)))
    v[:, n1:] = v[:, n1:] * vth + vb
    v *= 1 + A * torch.sin(2 * np.pi * n_mode * x / L)
    return x.to(dtype).contiguous(), v.to(dtype).contiguous()


PMC_SUMMARY = "r4_summary.json"          # committed rocprofv3 evidence of this round (profiles/collect.sh)
# VALU issue ceiling for the resident schedule's roofline, in 1e9 wave-instructions per second: 256 CUs x 4 SIMDs at the 2.4 GHz
# peak clock, one wave-instruction per SIMD every 2 cycles for 32-bit operations (a SIMD-32 takes a 64-lane wave in two
# passes) and every 4 cycles for the float64 operations that make up the float64 push (78.6 TFLOP/s FP64 vector = 16 FMA
# lanes per SIMD and cycle; MI355X_MICROARCH.md)
VALU_PEAK_GINST_S = {"float64": 256 * 4 * 2.4 / 4, "float32": 256 * 4 * 2.4 / 2}


def pmc_traffic(kernel, args, E, N, Ng):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, profiles/summarize.py) -- only when they were taken on this exact workload."""
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if not os.path.exists(path) or (E, N, Ng, args.dtype, args.positions) != (64, 1_000_000, 256, "float64", "float"):
        return None, None
    ks = json.load(open(path)).get("kernels", {})
    # inside a multi-step call sweeps B and D run as B2 / D2: the counters of whichever form the collection launched more
    forms = [k for k in (kernel, kernel + "2") if k in ks]
    val = ks[max(forms, key=lambda k: ks[k].get("calls", 0))].get("hbm_bytes_per_launch") if forms else None
    return val, f"profiles/{PMC_SUMMARY} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                "not measured by this run)"


def pmc_valu_per_particle_step(args, N, Ng):
    """VALU wave-instructions per particle-step of the resident kernel (SQ_INSTS_VALU of a rocprofv3 --pmc pass divided by the
    particle-steps of the launch, profiles/pmc_resident.sh), for the particle format of this run; None if not collected."""
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if not os.path.exists(path):
        return None, None
    table = json.load(open(path)).get("resident_valu_wave_insts_per_particle_step", {})
    key = f"{args.dtype}/{args.positions}/N={N}/Ng={Ng}" + ("/history" if args.history else "")     # (every refresh made: more work)
    return table.get(key), f"profiles/{PMC_SUMMARY}: SQ_INSTS_VALU per launch / particle-steps per launch, {key}"


def _cpu_env(args):
    """One environment of the CPU baseline: (steps done, seconds) for the faithful NumPy oracle."""
    N, Ng, L, dt, budget_s, seed = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import pic_oracle as po
    x0, v0 = po.synthetic_bump_on_tail(N, L, seed=seed)
    sim = po.OraclePIC(x0, v0, Ng, L=L, dt=dt, perturb=False, faithful=True)
    t0 = time.perf_counter()
    sim.update_state(None)
    one = time.perf_counter() - t0
    steps = int(max(2, min(30, budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.update_state(None)
    return steps, time.perf_counter() - t0


def _host_cpu():
    """'<model name>, <n> logical cores' of the box the baseline runs on (north star: core count stated)."""
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return f"{model}, {os.cpu_count()} logical cores ({usable} usable by this process)"


def cpu_baseline(N, Ng, L, dt, budget_s=20.0, procs=1):
    """The NumPy oracle with the reference's call structure (7 compute_E + refresh per step, dense
    Ng x Ng operators, np.bincount) on a bounded sample of the same workload: one thread (how the
    reference runs), or `procs` independent environments on `procs` processes (--cpu-procs)."""
    if procs <= 1:
        steps, el = _cpu_env((N, Ng, L, dt, budget_s, 1234))
        return {"value": N * steps / el, "unit": "particle-steps/s", "cores": 1, "kind": "port",
                "sample": f"1 env of N={N}, Ng={Ng}, {steps} steps of the NumPy oracle (faithful call structure), "
                          f"{el / steps * 1e3:.3g} ms/step", "host": _host_cpu()}
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(procs) as pool:
        t0 = time.perf_counter()
        res = pool.map(_cpu_env, [(N, Ng, L, dt, budget_s, 1234 + p) for p in range(procs)])
        wall = time.perf_counter() - t0
    rate = sum(N * s / e for s, e in res)
    return {"value": rate, "unit": "particle-steps/s", "cores": procs, "kind": "port",
            "sample": f"{procs} processes x 1 env of N={N}, Ng={Ng}, {res[0][0]} steps each of the NumPy oracle "
                      f"(faithful call structure), sum of per-process rates, {wall:.0f} s wall incl. start-up",
            "host": _host_cpu()}


def self_launch(n):
    """Start `n` ranks of this script under torch.distributed.run on this node (127.0.0.1, a free port) and wait for them."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this host driver
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=64, help="environments per GPU")
    ap.add_argument("--particles", type=int, default=1_000_000)
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--dtype", default="float64", choices=["float64", "float32"])
    ap.add_argument("--accum", default=None, choices=[None, "fix64", "fixed", "float64"])
    ap.add_argument("--positions", default="float", choices=["float", "fixed32"], help="fixed32: 32-bit fixed-point x (float32)")
    ap.add_argument("--interpol", default="CIC", choices=["CIC", "TSC"], help="particle shape (the headline workload is CIC)")
    ap.add_argument("--blocks-per-env", type=int, default=0)
    ap.add_argument("--init", default="bump-on-tail", choices=["bump-on-tail", "two-stream"],
                    help="two-stream: BASELINE config 3's ensemble, drawn by the device sampler")
    ap.add_argument("--actions", type=int, default=0, metavar="M",
                    help="a new random action (2M Fourier coefficients in [-1.25, 1.25], SURVEY 8d) for every environment and "
                         "step, turned into E_ext by the device actuator (BASELINE config 3); 0 = no control")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="BASELINE.json configuration by number, at its single-GPU share (overrides the workload flags): "
                         "1 = N=1e4/Ng=128/1 env fp64; 2 = the default; 3 = two-stream N=1e6/Ng=512/128 envs, float32 velocities with "
                         "32-bit FIXED-POINT positions (not BASELINE's plain \"fp32\": add --positions float for float32 positions), "
                         "a new random action every step; 4 = N=4e6/Ng=1024/64 envs fp64; 5 = N=1e7/Ng=256/128 envs fp32")
    ap.add_argument("--steady-steps", type=int, default=-1,
                    help="steps of a second, longer region printed as `steady_state` (0 = skip; -1 = about 2 s of GPU work, between "
                         "200 and 20000 steps: 1950 at config 2 -- long enough for an outside sampler such as rocm-smi to see the "
                         "device at work and for the figure to span the device's power management)")
    ap.add_argument("--order", default="warm", choices=["warm", "cold"],
                    help="warm (default): cold_start, the event-bracketed pass and steady_state run BEFORE the W warm-up and K timed "
                         "steps, so that the timed region sees the device in its working state; cold: W + K first (rounds 1-3), the "
                         "other regions behind them")
    ap.add_argument("--per-step-calls", action="store_true",
                    help="with --actions: one pic_step_actions call per step (a trainer's loop: the action of step s is only known "
                         "after step s-1) instead of ONE pic_step_actions_traj call for all steps")
    ap.add_argument("--history", action="store_true",
                    help="pic_step_history instead of pic_step: the energies of EVERY step are recorded, so the resident schedule "
                         "makes every step's post-step refresh (SURVEY 8d's definition of a step; a plain pic_step(nsteps) call "
                         "skips the refreshes nothing can observe, `refresh: last_only`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--probe-passes", type=int, default=100, help="passes of the copy-ceiling probe in front of the warm-up")
    ap.add_argument("--cpu-procs", type=int, default=1, help="processes (one env each) for the CPU baseline")
    ap.add_argument("--profile-steps", type=int, default=-1, help="steps of the event-bracketed pass (-1 = --steps)")
    args = ap.parse_args()
    presets = {1: dict(envs=1, particles=10_000, mesh=128),
               2: dict(),
               3: dict(envs=128, mesh=512, dtype="float32", positions="fixed32", init="two-stream", actions=3),
               4: dict(particles=4_000_000, mesh=1024),
               5: dict(envs=128, particles=10_000_000, dtype="float32")}
    explicit = {a.dest for a in ap._actions if any(o in sys.argv for o in a.option_strings)}
    for k, val in presets.get(args.config, {}).items():
        if k not in explicit:                                # an explicit flag beside --config wins
            setattr(args, k, val)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as the driver runs the 1-GPU leg: start the N ranks ourselves.  Child processes, made while this
        # process has not touched the GPU (an exec or fork from a GPU-initialised process takes the machine down on this pool);
        # their rank 0 prints the JSON line on our stdout, and their exit code is ours.
        raise SystemExit(self_launch(args.gpus))
    launched = "WORLD_SIZE" in os.environ       # under torch.distributed.run (also with ONE rank: the RCCL path runs then too)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run, or without WORLD_SIZE set")

    # CPU baseline first: with --cpu-procs > 1 it starts worker processes, which must happen before this
    # process initialises the GPU (no exec from a GPU-initialised process on this pool).
    cpu_leg = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_leg = cpu_baseline(args.particles, args.mesh, 50.0, 0.1, procs=args.cpu_procs)

    import torch
    import ocplasma_amd
    from ocplasma_amd.env.batched import BatchedPIC

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the PIC step has no CPU path")
    # BENCH_BACKEND=gloo + BENCH_SAME_DEVICE=1: rehearsal of the N>1 path on a one-GPU box (every rank
    # on cuda:0, collectives over gloo on host copies).  The driver's runs use the defaults: RCCL, one GPU per rank.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("BENCH_SAME_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    cdev = f"cuda:{dev_index}" if backend == "nccl" else "cpu"      # where collective buffers live
    dist = None
    if launched:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
            # one rank per GPU over RCCL, or this is not the run the line claims to be
            if dist.get_backend() != "nccl":
                raise SystemExit(f"process group backend is {dist.get_backend()!r}, not nccl (RCCL)")
            if world > torch.cuda.device_count():
                raise SystemExit(f"{world} ranks but only {torch.cuda.device_count()} GPUs visible: ranks would share a device")
        else:
            dist.init_process_group(backend)

    N, Ng, E, L = args.particles, args.mesh, args.envs, 50.0
    if args.steady_steps < 0:       # ~2 s: the step moves 96 (48) bytes per particle at ~6 TB/s, and takes >= 14 us however small
        est_ms = max(0.014, N * E * MOVED_BYTES_PER_PARTICLE_STEP[args.dtype] / 6.0e12 * 1e3)
        args.steady_steps = int(min(20000, max(200, round(2000.0 / est_ms, -1))))
    tdtype = torch.float64 if args.dtype == "float64" else torch.float32
    env = BatchedPIC(E, N, Ng, L=L, dt=0.1, device=dev_index, dtype=args.dtype, accum_dtype=args.accum,
                     blocks_per_env=args.blocks_per_env, position_dtype=args.positions, interpol=args.interpol)
    local_rank = dev_index
    if args.init == "two-stream":
        env.reset_sampled("two-stream", v0=3.0, sigma=1.0, A=0.1, n_mode=2, seed=1234 + rank)
    else:
        x0, v0 = synth_bump_on_tail_device(torch, E, N, L, tdtype, f"cuda:{local_rank}", seed=1234 + rank)
        torch.cuda.synchronize()
        env.reset_device(x0.data_ptr(), v0.data_ptr())
        env.sync()
        del x0, v0
    ke0, pe0, _ = env.energies()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        env.sync()

    acts = None
    if args.actions > 0:
        from ocplasma_amd.control.actuator import E_field
        env.set_actuator(E_field(L, Ng, args.actions))
        g = torch.Generator(device=f"cuda:{local_rank}")
        g.manual_seed(4321 + rank)
        nact = max(args.steps, args.warmup, args.steady_steps, args.profile_steps, 1)
        acts = (torch.rand((nact, E, 2 * args.actions), generator=g, device=f"cuda:{local_rank}", dtype=torch.float64) * 2.5 - 1.25)
        torch.cuda.synchronize()

    def run_steps(k):
        """k environment steps in one call: pic_step, or (with --actions) pic_step_actions_traj with a new action for every
        step, E_ext built from it inside the field phases; --per-step-calls: k pic_step_actions calls instead."""
        if k <= 0:
            return
        if acts is None and args.history:
            env.step_history(None, k)
        elif acts is None:
            env.step(None, nsteps=k)
        elif args.per_step_calls:
            for i in range(k):
                env.step_actions_device(acts[i].data_ptr(), 1)
        else:
            env._h.step_actions_traj_device(acts.data_ptr(), k)

    # The copy ceiling of this device for the sweeps' access shape (reported next to the 8 TB/s spec): 100 passes over scratch
    # arrays of the state's size.  Every rank does it.
    copy_gbs = env.stream_probe(args.probe_passes)

    def timed_region(k):
        """k steps on this rank's own clock (stream drained before and after) -> ms per step"""
        env.sync()
        t = time.perf_counter()
        run_steps(k)
        env.sync()
        return (time.perf_counter() - t) / k * 1e3

    def event_pass():
        """K steps with every launch bracketed by HIP events on the library's own stream (the brackets cost ~2 % of a step, so
        they stay out of `value`) -> (per-kernel {name: (ms, launches)}, ms per step of the pass)"""
        env.profile(True)
        ms = timed_region(psteps)
        prof = env.profile_read()
        env.profile(False)
        return prof, ms

    def setup_views_and_collectives():
        """Everything the timed region and its collectives do for the first time, done once untimed: the zero-copy views, RCCL's
        set-up of each collective.  In the default order this runs BEFORE the measured regions, so that nothing but the W warm-up
        steps and a barrier stands between the event pass and the timed steps (the device would rest for milliseconds here)."""
        if cdev != "cpu":
            env.sync()
            env.energy_views_torch()
        if dist is not None:
            w = env.energy_views_torch()["PE_reward"] if cdev != "cpu" else torch.as_tensor(env.energies()[2], device=cdev)
            dist.all_gather([torch.empty_like(w) for _ in range(world)], w)
            dist.all_reduce(torch.zeros(1, device=cdev, dtype=torch.float64), op=dist.ReduceOp.MAX)
            dist.barrier()

    psteps = args.steps if args.profile_steps < 0 else args.profile_steps
    cold = steady = prof = ms_per_step_events = None
    if args.order == "warm":
        setup_views_and_collectives()
        # The regions this script measures anyway, in front of the headline one (module docstring): every rank runs them, so that
        # all devices of a multi-GPU run are in the same state when the timed region starts.
        cold = {"steps": args.steps, "ms_per_step": timed_region(args.steps)}
        if psteps > 0:
            prof, ms_per_step_events = event_pass()
        if args.steady_steps > 0:        # last: nothing but the W warm-up steps and a barrier between its end and the timed region
            steady = {"steps": args.steady_steps, "ms_per_step": timed_region(args.steady_steps)}

    run_steps(args.warmup)
    if args.order == "cold":
        setup_views_and_collectives()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    env.sync()
    t_steps = time.perf_counter() - t0              # this rank's K steps, before any collective
    # What the ranks exchange is each environment's field energy PE_reward as the step left it -- the zero-copy device view when
    # the collective runs on the GPU (RCCL), a host copy for gloo -- and the return max(1 - PE_reward, 0) (reward.py:72) is formed from
    # the gathered array after the timed region: with one rank nothing at all is launched here (round 3 formed the reward first,
    # two torch kernels and their launch latency, 0.2 ms = 1 % of a 20-step region)
    returns = env.energy_views_torch()["PE_reward"] if cdev != "cpu" else torch.as_tensor(env.energies()[2], device=cdev)
    t_g0 = time.perf_counter()
    if dist is not None:
        gathered = [torch.empty_like(returns) for _ in range(world)]
        dist.all_gather(gathered, returns)          # the one collective: per-environment returns
        returns = torch.cat(gathered)
    t_gather = time.perf_counter() - t_g0           # includes waiting for the slowest rank to arrive
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [t_steps / args.steps * 1e3]
    gather_ms = [t_gather * 1e3]
    if dist is not None:
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([t_steps / args.steps * 1e3, t_gather * 1e3], device=cdev, dtype=torch.float64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                # after the timed region: per-rank step time and gather time
        per_rank_ms = [float(e[0]) for e in every]
        gather_ms = [float(e[1]) for e in every]

    if args.order == "cold":
        # rounds 1-3: the longer region and the event pass behind the headline one
        if rank == 0 and args.steady_steps > 0:
            steady = {"steps": args.steady_steps, "ms_per_step": timed_region(args.steady_steps)}
        if rank == 0 and psteps > 0:
            prof, ms_per_step_events = event_pass()
    for region in (cold, steady):
        if region is not None:
            region["value"] = N * E / (region["ms_per_step"] * 1e-3)       # this rank's particle-steps/s

    returns = torch.clamp(1.0 - returns, min=0.0)
    # health: nothing non-finite, energy conserved over the whole run (every region above)
    ke, pe, _ = env.energies()
    drift = float(np.max(np.abs((ke + pe) / (ke0 + pe0) - 1)))
    bad = env.bad_count()

    roof = None
    kernels = {}
    if rank == 0 and prof is not None:
        esz = 8 if args.dtype == "float64" else 4
        for k, (ms, cnt) in prof.items():
            kernels[k] = {"avg_ms": ms / cnt, "launches": cnt}
            if k == "resident":
                kernels[k]["steps_per_launch"] = 1 if (acts is not None and args.per_step_calls) else psteps
        if "resident" in prof:
            # Small environments: the whole call is one launch that keeps the particles in registers; HBM sees them once at
            # entry and once at exit.  What bounds it is VALU issue and dependent latency of the ONE CU an environment
            # occupies, so the roofline is priced in VALU wave-instructions, not bytes.
            dom = "resident"
            avg_s = prof[dom][0] / prof[dom][1] * 1e-3
            spl = kernels[dom]["steps_per_launch"]
            w, w_src = pmc_valu_per_particle_step(args, N, Ng)
            ach = None if w is None else w * N * E * spl / avg_s / 1e9
            cus = min(E, 256)
            peak = VALU_PEAK_GINST_S[args.dtype]
            roof = {"bound": "valu+latency", "kernel": dom, "achieved": ach, "peak": peak, "unit": "Gwave-inst/s",
                    "frac": None if ach is None else ach / peak,
                    "frac_of_the_CUs_in_use": None if ach is None else ach / (peak * cus / 256),
                    "traffic": None, "valu_wave_insts_per_particle_step": w, "counter_source": w_src,
                    "avg_launch_ms": avg_s * 1e3, "steps_per_launch": spl,
                    "ms_per_step_with_event_brackets": ms_per_step_events, "measured_inplace_copy_GBs": copy_gbs}
        else:
            dom = max((k for k in prof if k in SWEEP_WORDS), key=lambda k: prof[k][0])
            alg_bytes = SWEEP_WORDS[dom] * esz * N * E
            avg_s = prof[dom][0] / prof[dom][1] * 1e-3
            ach = alg_bytes / avg_s / 1e9
            traffic, traffic_source = pmc_traffic(dom, args, E, N, Ng)
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_s * 1e3,
                    "ms_per_step_with_event_brackets": ms_per_step_events,
                    "measured_inplace_copy_GBs": copy_gbs}

    total_ps = N * E * world * args.steps
    value = total_ps / elapsed
    out = {
        "metric": "particle-steps/sec (N×envs×steps) at N=1e6, Ng=256; % HBM roofline",   # BASELINE.json, verbatim
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if args.dtype == "float64" else "f32", "data": "synthetic",
        "config": {"workload": f"{'configs[1]: ' if (N, Ng, E, args.dtype) == (1_000_000, 256, 64, 'float64') else ''}"
                               f"{args.init}, N={N}, Ng={Ng}, {E} envs per GPU, {args.dtype}"
                               f"{' (fixed-point positions)' if args.positions == 'fixed32' else ''}{', TSC' if args.interpol == 'TSC' else ''}, "
                               + (f"a new random action of {2 * args.actions} coefficients per step through the device actuator "
                                  + ("(one pic_step_actions call per step)" if args.per_step_calls else "(one pic_step_actions_traj call)")
                                  if args.actions > 0 else "no control (E_ext = None)") + ", Yoshida-4 step = PIC.update_state",
                   "envs_per_gpu": E, "particles_per_env": N, "mesh": Ng, "dt": env.dt, "schedule": env._h.schedule(),
                   # the post-step refresh (density, E_mesh, phi, energies: pic.py:145-146) of the steps inside one call
                   "refresh": "every_step" if (env._h.schedule() == "streaming" or args.history or args.per_step_calls
                                               or max(args.steps, 1) == 1) else "last_only",
                   "sharding": f"{world} x {E} envs, all-gather of returns only"},
        # whole-step fraction of the 8 TB/s peak on the bytes the schedule really moves (12 words per particle-step).
        # survey_112B_equivalent prices the same rate with SURVEY 8d's algorithmic 14 words: NOT a bandwidth fraction (2 of the
        # 14 words -- sweep A's read -- are never moved), only the figure to hold against SURVEY's 7.14e10 = 100 %.
        "hbm_frac_of_step": value / world * MOVED_BYTES_PER_PARTICLE_STEP[args.dtype] / (HBM_PEAK_GBS * 1e9),
        "survey_112B_equivalent": value / world * ALGORITHMIC_BYTES_PER_PARTICLE_STEP[args.dtype] / (HBM_PEAK_GBS * 1e9),
        # the regions around the headline one, in the order they ran (module docstring); each on rank 0's own clock
        "order": (["cold_start", "event_pass", "steady_state", "warmup", "timed"] if args.order == "warm"
                  else ["warmup", "timed", "steady_state", "event_pass"]),
        "cold_start": cold, "steady_state": steady,
        "per_rank_ms_per_step": per_rank_ms, "returns_all_gather_ms": gather_ms,
        "collective_backend": None if dist is None else dist.get_backend(), "gpus_visible": torch.cuda.device_count(),
        "energy_drift": drift, "bad_positions": bad, "mean_return": float(returns.mean().item()),
        # pic_create's search for two different HBM regions for x and v: (x, v) placements timed, bare-stream GB/s of the pair
        # kept and of the slowest pair seen, how the search ended (found | patience | timeout | memory | none) and where its
        # time went
        "placement": env._h.placement_stats(),
        "roofline": roof, "kernels": kernels,
    }
    if rank == 0:
        out["cpu_baseline"] = cpu_leg          # measured before the GPU was touched (rank 0, N = 1 only)
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
