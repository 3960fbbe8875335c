"""Throughput of the five BASELINE.json configurations at their single-GPU share (not the headline
bench: `bench.py` is config 2).  Prints one line per configuration.

    python profiles/configs.py [1 2 3 4 5]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ocplasma_amd
from ocplasma_amd.control.actuator import E_field
from ocplasma_amd.env.batched import BatchedPIC
from ocplasma_amd.env.dist import BumpOnTail
from ocplasma_amd.env.pic import PIC

L = 50.0
DEV = "cuda:0"


def synth(kind, E, N, dtype, seed):
    """Synthetic ensembles of SURVEY 8d, generated on the device in the target dtype (row by row to
    keep the temporaries small at N = 1e7)."""
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    x = torch.empty((E, N), device=DEV, dtype=dtype)
    v = torch.empty((E, N), device=DEV, dtype=dtype)
    for e in range(E):
        xe = torch.rand(N, generator=g, device=DEV, dtype=torch.float64) * L
        ve = torch.randn(N, generator=g, device=DEV, dtype=torch.float64)
        if kind == "bump":
            n1 = int(N * (1 / 1.2))
            ve[n1:] = ve[n1:] + 3.0
        else:
            ve[: N // 2] += 3.0
            ve[N // 2:] -= 3.0
        ve *= 1 + 0.1 * torch.sin(2 * np.pi * 2 * xe / L)
        x[e] = xe.clamp_(max=float(np.nextafter(np.float32(L), np.float32(0)))).to(dtype)
        v[e] = ve.to(dtype)
    return x, v


def run_batched(tag, kind, E, N, Ng, dtype, accum, steps, actions_modes=0):
    tdt = torch.float64 if dtype == "float64" else torch.float32
    env = BatchedPIC(E, N, Ng, L=L, dt=0.1, dtype=dtype, accum_dtype=accum)
    x, v = synth(kind, E, N, tdt, 1234)
    torch.cuda.synchronize()
    env.reset_device(x.data_ptr(), v.data_ptr())
    env.sync()
    del x, v
    torch.cuda.empty_cache()
    ke0, pe0, _ = env.energies()
    if actions_modes:
        env.set_actuator(E_field(L, Ng, actions_modes))
        g = torch.Generator(device=DEV)
        g.manual_seed(5)

        def one():
            a = (torch.rand((E, 2 * actions_modes), generator=g, device=DEV, dtype=torch.float64) - 0.5) * 2.5
            torch.cuda.current_stream().synchronize()
            env.step_actions_device(a.data_ptr(), 1)
            return a
    else:
        def one():
            env.step(None, 1)
    keep = [one() for _ in range(3)]
    env.sync()
    t0 = time.perf_counter()
    if actions_modes:
        keep = [one() for _ in range(steps)]       # a new random action per step (config 3)
    else:
        env.step(None, steps)
    env.sync()
    el = time.perf_counter() - t0
    ke, pe, _ = env.energies()
    esz = 8 if dtype == "float64" else 4
    ps = N * E * steps / el
    print(f"{tag}: {ps:.3e} particle-steps/s  {el / steps * 1e3:.3f} ms/step  "
          f"{ps * 14 * esz / 8e12 * 100:.1f}% of 8 TB/s on the {14 * esz}-B metric  "
          f"energy drift {np.max(np.abs((ke + pe) / (ke0 + pe0) - 1)):.1e}  bad={env.bad_count()}", flush=True)
    env.close()


def run_config1(steps=2000):
    np.random.seed(42)
    sim = PIC(N=10000, N_mesh=128, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2,
              interpol="CIC", init_dist=BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=10000, L=L))
    for _ in range(1000):                          # the first few hundred steps run ~1.4x slower (clock ramp)
        sim.update_state(None)
        sim.get_energy()
    t0 = time.perf_counter()
    for _ in range(steps):                         # run_wo_oc.py:108-125 without plots / KL
        sim.update_state(None)
        sim.get_energy()
        sim.get_electric_energy()
    el = time.perf_counter() - t0
    print(f"config 1 (PIC drop-in, N=1e4, Ng=128, 1 env, fp64, update_state + 2 energy reads per step): "
          f"{el / steps * 1e6:.1f} us/step  {1e4 * steps / el:.3e} particle-steps/s", flush=True)
    h = sim._ensure_handle()
    h.sync()
    t0 = time.perf_counter()
    h.step(None, steps)
    h.sync()
    el = time.perf_counter() - t0
    print(f"config 1 ({steps} steps in one pic_step call, no host reads): {el / steps * 1e6:.1f} us/step  "
          f"{1e4 * steps / el:.3e} particle-steps/s", flush=True)
    sim.close()


if __name__ == "__main__":
    which = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5]
    if 1 in which:
        run_config1()
    if 2 in which:
        run_batched("config 2 (bump-on-tail N=1e6 Ng=256 64 envs fp64)", "bump", 64, 1_000_000, 256, "float64", None, 20)
    if 3 in which:
        run_batched("config 3 (two-stream N=1e6 Ng=512 128 envs fp32 particles / packed fixed-point LDS mesh, random "
                    "actions every step)", "two", 128, 1_000_000, 512, "float32", None, 20, actions_modes=3)
    if 4 in which:
        run_batched("config 4 share (bump-on-tail N=4e6 Ng=1024 64 envs fp64)", "bump", 64, 4_000_000, 1024,
                    "float64", None, 10)
    if 5 in which:
        run_batched("config 5 share (bump-on-tail N=1e7 Ng=256 128 envs, fp32 push / fixed-point deposit / fp64 Poisson)",
                    "bump", 128, 10_000_000, 256, "float32", None, 5)
