"""Does the speed of the config-2 step depend on where the particle arrays land in physical memory?
One process; the environment is created, timed and destroyed several times, then several are kept alive at once."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ocplasma_amd
from ocplasma_amd.env.batched import BatchedPIC


def make():
    env = BatchedPIC(64, 1_000_000, 256, L=50.0, dt=0.1)
    env.reset_sampled("two-stream", v0=3.0, sigma=1.0, A=0.1, n_mode=2, seed=7)
    env.sync()
    return env


def timed(env, k=20):
    env.step(None, nsteps=3)
    env.sync()
    t0 = time.perf_counter()
    env.step(None, nsteps=k)
    env.sync()
    return (time.perf_counter() - t0) / k * 1e6


print("create / time / destroy:", flush=True)
for i in range(8):
    env = make()
    x = env._h.device_ptrs()["x"]
    print(f"  #{i}: {timed(env):7.1f} us/step  again {timed(env):7.1f}   x=0x{int(x):x}  placement {env._h.placement_info()}", flush=True)
    env.close()
print("kept alive together:", flush=True)
envs = [make() for _ in range(6)]
for rnd in range(2):
    for i, env in enumerate(envs):
        x = env._h.device_ptrs()["x"]
        print(f"  round {rnd} env {i}: {timed(env):7.1f} us/step   x=0x{int(x):x}  placement {env._h.placement_info()}", flush=True)
for env in envs:
    env.close()
