"""Sweep time when the particles crowd into few cells (LDS atomic contention): us/step for a uniform plasma and for
beams of width sigma in x.  usage: python profiles/clustered.py [tree]"""
import os, sys, time
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import ocplasma_amd
from ocplasma_amd.env.batched import BatchedPIC
E, N, Ng, L = 16, 1_000_000, 256, 50.0
rng = np.random.default_rng(1)
env = BatchedPIC(E, N, Ng, L=L, dt=0.1)
for name, sigma in (("uniform", None), ("sigma = 5 dx", 5 * L / Ng), ("sigma = 0.5 dx", 0.5 * L / Ng), ("sigma = 0.01 dx", 0.01 * L / Ng)):
    x = rng.uniform(0, L, (E, N)) if sigma is None else np.mod(L / 2 + sigma * rng.normal(size=(E, N)), L)
    v = 1e-3 * rng.normal(size=(E, N))          # cold: the cluster stays where it is
    env.reset(x, v)
    env.step(None, 5); env.sync()
    t0 = time.perf_counter(); env.step(None, 40); env.sync()
    print(f"{name:16s} {(time.perf_counter() - t0) / 40 * 1e6:8.1f} us/step  bad={env.bad_count()}", flush=True)
env.close()
