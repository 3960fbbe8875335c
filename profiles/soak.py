"""Soak: config 2 (64 x 1e6 particles, Ng=256; fp64, or float32 as argv[2]) for thousands of steps; health checks
and elapsed time every 500 steps (the time column shows the sustained rate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ocplasma_amd
from ocplasma_amd import BatchedPIC
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dtype = sys.argv[2] if len(sys.argv) > 2 else "float64"          # float32 = packed fixed-point LDS accumulator
env = BatchedPIC(64, 1_000_000, 256, L=50.0, dt=0.1, dtype=dtype)
env.reset_sampled("bump-on-tail", seed=2026)
ke0, pe0, _ = env.energies()
t0 = time.perf_counter()
for k in range(0, steps, 500):
    env.step(None, 500)
    ke, pe, per = env.energies()
    n, Em, phi = env.fields()
    drift = np.max(np.abs((ke + pe) / (ke0 + pe0) - 1))
    charge = np.max(np.abs(n.sum(axis=1) * (50.0 / 256) - 50.0))
    print(f"step {k + 500:6d}  t={(k + 500) * env.dt:7.2f}  max energy drift {drift:.2e}  max |charge error| {charge:.1e}  "
          f"mean field energy {per.mean():.4e}  bad={env.bad_count()}  ({(time.perf_counter() - t0):.1f} s)", flush=True)
x, v = env.particles()
print("x in [0, L):", bool((x >= 0).all() and (x < 50.0).all()), " finite v:", bool(np.isfinite(v).all()))
