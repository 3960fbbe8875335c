"""Soak: thousands of steps with health checks and elapsed time every 500 steps (the time column shows the sustained rate).
    python profiles/soak.py [steps] [float64|float32] [float|fixed32] [envs] [N] [Ng]
defaults: config 2 (64 x 1e6 particles, Ng=256, fp64); `256 5000 250` runs the resident schedule on the reference's shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ocplasma_amd
from ocplasma_amd import BatchedPIC
arg = lambda i, d: sys.argv[i] if len(sys.argv) > i else d
steps, dtype, pos = int(arg(1, 5000)), arg(2, "float64"), arg(3, "float")
E, N, Ng = int(arg(4, 64)), int(arg(5, 1_000_000)), int(arg(6, 256))
env = BatchedPIC(E, N, Ng, L=50.0, dt=0.1, dtype=dtype, position_dtype=pos)
print(f"{E} envs x N={N}, Ng={Ng}, {dtype}, positions {pos}, schedule {env._h.schedule()}, dt={env.dt:.5f}")
env.reset_sampled("bump-on-tail", seed=2026)
ke0, pe0, _ = env.energies()
t0 = time.perf_counter()
for k in range(0, steps, 500):
    env.step(None, 500)
    ke, pe, per = env.energies()
    n, Em, phi = env.fields()
    drift = np.max(np.abs((ke + pe) / (ke0 + pe0) - 1))
    charge = np.max(np.abs(n.sum(axis=1) * (50.0 / Ng) - 50.0))
    print(f"step {k + 500:6d}  t={(k + 500) * env.dt:7.2f}  max energy drift {drift:.2e}  max |charge error| {charge:.1e}  "
          f"mean field energy {per.mean():.4e}  bad={env.bad_count()}  ({(time.perf_counter() - t0):.1f} s)", flush=True)
x, v = env.particles()
print("x in [0, L):", bool((x >= 0).all() and (x < 50.0).all()), " finite v:", bool(np.isfinite(v).all()))
