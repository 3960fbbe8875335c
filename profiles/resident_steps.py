"""Resident schedule, many steps per call: microseconds per step with and without an external field.
usage: python profiles/resident_steps.py [tree]   (tree = an exported copy made by profiles/mk_ab.sh; default: this checkout)"""
import os, sys, time
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import ocplasma_amd
from ocplasma_amd.env.batched import BatchedPIC

L = 50.0
rng = np.random.default_rng(0)
for E, N, Ng, dtype, pos in ((64, 5000, 250, "float64", "float"), (256, 5000, 250, "float64", "float"), (64, 2000, 250, "float64", "float"),
                             (64, 8000, 250, "float64", "float"), (256, 5000, 250, "float32", "fixed32")):
    env = BatchedPIC(E, N, Ng, L=L, dt=0.1, dtype=dtype, position_dtype=pos)
    env.reset_sampled("two-stream", v0=3.0, sigma=1.0, A=0.1, n_mode=2, seed=7)
    ext = 0.05 * rng.normal(size=(E, Ng))
    line = f"{E:4d} envs N={N} {dtype}/{pos} ({env._h.schedule()}):"
    for name, e in (("no field", None), ("E_ext", ext)):
        env.step(e, nsteps=50)
        env.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            env.step(e, nsteps=500)
            env.sync()
            best = min(best, (time.perf_counter() - t0) / 500 * 1e6)
        line += f"  {name} {best:6.2f} us/step"
    ke, pe, _ = env.energies()
    print(line + f"   H[0]={ke[0] + pe[0]:.12e}", flush=True)
    env.close()
