"""Experiment: per-step latency of one small environment vs workgroups per environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ocplasma_amd
from ocplasma_amd import _abi
rng = np.random.default_rng(0)
for N, Ng in ((10000, 128), (5000, 250), (40000, 400)):
    x0, v0 = rng.uniform(0, 50, (1, N)), rng.normal(0, 1, (1, N))
    for bpe in (0, 1, 2, 5, 10, 20, 40):
        h = _abi.Handle(N, Ng, 1, 50.0, 1.0, 0.05, blocks_per_env=bpe)
        h.reset(x0, v0)
        h.step(None, 50); h.sync()
        t0 = time.perf_counter(); h.step(None, 500); h.sync(); el = time.perf_counter() - t0
        print(f"N={N:6d} Ng={Ng:4d} blocks_per_env={bpe:3d}: {el / 500 * 1e6:6.1f} us/step", flush=True)
        h.close()
