#!/usr/bin/env python3
"""Step time of every particle format x shape x schedule of one exported tree (profiles/mk_ab.sh), through the API that has been
stable since round 1 (BatchedPIC.step): a wide net for regressions that parity tests cannot see (a spilling kernel computes the
same bits).   python profiles/variants.py profiles/ab/<tree> > table"""
import os
import sys
import time

tree = os.path.abspath(sys.argv[1])
sys.path.insert(0, tree)
import ocplasma_amd  # noqa: E402
from ocplasma_amd.env.batched import BatchedPIC  # noqa: E402

FORMATS = [("float64", "float"), ("float32", "float"), ("float32", "fixed32")]
CASES = [(64, n, 250, 200) for n in (2000, 4000, 5000, 8000)] + [(512, 5000, 250, 100), (8, 1_000_000, 256, 30), (64, 20000, 128, 100)]
for E, N, Ng, steps in CASES:
    for dtype, pos in FORMATS:
        for shape in ("CIC", "TSC"):
            try:
                env = BatchedPIC(E, N, Ng, L=50.0, dt=0.1, dtype=dtype, position_dtype=pos, interpol=shape)
            except Exception as e:          # a shape this tree refuses
                print(f"{E:5d} x {N:8d} Ng={Ng:4d} {dtype} {pos:7s} {shape}  refused: {str(e)[:60]}")
                continue
            env.reset_sampled("bump-on-tail", seed=3)
            env.step(None, nsteps=steps)
            env.sync()
            best = 1e9
            for _ in range(3):
                t = time.perf_counter()
                env.step(None, nsteps=steps)
                env.sync()
                best = min(best, (time.perf_counter() - t) / steps * 1e6)
            print(f"{E:5d} x {N:8d} Ng={Ng:4d} {dtype} {pos:7s} {shape}  {env._h.schedule():9s} {best:9.2f} us/step", flush=True)
            env.close()
