"""Energy history on the streaming schedule: us/step of pic_step_history for a few shapes.  usage: python profiles/history_steps.py [tree]"""
import os, sys, time
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import ocplasma_amd
from ocplasma_amd.env.batched import BatchedPIC
for E, N, Ng in ((1, 10000, 128), (4, 100000, 256), (12, 1000000, 256), (256, 5000, 250)):
    env = BatchedPIC(E, N, Ng, L=50.0, dt=0.1, blocks_per_env=(2 if N == 5000 else 0))
    env.reset_sampled("two-stream", seed=3)
    env.step_history(None, 20)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ke, pe, per = env.step_history(None, 200); best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    print(f"{E:4d} x N={N:8d} ({env._h.schedule()}): {best:8.1f} us/step with the energy history   sum(KE[-1])={ke[-1].sum():.10e}", flush=True)
    env.close()
