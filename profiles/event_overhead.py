"""Experiment: cost of per-launch HIP-event brackets inside the timed region (config 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ocplasma_amd
from ocplasma_amd.env.batched import BatchedPIC
E, N, Ng = 64, 1_000_000, 256
env = BatchedPIC(E, N, Ng, L=50.0, dt=0.1)
x0, v0 = bench.synth_bump_on_tail_device(torch, E, N, 50.0, torch.float64, "cuda:0", 1234)
torch.cuda.synchronize()
env.reset_device(x0.data_ptr(), v0.data_ptr()); env.sync()
env.step(None, 5); env.sync()
for rnd in range(3):
    for prof in (False, True):
        env.profile(prof)
        env.sync()
        t0 = time.perf_counter()
        env.step(None, 20); env.sync()
        el = time.perf_counter() - t0
        r = env.profile_read() if prof else {}
        print(f"events={prof!s:5} ms/step={el/20*1e3:.4f}", {k: round(v[0]/v[1], 4) for k, v in r.items()}, flush=True)
env.profile(False)
