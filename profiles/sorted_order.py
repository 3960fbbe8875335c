"""Upper bound on what a cell-binned particle order could buy: the same ensemble with its particles in random
order and sorted by position, timed over the first steps (the order decays as the particles move).  LDS atomics of neighbouring
lanes then hit the same or neighbouring cells: no bank conflicts, same-address merging.  Nothing in the product sorts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocplasma_amd
from ocplasma_amd.env.batched import BatchedPIC

L = 50.0
for E, N, Ng, steps in ((12, 1_000_000, 256, 10), (64, 1_000_000, 256, 10), (256, 5000, 250, 5), (256, 5000, 250, 50)):
    env = BatchedPIC(E, N, Ng, L=L, dt=0.1)
    env.reset_sampled("bump-on-tail", seed=5)
    t = env.torch_views(); env.sync()
    x0, v0 = t["x"].clone(), t["v"].clone()
    xs, idx = torch.sort(x0, dim=1)
    vs = torch.gather(v0, 1, idx)
    torch.cuda.synchronize()
    line = f"{E} x N={N} ({env._h.schedule()}), first {steps} steps:"
    for name, (xa, va) in (("random order", (x0, v0)), ("sorted by x", (xs.contiguous(), vs.contiguous()))):
        best = 1e9
        for rep in range(3):
            env.reset_device(xa.data_ptr(), va.data_ptr()); env.sync()
            t0 = time.perf_counter(); env.step(None, steps); env.sync()
            best = min(best, (time.perf_counter() - t0) / steps * 1e6)
        line += f"  {name} {best:8.1f} us/step"
    print(line, flush=True)
    env.close()
