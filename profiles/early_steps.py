import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, ocplasma_amd as oc
from ocplasma_amd.env.batched import BatchedPIC
E, N, Ng, L = 64, 1_000_000, 256, 50.0
env = BatchedPIC(E, N, Ng, L=L, dt=0.1)
x0, v0 = bench.synth_bump_on_tail_device(torch, E, N, L, torch.float64, "cuda:0", seed=1234)
torch.cuda.synchronize()
for trial in range(2):
    env.reset_device(x0.data_ptr(), v0.data_ptr()) if hasattr(env, "reset_device") else env._h.reset_device(x0.data_ptr(), v0.data_ptr())
    env.sync()
    out = []
    for chunk in range(12):
        t = time.perf_counter(); env.step(None, 10); env.sync(); out.append((time.perf_counter() - t) / 10 * 1e6)
    print("after reset, us/step per chunk of 10:", [round(o, 1) for o in out], flush=True)
env.step(None, 300); env.sync()
out = []
for chunk in range(6):
    t = time.perf_counter(); env.step(None, 10); env.sync(); out.append((time.perf_counter() - t) / 10 * 1e6)
print("300 steps later:", [round(o, 1) for o in out])
