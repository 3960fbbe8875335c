"""Does a short pic_step call cost more per step than a long one?  bench.py's 20-step regions read ~1 % above its 1950-step region on
the same warm device (r4h: 0.970 against 0.959 ms/step).  Config 2, device warmed by 600 steps, then back to back and repeated:
one call of 2000 steps; 100 calls of 20 steps, each timed as bench.py times its region (sync, clock, call, sync, clock); 20 calls of
100 steps; and the 20-step calls once more with the waits done by polling the stream instead of blocking on it.
    python3 profiles/call_size.py"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ocplasma_amd.env.batched import BatchedPIC
E, N, Ng, L = 64, 1_000_000, 256, 50.0
env = BatchedPIC(E, N, Ng, L=L, dt=0.1)
x0, v0 = bench.synth_bump_on_tail_device(torch, E, N, L, torch.float64, "cuda:0", seed=1234)
torch.cuda.synchronize()
env.reset_device(x0.data_ptr(), v0.data_ptr()); env.sync()
print("placement", env._h.placement_stats())
env.step(None, 600); env.sync()


def timed(k):
    env.sync()
    t = time.perf_counter()
    env.step(None, k)
    env.sync()
    return (time.perf_counter() - t) / k * 1e6


def timed_events(k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.use_torch_stream()
    e0.record(); env.step(None, k); e1.record(); e1.synchronize()
    env.use_own_stream()
    return e0.elapsed_time(e1) / k * 1e3


for rnd in range(2):
    a = timed(2000)
    b = [timed(20) for _ in range(100)]
    c = [timed(100) for _ in range(20)]
    d = [timed(5) for _ in range(100)]
    print(f"round {rnd}: one call of 2000 steps {a:.1f} us/step | 100 calls of 20: median {statistics.median(b):.1f} (min {min(b):.1f}, max {max(b):.1f}) | "
          f"20 calls of 100: median {statistics.median(c):.1f} | 100 calls of 5: median {statistics.median(d):.1f}", flush=True)
ev = [timed_events(20) for _ in range(50)]
print(f"20-step calls between two events on the stream (no host wait inside): median {statistics.median(ev):.1f} us/step, min {min(ev):.1f}")
ev = [timed_events(2000) for _ in range(2)]
print(f"2000-step calls between two events: {ev[0]:.1f} {ev[1]:.1f} us/step")
env.close()
