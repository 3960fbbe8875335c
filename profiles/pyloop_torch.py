"""Experiment: does importing torch change the per-step host cost of the Python loop?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    if len(sys.argv) > 2 and sys.argv[2] == "init":
        torch.zeros(1, device="cuda:0")
import numpy as np
import ocplasma_amd
from ocplasma_amd import PIC, BumpOnTail
np.random.seed(42)
sim = PIC(N=10000, N_mesh=128, n0=1.0, L=50.0, dt=0.1, A=0.1, n_mode=2, init_dist=BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=10000, L=50.0))
def loop(n):
    for _ in range(n):
        sim.update_state(None)
        sim.get_energy()
        sim.get_electric_energy()
loop(100)
for r in range(3):
    t0 = time.perf_counter(); loop(2000); print(" ".join(sys.argv[1:]) or "no torch", "us/step: %.1f" % ((time.perf_counter() - t0) / 2000 * 1e6), flush=True)
