"""Experiment: where does the host time of the reference-style Python loop go (config 1)?"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
loop(50)
t0 = time.perf_counter(); loop(2000); print("us/step: %.1f" % ((time.perf_counter() - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable(); loop(2000); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:2600])
