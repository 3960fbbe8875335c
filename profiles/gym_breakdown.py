#!/usr/bin/env python3
"""Where one iteration of a Gym-style loop at the reference's size (N = 5000, Ng = 250, one environment) spends its time,
layer by layer: raw C ABI calls through ctypes, the Handle methods, PIC.step.   python profiles/gym_breakdown.py [N [Ng]]"""
import os
import sys
import time

import numpy as np

tree = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, tree)
import ocplasma_amd as oc  # noqa: E402

N, Ng, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 5000), (int(sys.argv[2]) if len(sys.argv) > 2 else 250), 50.0
rng = np.random.default_rng(0)


def bench(label, fn, n=2000):
    for _ in range(200):
        fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    print(f"{label:86s} {(time.perf_counter() - t) / n * 1e6:7.1f} us/iteration", flush=True)


h = oc._abi.Handle(N, Ng, 1, L, 1.0, 0.1)
act = oc.E_field(L, Ng, 3)
h.set_actuator(act.basis_cos, act.basis_sin)
h.reset(rng.uniform(0, L, (1, N)), rng.normal(0, 1, (1, N)))
a = rng.uniform(-1, 1, (1, 6))
bench("Handle.step(None, 1) + sync", lambda: (h.step(None, 1), h.sync()))
bench("Handle.step_actions(a, 1) + sync", lambda: (h.step_actions(a, 1), h.sync()))
bench("Handle.step(None, 1) + energies()", lambda: (h.step(None, 1), h.energies()))
bench("Handle.step(None, 1) + particles()", lambda: (h.step(None, 1), h.particles()))
bench("Handle.step(None, 1) + particles() + energies()", lambda: (h.step(None, 1), h.particles(), h.energies()))
if hasattr(h, "step_observe"):
    bench("Handle.step_observe(None, None, 1, particles=False)", lambda: h.step_observe(None, None, 1, particles=False))
    bench("Handle.step_observe(None, None, 1)", lambda: h.step_observe(None, None, 1))
    bench("Handle.step_observe(actions=a)", lambda: h.step_observe(None, a, 1))
h.close()


class Dist:
    def reinit(self):
        pass

    def get_sample(self):
        return rng.uniform(0, L, N), rng.normal(0, 1, N)


sim = oc.PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=1.0, gamma=5.0, init_dist=Dist())
sim.set_actuator(act)
bench("PIC.step(action) -> (obs, reward, done, info)", lambda: sim.step(a[0]))
bench("PIC.update_state(None); PIC.get_state(); PIC.get_reward_electric_energy()",
      lambda: (sim.update_state(None), sim.get_state(), sim.get_reward_electric_energy()))
sim.close()
