import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ocplasma_amd as oc
for E in (8, 32, 64, 256):
    env = oc.BatchedPIC(E, 5000, 250, L=50.0, dt=0.1)
    env.set_actuator(oc.E_field(50.0, 250, 3))
    env.reset_sampled("bump-on-tail", seed=1)
    a = np.random.default_rng(0).uniform(-1, 1, (E, 6))
    for f, name in ((lambda: env.step_observe(actions=a), "step_observe(actions)"), (lambda: (env.step_actions(a), env.get_state(), env.energies()), "step_actions + get_state + energies")):
        for _ in range(20): f()
        t = time.perf_counter()
        for _ in range(200): f()
        print(f"E={E:4d} {name:40s} {(time.perf_counter() - t) / 200 * 1e6:8.1f} us/iteration", flush=True)
    env.close()
