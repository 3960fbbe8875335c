"""Host-side cost of the reference-style control loop on the reference's own environment size (N = 5000, Ng = 250,
two-stream): microseconds per iteration for growing slices of what ddpg.py:440-470 does per step."""
import os, sys, time
# usage: python profiles/pyloop_rl.py [tree]   (tree = an exported copy made by profiles/mk_ab.sh; default: this checkout)
sys.path.insert(0, os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ocplasma_amd
from ocplasma_amd import PIC, TwoStream, E_field

np.random.seed(42)
N, Ng, L, M = 5000, 250, 50.0, 5
sim = PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, A=0.1, n_mode=2, init_dist=TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))
act = E_field(L, Ng, M)
sim.set_actuator(act)
rng = np.random.default_rng(0)
actions = rng.uniform(-1.25, 1.25, (4096, 2 * M))
state0 = sim.get_state()


def timed(label, body, n=3000, warm=500):
    for k in range(warm):
        body(k)
    t0 = time.perf_counter()
    for k in range(n):
        body(k)
    sim._ensure_handle().sync()
    print(f"{label:78s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us/iteration", flush=True)


def host_field(k):
    a = actions[k % 4096]
    act.update_E(a[:M], a[M:])
    sim.update_state(act.compute_E())


timed("update_state(None), nothing read back", lambda k: sim.update_state(None))
timed("update_state(None) + get_energy()", lambda k: (sim.update_state(None), sim.get_energy()))
timed("host actuator (E_field.compute_E) + update_state(E_ext)", host_field)
timed("step(action): device actuator + update_state + energies + get_state (Gym tuple)", lambda k: sim.step(actions[k % 4096]))
timed("update_state(None) + get_state() (2N x 1 host copy)", lambda k: (sim.update_state(None), sim.get_state()))


def host_feedback(k):            # run_feedback.py:130-168 on the host: E_mesh read back every step, modes by FFT, actuator, step
    Ek = np.fft.fft(sim.E_mesh[:, 0]) / Ng * 2.0
    act.update_E(-Ek[1:M + 1].real, Ek[1:M + 1].imag)
    sim.update_state(act.compute_E())


timed("host feedback loop: E_mesh read back + FFT + E_field.compute_E + update_state(E_ext)", host_feedback)

from ocplasma_amd import Reward  # noqa: E402
rew = Reward(sim.get_state(), N_mesh=Ng, L=L, n_actions=2 * M)


def trainer_iteration(k):        # ddpg.py:421-468 with only the import lines changed: the reward deposits the state once more
    a = actions[k % 4096]
    act.update_E(a[:M], a[M:])
    sim.update_state(act.compute_E())
    nxt = sim.get_state()
    return rew.compute_reward(nxt, a)


timed("trainer iteration as the reference writes it: update_state(E) + get_state + Reward.compute_reward", trainer_iteration, n=1500, warm=200)
timed("  ... Reward.compute_reward(state, action) alone (a deposit + solve of a host state)", lambda k: rew.compute_reward(state0, actions[k % 4096]), n=1500, warm=200)
h = sim._ensure_handle()
timed("raw handle: pic_step(NULL, 1) through ctypes", lambda k: h.step(None, 1))
timed("raw handle: 10 steps per call", lambda k: h.step(None, 10), n=500, warm=50)
