"""Linear feedback control of the two-stream instability, entirely through the batched device API.

Shape of the reference's run_feedback.py:130-168: every step the first `max_mode` Fourier modes of the
self-consistent field become the action (cos coefficients -Re E_k, sin coefficients +Im E_k), the
actuator turns the action into an external field and the environment is stepped with it.  Here the
modes, the actuator product and the step all run on the MI355X.  With the reference's unit gain the whole
loop is ONE call (`BatchedPIC.step_feedback` -> pic_step_feedback: the action of every step is computed on the
device from the field the step before left, nothing crosses the host boundary until the recorded energies
and actions come back at the end); with another gain the action passes through the host every step
(`2 * max_mode` numbers per environment).

    python examples/feedback_control.py [num_envs] [N] [steps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocplasma_amd
from ocplasma_amd import BatchedPIC, E_field, TwoStream


def run(num_envs=4, N=20000, steps=400, N_mesh=128, L=50.0, max_mode=5, gain=1.0, seed=11, verbose=True):
    np.random.seed(seed)
    dist = TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L)
    xs, vs = [], []
    for _ in range(num_envs):
        dist.reinit()
        x, v = dist.get_sample()
        xs.append(x)
        vs.append(v * (1 + 0.1 * np.sin(2 * np.pi * 2 * x / L)))     # PIC.initialize perturbation, n_mode = 2
    x0, v0 = np.stack(xs), np.stack(vs)
    free = BatchedPIC(num_envs, N, N_mesh, L=L, dt=0.1)
    ctrl = BatchedPIC(num_envs, N, N_mesh, L=L, dt=0.1)
    ctrl.set_actuator(E_field(L, N_mesh, max_mode))
    for env in (free, ctrl):
        env.reset(x0, v0)
    if gain == 1.0:
        rec = ctrl.step_feedback(steps, actions=True, history=True)     # the closed loop, one call
        pe_free = free.step_history(None, steps)[2].mean(axis=1)
        pe_ctrl = rec["PE_reward"].mean(axis=1)
        effort = np.abs(rec["actions"]).max(axis=(1, 2))
        if verbose:
            for k in range(0, steps, 50):
                print(f"step {k:4d}  field energy: free {pe_free[k]:.4e}  controlled {pe_ctrl[k]:.4e}  max|action| {effort[k]:.3f}")
        free.close()
        ctrl.close()
        return pe_free, pe_ctrl, effort
    pe_free, pe_ctrl, effort = [], [], []
    for k in range(steps):
        action = gain * ctrl.feedback_actions(max_mode)               # [num_envs, 2 max_mode], from device modes
        ctrl.step_actions(action)
        free.step()
        pe_free.append(free.energies()[2].mean())
        pe_ctrl.append(ctrl.energies()[2].mean())
        effort.append(np.abs(action).max())
        if verbose and k % 50 == 0:
            print(f"step {k:4d}  field energy: free {pe_free[-1]:.4e}  controlled {pe_ctrl[-1]:.4e}  max|action| {effort[-1]:.3f}")
    free.close()
    ctrl.close()
    return np.array(pe_free), np.array(pe_ctrl), np.array(effort)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    pf, pc, eff = run(*a)
    print(f"peak field energy: free {pf.max():.4e}, controlled {pc.max():.4e} (ratio {pc.max() / pf.max():.3f}); "
          f"mean over the last quarter: free {pf[-len(pf)//4:].mean():.4e}, controlled {pc[-len(pc)//4:].mean():.4e}")
