"""Reward terms of the control problem, with the public surface of the reference's `Reward`
(``src/control/rl/reward.py:5-76``) and of the helpers it calls (``src/control/objective.py:8-35``).

What the trainers use (ddpg.py:344,381,455; ppo.py; sac.py) is

    reward = alpha * max(1 - PE / r_pe_n, 0) + beta * max(1 - IE / r_ie_n, 0)
    PE = 0.5 * sum(E_mesh(state)^2) * dx            (CIC, no N/L factor; objective.py:33)
    IE = sum(action^2) * L / 4                      (reward.py:52-54)
    r_pe_n = 1, r_ie_n = IE(ones(n_actions))        (reward.py:32-33)

`PE` needs a deposit + Poisson solve of the given state; here that evaluation runs on the device
(`pic_eval_field` through a cached single-environment probe handle) instead of a NumPy `compute_E`
with freshly built dense operators.  When the state is the environment's current one, the step has
already produced the same number (`PIC.get_reward_electric_energy`, `BatchedPIC.energies()[2]`) and
`Reward.reward_from_energy` turns it into the reward without another deposit.

The phase-space histogram / KL divergence (objective.py:8-18) is a logged diagnostic
(run_wo_oc.py:121), never part of `compute_reward`; it stays a host histogram.
"""
import numpy as np

from .._params import ParamMixin
from ..env.util import probe_handle

_TINY = 1e-12


def _probe(N, N_mesh, L, n0, device=0):
    """Single-environment handle used only to evaluate fields of host-supplied states (shared cache)."""
    return probe_handle(N, N_mesh, L, n0, "CIC", device)


def estimate_f(state, N_mesh, L, vmin, vmax, n0):
    """Phase-space density f(x, v) on an N_mesh x N_mesh grid, normalised so that sum(f) dx dv = n0."""
    n_part = state.shape[0] // 2
    dx, dv = L / N_mesh, (vmax - vmin) / N_mesh
    counts = np.histogram2d(state[:n_part].ravel(), state[n_part:].ravel(), bins=[N_mesh, N_mesh],
                            range=[[0, L], [vmin, vmax]])[0]
    counts *= n0 / dx / dv / n_part          # same operand order as objective.py:13
    return counts


def estimate_KL_divergence(f, feq, dx=0.1, dv=0.04):
    """sum_ij f log(f / (feq + 1e-12)) dx dv with the convention 0 log 0 = 0 (scipy's rel_entr)."""
    q = feq + _TINY
    terms = np.zeros_like(f, dtype=float)
    pos = f > 0
    terms[pos] = f[pos] * np.log(f[pos] / q[pos])
    return terms.sum() * dx * dv


def estimate_electric_energy(state, E_external, N_mesh, L, n0, device=0):
    """0.5 * sum((E_mesh + E_external)^2) * dx for the positions stored in ``state[:N]``."""
    state = np.asarray(state, dtype=np.float64).reshape(-1)
    n_part = state.shape[0] // 2
    ext = None if E_external is None else np.asarray(E_external, dtype=np.float64).reshape(1, -1)
    half_sum = _probe(n_part, N_mesh, L, n0, device).eval_field(state[:n_part].reshape(1, n_part), ext, fields=False)[2]
    return float(half_sum[0])


def input_energy(actions, L):
    return np.sum(np.asarray(actions) ** 2) * L * 0.25


def _clipped(value, scale):
    return max(1.0 - value / scale, 0)


def _tanh_score(value, scale):
    return np.tanh(1 - np.sqrt(value / scale))


class Reward(ParamMixin):
    def __init__(self, init_state, N_mesh=500, L=50.0, vmin=-25.0, vmax=25.0, n0=1.0, alpha=1.0, beta=1.0,
                 n_actions=10, device=0):
        self.init_state = init_state
        self.N_mesh, self.L, self.n0 = N_mesh, L, n0
        self.vmin, self.vmax = vmin, vmax
        self.alpha, self.beta, self.n_actions = alpha, beta, n_actions
        self.device = device
        self.r_pe_n = 1.0
        self.r_ie_n = input_energy(np.ones(n_actions), L)
        self.reinit()

    def reinit(self):
        """Recompute the reference distribution of the KL diagnostic from ``init_state``."""
        self.feq = estimate_f(self.init_state, self.N_mesh, self.L, self.vmin, self.vmax, self.n0)

    # -- the three cost terms ------------------------------------------------------------------
    def compute_kl_divergence(self, state):
        f = estimate_f(state, self.N_mesh, self.L, self.vmin, self.vmax, self.n0)
        return estimate_KL_divergence(f, self.feq, self.L / self.N_mesh, (self.vmax - self.vmin) / self.N_mesh)

    def compute_electric_energy(self, state, E_external=None):
        return estimate_electric_energy(state, E_external, self.N_mesh, self.L, self.n0, self.device)

    def compute_input_energy(self, actions):
        return input_energy(actions, self.L)

    def compute_cost(self, state, action):
        return (self.compute_kl_divergence(state), self.compute_electric_energy(state),
                self.compute_input_energy(action))

    # -- rewards -------------------------------------------------------------------------------
    def compute_reward(self, state, E_external=None):
        """The trainers' reward; note that the second argument is the ACTION vector (reward.py:71-76)."""
        return self.reward_from_energy(self.compute_electric_energy(state), E_external)

    def reward_from_energy(self, pe_reward, action):
        """Same value from a field energy the step already produced (no second deposit)."""
        return (self.alpha * _clipped(float(pe_reward), self.r_pe_n)
                + self.beta * _clipped(self.compute_input_energy(action), self.r_ie_n))

    def compute_reward_kl_divergence(self, state):
        return _tanh_score(self.compute_kl_divergence(state), 25)

    def compute_reward_electric_energy(self, state, E_external=None):
        return _tanh_score(self.compute_electric_energy(state, E_external), 10.0)

    def compute_reward_input_energy(self, action):
        return _tanh_score(self.compute_input_energy(action), 50.0)
