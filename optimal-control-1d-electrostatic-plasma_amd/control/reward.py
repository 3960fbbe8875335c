"""`Reward` with the interface of ``src/control/rl/reward.py:5-76``.

The electric-energy reduction ``0.5 * sum(E_mesh^2) * dx`` (``src/control/objective.py:20-35``: CIC
forced, no N/L factor) is evaluated on the device through ``pic_eval_field`` -- deposit the
given state's positions, solve, reduce -- instead of a NumPy ``compute_E`` with freshly built
dense matrices.  The phase-space histogram / KL diagnostic (objective.py:8-18) is a logging-only
quantity and stays a NumPy histogram on the host.
"""
from typing import Optional

import numpy as np

from .. import _abi

_EPS = 1e-12
_probe_cache = {}


def _probe(N, N_mesh, L, n0, device=0):
    """One single-environment handle per problem shape, used only for field evaluation."""
    key = (int(N), int(N_mesh), float(L), float(n0), int(device))
    h = _probe_cache.get(key)
    if h is None:
        if len(_probe_cache) > 8:
            _probe_cache.pop(next(iter(_probe_cache))).close()
        h = _abi.Handle(N, N_mesh, 1, L, n0, 1.0, 5.0, "float64", None, "CIC", device)
        _probe_cache[key] = h
    return h


def estimate_f(state, N_mesh, L, vmin, vmax, n0):
    """Phase-space density on an N_mesh x N_mesh grid (objective.py:8-14)."""
    N = state.shape[0] // 2
    dx = L / N_mesh
    dv = (vmax - vmin) / N_mesh
    hist, _, _ = np.histogram2d(state[:N].ravel(), state[N:].ravel(), bins=[N_mesh, N_mesh], density=False,
                                range=np.array([[0, L], [vmin, vmax]]))
    hist *= n0 / dx / dv / N
    return hist


def estimate_KL_divergence(f, feq, dx=0.1, dv=0.04):
    """sum rel_entr(f, feq + eps) dx dv (objective.py:16-18) without scipy."""
    q = feq + _EPS
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(f > 0, f * np.log(f / q), 0.0)
    t = np.where((f > 0) & (q <= 0), np.inf, t)
    return np.sum(t) * dx * dv


def estimate_electric_energy(state, E_external, N_mesh, L, n0, device=0):
    """0.5 * sum((E_mesh + E_ext)^2) * dx for the positions in ``state[:N]`` (objective.py:20-35)."""
    state = np.asarray(state, dtype=np.float64)
    N = state.shape[0] // 2
    ext = None if E_external is None else np.asarray(E_external, dtype=np.float64).reshape(-1)
    _, _, pe = _probe(N, N_mesh, L, n0, device).eval_field(state[:N].reshape(1, N), ext)
    return float(pe[0])


class Reward:
    def __init__(self, init_state, N_mesh=500, L=50.0, vmin=-25.0, vmax=25.0, n0=1.0, alpha=1.0, beta=1.0,
                 n_actions=10, device=0):
        self.feq = estimate_f(init_state, N_mesh, L, vmin, vmax, n0)
        self.init_state = init_state
        self.N_mesh, self.L, self.vmin, self.vmax, self.n0 = N_mesh, L, vmin, vmax, n0
        self.n_actions = n_actions
        self.alpha, self.beta = alpha, beta
        self.device = device
        self.r_pe_n = 1.0                                                    # reward.py:32
        self.r_ie_n = self.compute_input_energy(np.ones(n_actions))          # reward.py:33

    def update_params(self, **kwargs):
        for key, val in kwargs.items():
            if hasattr(self, key) and val is not None:
                setattr(self, key, val)

    def reinit(self):
        self.feq = estimate_f(self.init_state, self.N_mesh, self.L, self.vmin, self.vmax, self.n0)

    def compute_kl_divergence(self, state):
        f = estimate_f(state, self.N_mesh, self.L, self.vmin, self.vmax, self.n0)
        return estimate_KL_divergence(f, self.feq, self.L / self.N_mesh, (self.vmax - self.vmin) / self.N_mesh)

    def compute_electric_energy(self, state, E_external: Optional[np.ndarray] = None):
        return estimate_electric_energy(np.asarray(state).reshape(-1, 1), E_external, self.N_mesh, self.L, self.n0,
                                        self.device)

    def compute_input_energy(self, actions):
        return np.sum(np.asarray(actions) ** 2) * self.L * 0.25

    def compute_cost(self, state, action):
        return self.compute_kl_divergence(state), self.compute_electric_energy(state), self.compute_input_energy(action)

    def compute_reward_kl_divergence(self, state):
        return np.tanh(1 - np.sqrt(self.compute_kl_divergence(state) / 25))

    def compute_reward_electric_energy(self, state, E_external=None):
        return np.tanh(1 - np.sqrt(self.compute_electric_energy(state, E_external) / 10.0))

    def compute_reward_input_energy(self, action):
        return np.tanh(1 - np.sqrt(self.compute_input_energy(action) / 50.0))

    def compute_reward(self, state, E_external=None):
        """reward.py:71-76 -- note the second argument is the ACTION vector there."""
        r_pe = max(1.0 - self.compute_electric_energy(state) / self.r_pe_n, 0)
        r_ie = max(1.0 - self.compute_input_energy(E_external) / self.r_ie_n, 0)
        return r_pe * self.alpha + r_ie * self.beta

    def reward_from_energy(self, pe_reward, action):
        """Same value from a PE_reward the step already produced (no second deposit)."""
        r_pe = max(1.0 - float(pe_reward) / self.r_pe_n, 0)
        r_ie = max(1.0 - self.compute_input_energy(action) / self.r_ie_n, 0)
        return r_pe * self.alpha + r_ie * self.beta
