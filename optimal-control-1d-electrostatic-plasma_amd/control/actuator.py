"""Actuator: Fourier coefficients of the control -> external electric field on the mesh.

Public surface of the reference's `E_field` (``src/control/actuator.py:4-63``): attributes ``L,
N_mesh, dx, max_mode, xm, k, basis_cos, basis_sin, coeff_cos, coeff_sin`` and methods ``reinit,
update_E, compute_E, update_params``.  The field is

    E_ext(x_j) = sum_m a_m cos(k_m x_j) + b_m sin(k_m x_j),   k_m = 2 pi m / L,  m = 1..max_mode

on the mesh ``x_j = linspace(0, L, N_mesh)[j]`` -- end point INCLUDED (actuator.py:13), i.e. the
nodes are ``j L / (N_mesh - 1)``, not the PIC mesh ``j dx``.  That quirk is kept on purpose: fields
computed here equal the reference's bit for bit (tests/golden/g8_actuator.npz).

Host side, for one environment.  For batches, ``compute_E_batched`` does the same product for
``[num_envs, 2 max_mode]`` actions, and ``BatchedPIC.set_actuator(self)`` uploads the two basis
tables so that ``BatchedPIC.step_actions`` builds ``E_ext`` on the device.
"""
import numpy as np

from .._params import ParamMixin


def fourier_basis(L, N_mesh, max_mode):
    """(xm, k, cos table, sin table); tables are [N_mesh, max_mode] with entry cos/sin(k_m * xm_j)."""
    xm = np.linspace(0, L, N_mesh)
    k = np.array([2 * np.pi / L * m for m in range(1, max_mode + 1)])
    phase = np.multiply.outer(xm, k)          # xm_j * k_m, the product the reference forms as k * xm
    return xm, k, np.cos(phase), np.sin(phase)


def _column(values):
    return np.array(values, dtype=float).reshape(-1, 1)


class E_field(ParamMixin):
    def __init__(self, L: float, N_mesh: int, max_mode: int):
        self.L, self.N_mesh, self.max_mode = L, N_mesh, max_mode
        self.dx = L / N_mesh
        self.reinit()

    def reinit(self):
        """Rebuild mesh and tables from the current L / N_mesh / max_mode and zero the coefficients."""
        self.xm, self.k, self.basis_cos, self.basis_sin = fourier_basis(self.L, self.N_mesh, self.max_mode)
        self.coeff_cos = np.zeros((self.max_mode, 1))
        self.coeff_sin = np.zeros((self.max_mode, 1))

    def update_E(self, coeff_cos=None, coeff_sin=None):
        """Store new coefficients (copies, as columns); None leaves that half unchanged."""
        if coeff_cos is not None:
            self.coeff_cos = _column(coeff_cos)
        if coeff_sin is not None:
            self.coeff_sin = _column(coeff_sin)

    def compute_E(self, coeff_cos=None, coeff_sin=None):
        """External field as an ``(N_mesh, 1)`` column, from the given or the stored coefficients."""
        a = self.coeff_cos if coeff_cos is None else _column(coeff_cos)
        b = self.coeff_sin if coeff_sin is None else _column(coeff_sin)
        return self.basis_cos @ a + self.basis_sin @ b

    def compute_E_batched(self, actions):
        """``[num_envs, 2 max_mode]`` actions (cos half, then sin half) -> ``[num_envs, N_mesh]``."""
        act = np.asarray(actions, dtype=float)
        m = self.max_mode
        return act[:, :m] @ self.basis_cos.T + act[:, m:] @ self.basis_sin.T
