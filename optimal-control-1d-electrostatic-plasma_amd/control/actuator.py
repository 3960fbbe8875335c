"""`E_field` -- the actuator of ``src/control/actuator.py:4-63``: Fourier coefficients -> external
field on the mesh, ``E_ext = basis_cos @ a + basis_sin @ b`` with shape ``(N_mesh, 1)``.

Kept on the host (an ``Ng x 2M`` product with M <= 5); it produces the ``E_external`` that
``PIC.update_state`` hands to the device.  The mesh is ``linspace(0, L, N_mesh)`` with the end
point INCLUDED (actuator.py:13), not ``j*dx`` -- preserved on purpose.
"""
from typing import Optional

import numpy as np


class E_field:
    def __init__(self, L: float, N_mesh: int, max_mode: int):
        self.L = L
        self.N_mesh = N_mesh
        self.dx = L / N_mesh
        self.max_mode = max_mode
        self.reinit()

    def update_params(self, **kwargs):
        for key, val in kwargs.items():
            if hasattr(self, key) and val is not None:
                setattr(self, key, val)

    def reinit(self):
        self.xm = np.linspace(0, self.L, self.N_mesh)
        self.coeff_cos = np.zeros((self.max_mode, 1))
        self.coeff_sin = np.zeros((self.max_mode, 1))
        self.k = np.array([2 * np.pi / self.L * m for m in range(1, self.max_mode + 1)])
        phase = self.xm.reshape(-1, 1) * self.k.reshape(1, -1)     # (Ng, M): k * xm per column
        self.basis_cos = np.cos(phase)
        self.basis_sin = np.sin(phase)

    def update_E(self, coeff_cos: Optional[np.ndarray] = None, coeff_sin: Optional[np.ndarray] = None):
        if coeff_cos is not None:
            self.coeff_cos = np.array(coeff_cos, dtype=float).reshape(-1, 1)
        if coeff_sin is not None:
            self.coeff_sin = np.array(coeff_sin, dtype=float).reshape(-1, 1)

    def compute_E(self, coeff_cos: Optional[np.ndarray] = None, coeff_sin: Optional[np.ndarray] = None):
        cc = self.coeff_cos if coeff_cos is None else np.asarray(coeff_cos, dtype=float)
        cs = self.coeff_sin if coeff_sin is None else np.asarray(coeff_sin, dtype=float)
        return self.basis_cos @ cc.reshape(-1, 1) + self.basis_sin @ cs.reshape(-1, 1)

    def compute_E_batched(self, actions: np.ndarray):
        """actions [num_envs, 2M] (cos coefficients then sin) -> E_ext [num_envs, N_mesh]."""
        a = np.asarray(actions, dtype=float)
        M = self.max_mode
        return a[:, :M] @ self.basis_cos.T + a[:, M:] @ self.basis_sin.T
