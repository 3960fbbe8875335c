"""Function-level drop-in for src/env/solve.py as the PIC path uses it (util.py:99, pic.py:116): the periodic
3-point Poisson problem, solved on the device by two prefix scans instead of Thomas +
Sherman-Morrison on a dense matrix.

`Gaussian_Elimination_Periodic(A, B, gamma)` keeps the reference's signature.  A must be the periodic Laplacian
`generate_laplacian(L, N_mesh)` for some dx (that is the only matrix the reference ever passes); anything else
raises ValueError, as the reference does for a malformed system (solve.py:32-33).  The returned potential has zero
mean: the reference's own gauge is round-off of a singular solve, `gamma` therefore has no effect.
"""
import numpy as np

from .util import probe_handle


def solve_periodic_poisson(rhs, dx, device: int = 0):
    """phi with (phi[j+1] - 2 phi[j] + phi[j-1]) / dx^2 = rhs[j], periodic, mean(phi) = 0; and E = -grad phi.
    rhs: (N_mesh,) summing to zero (as n - n0 does)."""
    b = np.asarray(rhs, dtype=np.float64).reshape(-1)
    Ng = b.shape[0]
    # the probe's particle count and density only size buffers here; its mesh spacing is L / Ng = dx
    phi, E = probe_handle(64, Ng, dx * Ng, 1.0, "CIC", device).solve_poisson(b.reshape(1, Ng))
    return phi[0], E[0]


def _laplacian_spacing(A):
    """dx if A is the periodic 3-point Laplacian / dx^2, else ValueError."""
    A = np.asarray(A)
    if A.ndim != 2 or A.shape[0] != A.shape[1] or A.shape[0] < 4:
        raise ValueError("Gaussian_Elimination_Periodic: A must be a square matrix of size >= 4")
    Ng = A.shape[0]
    off = A[0, 1]
    if not off > 0:
        raise ValueError("Gaussian_Elimination_Periodic: A is not a periodic 3-point Laplacian")
    j = np.arange(Ng)
    model = np.zeros_like(A, dtype=np.float64)
    model[j, j] = -2.0 * off
    model[j, (j + 1) % Ng] = off
    model[j, (j - 1) % Ng] = off
    if not np.array_equal(model, A):
        raise ValueError("Gaussian_Elimination_Periodic: the device solver handles the periodic 3-point Laplacian "
                         "only (the one matrix the PIC path passes)")
    return 1.0 / np.sqrt(off)


def Gaussian_Elimination_Periodic(A: np.ndarray, B: np.ndarray, gamma: float = 5.0, device: int = 0):
    """solve.py:27-53 for A = generate_laplacian(L, N_mesh): returns phi (N_mesh,) with zero mean."""
    B = np.asarray(B, dtype=np.float64)
    if B.reshape(-1).shape[0] != np.asarray(A).shape[0]:
        raise ValueError("Gaussian_Elimination_Periodic: A and B sizes differ")
    dx = _laplacian_spacing(A)
    return solve_periodic_poisson(B, dx, device)[0]
