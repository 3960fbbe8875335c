"""Fourier spectrum of the self-consistent field along a trajectory.

`compute_E_k_spectrum` keeps the call signature and return convention of the reference
(``src/interpret/spectrum.py:4-27``): for a snapshot matrix whose columns are states ``[x; v]`` it
returns the non-negative wavenumbers and ``fft(E_mesh) / N_mesh * 2`` per column (magnitudes unless
``return_abs=False``).  Rows ``1..max_mode`` of the complex result are what the feedback controller
and the behaviour-cloning warm start use as their action (run_feedback.py:133-135,
src/control/rl/ddpg.py:369-371).

The mesh field of each column is evaluated on the device (deposit -> scan solve, `pic_eval_field`);
only the short FFT runs in NumPy.  For states that already live on the device use
`BatchedPIC.modes()` / `feedback_actions()`, which skip the host round trip altogether.
"""
import numpy as np

from ..control.reward import _probe


def mesh_fields(snapshot, N_mesh, L, n0, device=0):
    """E_mesh for every column of a ``(2N, Nt)`` snapshot -> ``(N_mesh, Nt)``."""
    snap = np.asarray(snapshot, dtype=np.float64)
    n_part, n_t = snap.shape[0] // 2, snap.shape[1]
    probe = _probe(n_part, int(N_mesh), L, n0, device)
    out = np.empty((int(N_mesh), n_t))
    for t in range(n_t):
        out[:, t] = probe.eval_field(snap[:n_part, t].reshape(1, n_part))[1][0]
    return out


def compute_E_k_spectrum(n0, L, dx, N_mesh, snapshot, return_abs=True, device=0):
    n_mesh = int(N_mesh)
    coeff = np.fft.fft(mesh_fields(snapshot, n_mesh, L, n0, device), axis=0) / N_mesh * 2.0
    wavenumber = 2.0 * np.pi * np.fft.fftfreq(n_mesh, d=dx)
    keep = wavenumber >= 0
    spectrum = np.abs(coeff) if return_abs else coeff
    return wavenumber[keep], spectrum[keep, :]
