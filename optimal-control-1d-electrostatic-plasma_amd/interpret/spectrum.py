"""`compute_E_k_spectrum` with the signature of ``src/interpret/spectrum.py:4-27``.

For every snapshot column the mesh field comes from the device (``pic_eval_field``: deposit ->
scan solve); the N_mesh-point FFT of the few columns stays in NumPy.  The trainers use rows
``1..max_mode`` of the complex result as the feedback / behaviour-cloning action
(``src/control/rl/ddpg.py:369-371``).
"""
import numpy as np

from ..control.reward import _probe


def compute_E_k_spectrum(n0, L, dx, N_mesh, snapshot, return_abs=True, device=0):
    snapshot = np.asarray(snapshot, dtype=np.float64)
    N = snapshot.shape[0] // 2
    Nt = snapshot.shape[1]
    h = _probe(N, int(N_mesh), L, n0, device)
    cols = [h.eval_field(snapshot[:N, i].reshape(1, N))[1][0] for i in range(Nt)]
    E_mesh_t = np.stack(cols, axis=1)                       # (N_mesh, Nt)
    Ek_t = np.fft.fft(E_mesh_t, axis=0) / N_mesh * 2.0
    ks = np.fft.fftfreq(int(N_mesh), d=dx) * 2.0 * np.pi
    spec = np.abs(Ek_t) if return_abs else Ek_t
    mask = ks >= 0
    return ks[mask], spec[mask, :]
