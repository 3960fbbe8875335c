"""Function-level drop-ins for src/env/interpolate.py (CIC, TSC), evaluated on the device: the deposit is the
LDS-mesh sweep of the step path, the indices and weights come from its `locate` (shape_query_kernel)."""
import numpy as np

from .util import _columns, probe_handle


def _deposit(x, n0, L, N, N_mesh, interpol, device):
    pos = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(1, N))
    out = probe_handle(N, N_mesh, L, n0, interpol, device).compute_E(pos, None, particles=False, shape=True)
    return out["n"][0], out["idx"][0], out["w"][0]


def CIC(x: np.ndarray, n0: float, L: float, N: int, N_mesh: int, dx: float, device: int = 0):
    """interpolate.py:4-20 -> n (N_mesh,), indx_l, indx_r (N,1) int64, weight_l, weight_r (N,1).  The input is
    not modified; positions are wrapped into [0, L) on the device as `np.mod(x, L)` does."""
    n, idx, w = _deposit(x, n0, L, N, N_mesh, "CIC", device)
    return (n, *_columns(idx[:2]), *_columns(w[:2]))


def TSC(x: np.ndarray, n0: float, L: float, N: int, N_mesh: int, dx: float, device: int = 0):
    """interpolate.py:22-44 -> n, indx_l, indx_m, indx_r, weight_l, weight_m, weight_r."""
    n, idx, w = _deposit(x, n0, L, N, N_mesh, "TSC", device)
    return (n, *_columns(idx), *_columns(w))
