"""Function-level drop-ins for src/env/util.py: same names, argument order and return shapes, evaluated by the
HIP library (deposit sweep -> scan Poisson solve -> gather) through a small cache of single-environment probe
handles.  There is no NumPy evaluation path here; without libpicstep.so and a GPU these raise PicError.

Differences from the reference, all documented in DESIGN.md: `phi` / `phi_mesh` come in the mean-zero gauge (the
reference's gauge is round-off of a singular solve), `gamma`-style arguments have no effect, and the dense
`grad` / `laplacian` arguments of compute_E are accepted for signature compatibility and not used.
"""
from typing import Optional

import numpy as np

from .. import _abi

import threading

# One cache PER THREAD: calls on a handle are not re-entrant (include/picstep.h), and two threads evaluating fields of states of
# the same shape would otherwise meet on one probe handle.  Handles own device memory, so each cache stays small.
_local = threading.local()
_MAX_PROBES = 8


def _probe_cache():
    cache = getattr(_local, "probes", None)
    if cache is None:
        cache = _local.probes = {}
    return cache


def probe_handle(N, N_mesh, L, n0, interpol="CIC", device=0):
    """Single-environment float64 handle used only to evaluate fields of caller-supplied positions (this thread's own)."""
    key = (int(N), int(N_mesh), float(L), float(n0), str(interpol), int(device))
    probes = _probe_cache()
    h = probes.get(key)
    if h is None:
        while len(probes) >= _MAX_PROBES:
            probes.pop(next(iter(probes))).close()
        h = _abi.Handle(key[0], key[1], 1, key[2], key[3], 1.0, 5.0, "float64", None, key[4], key[5])
        probes[key] = h
    return h


def _shift_matrix(N_mesh, k):
    """S with S[i, (i + k) mod N_mesh] = 1."""
    return np.roll(np.eye(N_mesh), k, axis=1)


def generate_grad(L: float, N_mesh: int):
    """Dense periodic central difference / (2 dx) (util.py:7-26).  Not used by the device path."""
    dx = L / N_mesh
    grad = _shift_matrix(N_mesh, 1) - _shift_matrix(N_mesh, -1)
    grad /= 2 * dx
    return grad


def generate_laplacian(L: float, N_mesh: int):
    """Dense periodic 3-point Laplacian / dx**2 (util.py:28-46).  Not used by the device path."""
    dx = L / N_mesh
    lap = _shift_matrix(N_mesh, 1) + _shift_matrix(N_mesh, -1) - 2.0 * np.eye(N_mesh)
    lap /= dx ** 2
    return lap


def _positions(u, N):
    """compute_n's in-place side effect (util.py:51): u[:N] = mod(u[:N], L) is applied by the callers below;
    this returns the (N,) float64 view handed to the device."""
    return np.ascontiguousarray(np.asarray(u[:N], dtype=np.float64).reshape(1, N))


def _columns(a):
    return [row.reshape(-1, 1) for row in a]


def compute_n(u: np.ndarray, dx: float, N_mesh: int, n0: float, L: float, N: int, return_all: bool = False,
              interpol: str = "CIC", device: int = 0):
    """util.py:48-70: wrap u[:N] in place, deposit with CIC or TSC.  -> n (N_mesh,), or with return_all the
    index and weight columns as well ((N, 1) int64 / float64; l, r for CIC and l, m, r for TSC)."""
    u[:N] = np.mod(u[:N], L)
    out = probe_handle(N, N_mesh, L, n0, interpol, device).compute_E(_positions(u, N), None, particles=False,
                                                                       shape=return_all)
    n = out["n"][0]
    if not return_all:
        return n
    rows = 2 if interpol == "CIC" else 3
    return (n, *_columns(out["idx"][0][:rows]), *_columns(out["w"][0][:rows]))


def compute_E(u: np.ndarray, dx: float, N_mesh: int, n0: float, L: float, N: int, grad: Optional[np.ndarray] = None,
              laplacian: Optional[np.ndarray] = None, return_all: bool = False, interpol: str = "CIC",
              E_external: Optional[np.ndarray] = None, device: int = 0):
    """util.py:73-116: deposit -> periodic Poisson solve -> E_mesh = -grad phi (+ E_external) -> gather at the
    particles.  -> (E (N,1), E_mesh (N_mesh,1)) or, with return_all, (E, phi, E_mesh, phi_mesh)."""
    u[:N] = np.mod(u[:N], L)
    out = probe_handle(N, N_mesh, L, n0, interpol, device).compute_E(_positions(u, N), E_external, particles=True)
    E, E_mesh = out["E"][0].reshape(-1, 1), out["E_mesh"][0].reshape(-1, 1)
    if return_all:
        return E, out["phi"][0].reshape(-1, 1), E_mesh, out["phi_mesh"][0].reshape(-1, 1)
    return E, E_mesh


def compute_electric_energy(x: np.ndarray, dx: float, N: int, N_mesh: int, n0: float, L: float, interpol: str = "CIC",
                            device: int = 0):
    """util.py:119-131: 0.5 * sum(E_mesh^2) * dx * N / L, reduced on the device."""
    x[:N] = np.mod(x[:N], L)
    half_sum = probe_handle(N, N_mesh, L, n0, interpol, device).eval_field(_positions(x, N), None, fields=False)[2]
    return float(half_sum[0]) * (N / L)


def compute_hamiltonian(x: np.ndarray, v: np.ndarray, dx: float, N: int, N_mesh: int, n0: float = 1.0, L: float = 50.0,
                        interpol: str = "CIC", device: int = 0):
    """util.py:133-147: kinetic + electric energy."""
    kinetic = 0.5 * np.sum(v * v)
    return kinetic + compute_electric_energy(x, dx, N, N_mesh, n0, L, interpol, device)
