"""`PIC` -- drop-in for the reference environment ``src/env/pic.py:11-223`` whose step runs on
the MI355X through libpicstep.so.

Same constructor keywords, methods and attribute names/shapes as the reference
class, so ``run_wo_oc.py`` style loops and ``src/control/rl/{ddpg,ppo,sac}.train``
drive it unchanged: ``update_state(E_external)`` (= step), ``reinit()`` (= reset),
``get_state()``, ``get_energy()``, ``get_electric_energy()``, ``simulate()``,
``update_params()``, ``x, v, n, E, E_mesh, phi_mesh, indx_l/r, weight_l/r, dt, dx,
N, N_mesh, L, n0, init_dist``.  Added: ``step()/reset()`` Gym-style aliases.

State lives on the device; the array attributes are host copies fetched lazily
and cached until the next step.  ``phi_mesh`` is returned in the zero-mean gauge
(the reference's gauge is a round-off artefact of its singular Sherman-Morrison
solve; ``E_mesh`` is gauge free -- DESIGN.md).
"""
from typing import Callable, List, Optional

import numpy as np

from .. import _abi


class PIC:
    np.random.seed(42)   # the reference seeds the global RNG when its class body runs (pic.py:12)

    def __init__(self, N: int = 40000, N_mesh: int = 400, n0: float = 1.0, L: float = 50.0, dt: float = 1.0,
                 tmin: float = 0.0, tmax: float = 50.0, gamma: float = 5.0, A: float = 0.1, n_mode: int = 4,
                 interpol: str = "CIC", init_dist=None, device: int = 0, dtype="float64", position_dtype=None):
        self.N = N
        self.N_mesh = N_mesh
        self.n0 = n0
        self.L = L
        self.dt = dt
        self.tmin = tmin
        self.tmax = tmax
        self.dx = L / N_mesh
        self.gamma = gamma
        self.A = A
        self.n_mode = n_mode
        self.init_dist = init_dist
        self.interpol = interpol
        self.device = device
        self.dtype = np.dtype(dtype)
        self.position_dtype = position_dtype          # None / "float", or "fixed32" with dtype="float32" 
        self._handle = None
        self._handle_key = None
        self._cache = {}
        self._fields_hidden = False
        self.initialize()

    # -- device plumbing ---------------------------------------------------------------------
    def _key(self):
        return (self.N, self.N_mesh, float(self.L), float(self.n0), float(self.dt), self.interpol, self.device,
                str(self.dtype), self.position_dtype)

    def _ensure_handle(self):
        key = self._key()
        if self._handle is None or key != self._handle_key:
            carry = None
            if self._handle is not None:
                if key[0] == self._handle_key[0]:
                    carry = self._handle.particles()
                self._handle.close()
            self._handle = _abi.Handle(self.N, self.N_mesh, 1, self.L, self.n0, self.dt, self.gamma, self.dtype,
                                       None, self.interpol, self.device, position_dtype=self.position_dtype)
            self._handle_key = key
            if carry is not None:
                self._handle.reset(*carry)
        return self._handle

    def _invalidate(self):
        self._cache = {}

    def _particles(self):
        if "x" not in self._cache:
            x, v = self._ensure_handle().particles()
            self._cache["x"] = np.asarray(x, dtype=np.float64).reshape(-1, 1)
            self._cache["v"] = np.asarray(v, dtype=np.float64).reshape(-1, 1)
        return self._cache["x"], self._cache["v"]

    def _fields(self):
        if "n" not in self._cache:
            n, E, phi = self._ensure_handle().fields()
            self._cache["n"] = n[0].copy()
            self._cache["E_mesh"] = E[0].reshape(-1, 1).copy()
            self._cache["phi_mesh"] = phi[0].reshape(-1, 1).copy()
        return self._cache

    # -- attributes of the reference object --------------------------------------------------
    @property
    def x(self):
        return self._particles()[0]

    @x.setter
    def x(self, value):
        _, v = self._particles()
        self._load(np.asarray(value, dtype=float).reshape(-1), v.reshape(-1))

    @property
    def v(self):
        return self._particles()[1]

    @v.setter
    def v(self, value):
        x, _ = self._particles()
        self._load(x.reshape(-1), np.asarray(value, dtype=float).reshape(-1))

    def _load(self, x, v):
        self._ensure_handle().reset(x, v)
        self._invalidate()

    # Dense mesh operators of the reference object (pic.py:52-53).  The device path has no use for them (its field
    # solve is two scans); they are built on first access for callers that read the attributes.
    @property
    def grad(self):
        from .util import generate_grad
        key = ("grad", float(self.L), self.N_mesh)
        if getattr(self, "_grad_key", None) != key:
            self._grad, self._grad_key = generate_grad(self.L, self.N_mesh), key
        return self._grad

    @property
    def laplacian(self):
        from .util import generate_laplacian
        key = ("lap", float(self.L), self.N_mesh)
        if getattr(self, "_lap_key", None) != key:
            self._lap, self._lap_key = generate_laplacian(self.L, self.N_mesh), key
        return self._lap

    @property
    def n(self):
        return self._fields()["n"]

    @property
    def E_mesh(self):
        return None if self._fields_hidden else self._fields()["E_mesh"]

    @property
    def phi_mesh(self):
        return None if self._fields_hidden else self._fields()["phi_mesh"]

    @property
    def E(self):
        if self._fields_hidden:
            return None
        if "E" not in self._cache:
            self._cache["E"] = self._ensure_handle().gather_E()[0].astype(np.float64).reshape(-1, 1)
        return self._cache["E"]

    def _shape(self):
        """indx_* / weight_* of the current particles as update_density leaves them (pic.py:93-112):
        (l, r) for CIC, (l, m, r) for TSC, each an (N, 1) column."""
        if "shape" not in self._cache:
            h = self._ensure_handle()
            if self.interpol == "CIC":
                jl, jr, wl, wr = h.cic(0)
                cols = {"indx_l": jl, "indx_r": jr, "weight_l": wl, "weight_r": wr, "indx_m": None, "weight_m": None}
            else:
                out = h.compute_E(self._particles()[0], None, particles=False, shape=True)
                idx, w = out["idx"][0], out["w"][0]
                cols = {"indx_l": idx[0], "indx_m": idx[1], "indx_r": idx[2],
                        "weight_l": w[0], "weight_m": w[1], "weight_r": w[2]}
            self._cache["shape"] = {k: (None if a is None else a.reshape(-1, 1)) for k, a in cols.items()}
        return self._cache["shape"]

    indx_l = property(lambda self: self._shape()["indx_l"])
    indx_m = property(lambda self: self._shape()["indx_m"])
    indx_r = property(lambda self: self._shape()["indx_r"])
    weight_l = property(lambda self: self._shape()["weight_l"])
    weight_m = property(lambda self: self._shape()["weight_m"])
    weight_r = property(lambda self: self._shape()["weight_r"])

    # -- reference methods -------------------------------------------------------------------
    def initialize(self):
        """pic.py:63-77: fresh sample, velocity perturbation, CFL clamp, density + field."""
        self.init_dist.reinit()
        x, v = self.init_dist.get_sample()
        x = x.reshape(-1, 1)
        v = v.reshape(-1, 1)
        v *= (1 + self.A * np.sin(2 * np.pi * self.n_mode * x / self.L))
        if self.dt > 2 / np.sqrt(self.N / self.L):
            self.dt = 2 / np.sqrt(self.N / self.L)
            print("CFL condtion invalid: change dt = {:.4f}".format(self.dt))
        self._ensure_handle().reset(x, v)     # update_density + update_E_field on the device
        self._invalidate()
        self._fields_hidden = False

    def update_params(self, **kwargs):
        for key in kwargs.keys():
            if hasattr(self, key) is True and kwargs[key] is not None:
                setattr(self, key, kwargs[key])
        self.dx = self.L / self.N_mesh if ("L" in kwargs or "N_mesh" in kwargs) else self.dx

    def reinit(self):
        """pic.py:84-91: re-initialise; E, E_mesh, phi_mesh read None until the next step."""
        self.initialize()
        self._fields_hidden = True

    def update_density(self):
        self._ensure_handle().refresh()
        self._invalidate()

    def update_E_field(self):
        self._ensure_handle().refresh()
        self._invalidate()
        self._fields_hidden = False

    def compute_state_gradient(self, eta: np.ndarray, E_external: Optional[np.ndarray] = None):
        """pic.py:125-129: d/dt [x; v] = [v; -E(x)] for an arbitrary (2N, 1) state.  `update_state` does not go
        through this (the sub-stages are fused into the sweeps); it is here for callers that drive their own
        integrator.  Like the reference it wraps eta[:N] in place; the field comes from the device."""
        from .util import compute_E
        force = compute_E(eta, self.dx, self.N_mesh, self.n0, self.L, self.N, None, None, False, self.interpol,
                          E_external, device=self.device)[0]
        return np.concatenate([eta[self.N:, :], -force], axis=0)

    def update_state(self, E_external: Optional[np.ndarray] = None):
        """pic.py:131-146: one Yoshida-4 step with an optional external mesh field (Ng,1)."""
        h = self._ensure_handle()
        if E_external is not None:
            E_external = np.asarray(E_external, dtype=np.float64).reshape(-1)
            if E_external.shape[0] != self.N_mesh:
                raise ValueError("E_external must have N_mesh entries")
        h.step(E_external, 1)
        self._invalidate()
        self._fields_hidden = False

    def update_state_w_input_func(self, input_func: Optional[Callable]):
        """pic.py:148-163: a step whose external field is `input_func(eta)` of the sub-stage state.  The three
        force evaluations whose result the integrator uses happen at [q1; p0], [q2; p1], [q3; p2]
        (integration.py:32); `input_func` is called at exactly those states (the reference also calls it for the
        evaluations whose field it discards).  Each call costs a host round trip of the state, by construction."""
        if input_func is None:
            return self.update_state(None)
        h = self._ensure_handle()
        x, v = self._particles()
        c1 = 0.5 * (1 / (2 - 2 ** (1 / 3)))                        # integration.py:62-66, same expression order
        eta = np.concatenate([x.reshape(-1, 1) + c1 * v.reshape(-1, 1) * self.dt, v.reshape(-1, 1)], axis=0)
        for stage in (1, 2, 3):
            field = input_func(eta)
            h.step_stage(stage, None if field is None else np.asarray(field, dtype=np.float64).reshape(-1))
            if stage < 3:
                xs, vs = h.particles()
                eta = np.concatenate([xs[0].astype(np.float64).reshape(-1, 1), vs[0].astype(np.float64).reshape(-1, 1)], axis=0)
        self._invalidate()
        self._fields_hidden = False

    def get_state(self):
        x, v = self._particles()
        return np.concatenate([x, v], axis=0)             # a fresh (2N, 1) array, as the reference returns

    def _energies(self):
        """(KE, PE, PE_reward) of the current state: one small device read per step, then cached."""
        if "energies" not in self._cache:
            ke, pe, per = self._ensure_handle().energies()
            self._cache["energies"] = (float(ke[0]), float(pe[0]), float(per[0]))
        return self._cache["energies"]

    def get_energy(self):
        ke, pe, _ = self._energies()
        return ke + pe

    def get_electric_energy(self):
        return self._energies()[1]

    def get_kinetic_energy(self):
        return self._energies()[0]

    def get_reward_electric_energy(self):
        """0.5*sum(E_mesh^2)*dx of the current state = Reward.compute_electric_energy(get_state())."""
        return self._energies()[2]

    def simulate(self, E_external_traj: Optional[List[np.ndarray]] = None):
        """pic.py:175-223: returns snapshot (2N, Nt+1), E (Nt+1,), PE (Nt+1,).  The Nt steps run back to back on the device --
        under E_external_traj[i] in step i when a trajectory is given (pic_step_ext_traj), else free (pic_step_snapshots) --
        and particles and energies of every step are read back once at the end, in chunks of at most ~256 MB."""
        Nt = int(np.ceil((self.tmax - self.tmin) / self.dt))
        pos, vel, Es, PEs = [self.x.copy()], [self.v.copy()], [self.get_energy()], [self.get_electric_energy()]
        if E_external_traj is None:
            h = self._ensure_handle()
            chunk = max(1, min(Nt, int(256e6 // (2 * self.N * self.dtype.itemsize))))
            done = 0
            while done < Nt:
                k = min(chunk, Nt - done)
                x, v, ke, pe, _ = h.step_snapshots(None, k)
                pos.append(np.asarray(x[:, 0, :], dtype=np.float64).T)
                vel.append(np.asarray(v[:, 0, :], dtype=np.float64).T)
                Es.extend((ke[:, 0] + pe[:, 0]).tolist())
                PEs.extend(pe[:, 0].tolist())
                done += k
            self._invalidate()
            self._fields_hidden = False
        else:
            # every step under its own field, still back to back on the device (pic_step_ext_traj)
            h = self._ensure_handle()
            traj = np.stack([np.asarray(E_external_traj[i], dtype=np.float64).reshape(self.N_mesh) for i in range(Nt)])
            chunk = max(1, min(Nt, int(256e6 // (2 * self.N * self.dtype.itemsize))))
            done = 0
            while done < Nt:
                k = min(chunk, Nt - done)
                x, v, ke, pe, _ = h.step_ext_traj(traj[done:done + k, None, :], snapshots=True)
                pos.append(np.asarray(x[:, 0, :], dtype=np.float64).T)
                vel.append(np.asarray(v[:, 0, :], dtype=np.float64).T)
                Es.extend((ke[:, 0] + pe[:, 0]).tolist())
                PEs.extend(pe[:, 0].tolist())
                done += k
            self._invalidate()
            self._fields_hidden = False
        snapshot = np.concatenate([np.concatenate(pos, axis=1), np.concatenate(vel, axis=1)], axis=0)
        return snapshot, np.array(Es), np.array(PEs)

    # -- Gym-style aliases (north star) ------------------------------------------------------
    def reset(self):
        self.reinit()
        return self.get_state()

    def set_actuator(self, actuator):
        """Attach an `E_field` (src/control/actuator.py): `step(action)` then takes the 2*max_mode coefficient
        vector the policies emit and builds E_external on the device (pic_step_actions)."""
        self._actuator = actuator
        self._actuator_key = None

    def step(self, E_external: Optional[np.ndarray] = None):
        """-> (obs, reward, done, info); reward = max(1 - PE_r, 0) of the PRE-step state, i.e. the
        electric-energy term of Reward.compute_reward (src/control/rl/reward.py:72) as the trainers
        evaluate it (ddpg.py:455).  The argument is the mesh field E_external (N_mesh values) or, after
        `set_actuator`, an action of 2*max_mode Fourier coefficients (cos, then sin: actuator.py:54-63)."""
        pe_pre = self.get_reward_electric_energy()
        h = self._ensure_handle()
        act = getattr(self, "_actuator", None)
        field = action = None
        if E_external is not None and act is not None and np.size(E_external) == 2 * act.max_mode != self.N_mesh:
            if getattr(self, "_actuator_key", None) != id(h):       # a re-created handle needs the tables again
                h.set_actuator(act.basis_cos, act.basis_sin)
                self._actuator_key = id(h)
            action = np.asarray(E_external, dtype=np.float64).reshape(1, -1)
        elif E_external is not None:
            field = np.asarray(E_external, dtype=np.float64).reshape(-1)
            if field.shape[0] != self.N_mesh:
                raise ValueError("E_external must have N_mesh entries")
        # step, observation and energies in one call with one synchronisation (pic_step_observe)
        if h.dtype == np.float64:      # the observation is written where it is returned: x | v halves of one fresh (2N, 1) array
            obs = np.empty((2 * self.N, 1))
            x, v = obs[:self.N], obs[self.N:]
            _, _, ke, pe, per = h.step_observe(field, action, 1, out=(x, v))
        else:
            x, v, ke, pe, per = h.step_observe(field, action, 1)
            x, v = x.astype(np.float64).reshape(-1, 1), v.astype(np.float64).reshape(-1, 1)
            obs = np.concatenate([x, v], axis=0)
        self._invalidate()
        self._fields_hidden = False
        # (x and v are NOT cached from `obs`: the caller owns that array and may normalise it in place; a later read of
        # PIC.x goes to the device)
        self._cache["energies"] = (float(ke[0]), float(pe[0]), float(per[0]))
        info = {"KE": self._cache["energies"][0], "PE": self._cache["energies"][1], "PE_reward": self._cache["energies"][2]}
        return obs, max(1.0 - pe_pre, 0.0), False, info

    def close(self):
        if self._handle is not None:
            self._handle.close()
            self._handle = None
