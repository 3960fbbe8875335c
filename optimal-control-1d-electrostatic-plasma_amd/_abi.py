"""ctypes binding of include/picstep.h (the only way Python reaches the HIP path).

There is no CPU fallback: if libpicstep.so is missing or no HIP device is
visible, loading / creating a handle raises.
"""
import ctypes as C
import os

import numpy as np

from . import _build

PIC_F64, PIC_F32 = 0, 1
PIC_POS_FLOAT, PIC_POS_FIXED32 = 0, 1
PIC_ACC_AUTO, PIC_ACC_FIX64, PIC_ACC_PACKED, PIC_ACC_F64 = 0, 1, 2, 3
PIC_CIC, PIC_TSC = 0, 1
PIC_HOST, PIC_DEVICE = 0, 1
ABI_VERSION = 4

# accum_dtype spellings of the Python layer -> PIC_ACC_*
ACCUMULATORS = {None: PIC_ACC_AUTO, "auto": PIC_ACC_AUTO, "fix64": PIC_ACC_FIX64, "fixed": PIC_ACC_PACKED,
                "packed": PIC_ACC_PACKED, "float64": PIC_ACC_F64}
POSITION_FORMATS = {None: PIC_POS_FLOAT, "float": PIC_POS_FLOAT, "fixed32": PIC_POS_FIXED32}

KIND_NAMES = ("sweep_A", "sweep_B", "sweep_C", "sweep_D", "field_solve", "sweep_aux", "resident", "")


class PicConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int64), ("Ng", C.c_int32), ("num_envs", C.c_int32),
        ("L", C.c_double), ("n0", C.c_double), ("dt", C.c_double), ("gamma", C.c_double),
        ("particle_dtype", C.c_int32), ("accum_dtype", C.c_int32), ("interpol", C.c_int32),
        ("device_id", C.c_int32), ("blocks_per_env", C.c_int32), ("env_index_base", C.c_int32),
        ("position_dtype", C.c_int32), ("placement", C.c_int32), ("placement_ms", C.c_int32),
    ]


class PicError(RuntimeError):
    pass


_lib = None

# name -> argtypes; every entry point of include/picstep.h is listed (tests check the exports)
_vp, _dp, _i64p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)
SIGNATURES = {
    "pic_create": [C.POINTER(PicConfig), C.POINTER(_vp)],
    "pic_destroy": [_vp],
    "pic_reset": [_vp, _vp, _vp, C.c_int],
    "pic_reset_sampled": [_vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint64],
    "pic_step": [_vp, _vp, C.c_int, C.c_int],
    "pic_step_stage": [_vp, C.c_int, _vp, C.c_int],
    "pic_step_history": [_vp, _vp, C.c_int, C.c_int, _vp],
    "pic_step_snapshots": [_vp, _vp, C.c_int, C.c_int, _vp, _vp],
    "pic_get_particles": [_vp, _vp, _vp, C.c_int],
    "pic_set_particles": [_vp, _vp, _vp, C.c_int],
    "pic_refresh": [_vp],
    "pic_invalidate": [_vp],
    "pic_device_ptrs": [_vp] + [C.POINTER(_vp)] * 2 + [_i64p] + [C.POINTER(_vp)] * 6,
    "pic_get_fields": [_vp, _vp, _vp, _vp],
    "pic_get_energies": [_vp, _vp, _vp, _vp],
    "pic_gather_E": [_vp, _vp, C.c_int],
    "pic_get_cic": [_vp, C.c_int, _vp, _vp, _vp, _vp],
    "pic_eval_field": [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp],
    "pic_compute_E": [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pic_solve_poisson": [_vp, _vp, _vp, _vp],
    "pic_profile": [_vp, C.c_int],
    "pic_profile_read": [_vp, _dp, _i64p],
    "pic_set_actuator": [_vp, C.c_int, _vp, _vp],
    "pic_step_actions": [_vp, _vp, C.c_int, C.c_int],
    "pic_step_actions_traj": [_vp, _vp, C.c_int, C.c_int, _vp],
    "pic_step_ext_traj": [_vp, _vp, C.c_int, C.c_int, _vp, _vp],
    "pic_step_feedback": [_vp, C.c_int, C.c_int, _vp, _vp],
    "pic_step_observe": [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp],
    "pic_get_modes": [_vp, C.c_int, _vp, _vp, C.c_int],
    "pic_phase_histogram": [_vp, C.c_int, C.c_double, C.c_double, _vp],
    "pic_phase_kl": [_vp, C.c_int, C.c_double, C.c_double, _vp, _vp],
    "pic_stream_probe": [_vp, C.c_int, _dp],
    "pic_set_stream": [_vp, _vp],
    "pic_own_stream": [_vp],
    "pic_schedule": [_vp],
    "pic_placement_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "pic_placement_stats": [_vp, _vp],
    "pic_sync": [_vp],
    "pic_bad_count": [_vp, _i64p],
    "pic_last_error": [_vp],
    "pic_abi_version": [],
}


class _Placement(C.Structure):
    _fields_ = [("pairs_timed", C.c_int32), ("blocks", C.c_int32), ("outcome", C.c_int32), ("legs", C.c_int32),
                ("kept_gbytes_per_s", C.c_double), ("slowest_gbytes_per_s", C.c_double), ("seconds", C.c_double),
                ("malloc_seconds", C.c_double), ("timing_seconds", C.c_double), ("free_seconds", C.c_double)]


def library_path():
    """The one library this package loads: csrc/libpicstep.so next to it (no environment override)."""
    return _build.LIB


def _preload_torch_hip_runtime():
    """One HIP runtime per process: the torch wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, the name libpicstep.so needs).  Loading that copy first makes the dynamic
    loader bind libpicstep.so to it, so torch tensors and library buffers share one runtime,
    whichever of the two is imported first.  Without torch the system ROCm runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load csrc/libpicstep.so (built by __graft_entry__.build()); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path) and path == _build.LIB:
        try:                                   # a checkout without the built library: compile it now (hipcc, gfx950)
            _build.build_library()
        except RuntimeError as e:
            raise PicError(f"{path} is missing and could not be built ({e}). There is no CPU fallback for the "
                           "PIC step.") from e
    if not os.path.exists(path):
        raise PicError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc, gfx950). There is no CPU fallback for the PIC step.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_char_p if name == "pic_last_error" else C.c_int
    if lib.pic_abi_version() != ABI_VERSION:
        raise PicError("libpicstep.so ABI version mismatch: rebuild it")
    _lib = lib
    return lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    # (ndarray.ctypes costs ~1.5 us per use and a Gym iteration passes nine pointers.  The bare address keeps no reference to
    # the array: every caller passes a name, or a view of one, that outlives the call)
    return C.c_void_p(a.__array_interface__["data"][0])


class Handle:
    """Owns one pic_handle (one device, one stream, `num_envs` environments)."""

    def __init__(self, N, Ng, num_envs=1, L=50.0, n0=1.0, dt=0.1, gamma=5.0, particle_dtype="float64",
                 accum_dtype=None, interpol="CIC", device_id=0, blocks_per_env=0, env_index_base=0,
                 position_dtype=None, placement="auto", placement_ms=0):
        self.lib = load()
        pd = {"float64": PIC_F64, "float32": PIC_F32}[str(np.dtype(particle_dtype))]
        # LDS mesh accumulator (include/picstep.h PIC_ACC_*).  None: the library's choice -- the packed word for
        # float32 CIC (one integer LDS atomic per deposit), 64-bit fixed point otherwise; both are integer sums, so
        # a step is bitwise reproducible.  "float64" = float64 running sums (ds_add_f64), float64 particles only.
        key = accum_dtype if accum_dtype is None else str(accum_dtype)
        if key not in ACCUMULATORS:
            raise ValueError(f"accum_dtype must be one of {sorted(k for k in ACCUMULATORS if k)} or None, not {accum_dtype!r}")
        pkey = position_dtype if position_dtype is None else str(position_dtype)
        if pkey not in POSITION_FORMATS:
            raise ValueError(f"position_dtype must be 'float', 'fixed32' or None, not {position_dtype!r}")
        self.cfg = PicConfig(int(N), int(Ng), int(num_envs), float(L), float(n0), float(dt), float(gamma), pd,
                             ACCUMULATORS[key], {"CIC": PIC_CIC, "TSC": PIC_TSC}[interpol], int(device_id),
                             int(blocks_per_env), int(env_index_base), POSITION_FORMATS[pkey],
                             {"auto": 0, "off": 1}[placement], int(placement_ms))
        self.fixed_positions = POSITION_FORMATS[pkey] == PIC_POS_FIXED32
        self.N, self.Ng, self.num_envs = int(N), int(Ng), int(num_envs)
        self.dtype = np.dtype(particle_dtype)
        self._h = C.c_void_p()
        rc = self.lib.pic_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            msg = self.lib.pic_last_error(None)
            self._h = C.c_void_p()
            raise PicError(f"pic_create failed ({rc}): {msg.decode() if msg else ''}")

    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.pic_last_error(self._h)
            raise PicError(f"libpicstep error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.pic_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state -------------------------------------------------------------------------------
    def _particles_in(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=self.dtype).reshape(self.num_envs, self.N))
        return a

    def reset(self, x0, v0):
        x0, v0 = self._particles_in(x0), self._particles_in(v0)
        self._chk(self.lib.pic_reset(self._h, _ptr(x0), _ptr(v0), PIC_HOST))

    def reset_device(self, x_ptr, v_ptr):
        self._chk(self.lib.pic_reset(self._h, _ptr(int(x_ptr)), _ptr(int(v_ptr)), PIC_DEVICE))

    def reset_sampled(self, kind, a=0.2, v0=3.0, sigma=1.0, A=0.1, n_mode=2, seed=0):
        k = {"two-stream": 0, "bump-on-tail": 1}[kind]
        self._chk(self.lib.pic_reset_sampled(self._h, k, float(a), float(v0), float(sigma), float(A), int(n_mode),
                                             int(seed) & 0xFFFFFFFFFFFFFFFF))

    def set_particles(self, x, v):
        x, v = self._particles_in(x), self._particles_in(v)
        self._chk(self.lib.pic_set_particles(self._h, _ptr(x), _ptr(v), PIC_HOST))

    def refresh(self):
        self._chk(self.lib.pic_refresh(self._h))

    def invalidate(self):
        """Call after writing x / v through the device views (or call refresh())."""
        self._chk(self.lib.pic_invalidate(self._h))

    def step(self, E_ext=None, nsteps=1):
        if E_ext is None:
            self._chk(self.lib.pic_step(self._h, None, PIC_HOST, int(nsteps)))
        else:
            e = np.ascontiguousarray(np.asarray(E_ext, dtype=np.float64).reshape(self.num_envs, self.Ng))
            self._chk(self.lib.pic_step(self._h, _ptr(e), PIC_HOST, int(nsteps)))

    def step_device(self, E_ext_ptr, nsteps=1):
        self._chk(self.lib.pic_step(self._h, _ptr(int(E_ext_ptr)) if E_ext_ptr else None, PIC_DEVICE, int(nsteps)))

    def sync(self):
        self._chk(self.lib.pic_sync(self._h))

    def schedule(self):
        """'resident' (one launch per pic_step call, small environments) or 'streaming' (sweeps)."""
        return "resident" if self.lib.pic_schedule(self._h) == 1 else "streaming"

    def placement_info(self):
        """((x, v) placements pic_create timed, GB/s of the pair kept, GB/s of the slowest pair, seconds the search took);
        (1, 0, 0, 0) for small states and with placement="off"."""
        n, kept, slow, sec = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.pic_placement_info(self._h, C.byref(n), C.byref(kept), C.byref(slow), C.byref(sec)))
        return n.value, kept.value, slow.value, sec.value

    def placement_stats(self):
        """pic_placement_stats as a dict: pairs_timed, blocks, legs, outcome ("none" | "found" | "patience" | "timeout" | "memory"),
        kept_GBs, slowest_GBs, seconds, malloc_seconds, timing_seconds, free_seconds."""
        st = _Placement()
        self._chk(self.lib.pic_placement_stats(self._h, C.byref(st)))
        return {"pairs_timed": st.pairs_timed, "blocks": st.blocks, "legs": st.legs,
                "outcome": ("none", "found", "patience", "timeout", "memory")[st.outcome], "kept_GBs": st.kept_gbytes_per_s,
                "slowest_GBs": st.slowest_gbytes_per_s, "seconds": st.seconds, "malloc_seconds": st.malloc_seconds,
                "timing_seconds": st.timing_seconds, "free_seconds": st.free_seconds}

    def particles(self):
        x = np.empty((self.num_envs, self.N), dtype=self.dtype)
        v = np.empty_like(x)
        self._chk(self.lib.pic_get_particles(self._h, _ptr(x), _ptr(v), PIC_HOST))
        return x, v

    def fields(self):
        n = np.empty((self.num_envs, self.Ng))
        E = np.empty_like(n)
        phi = np.empty_like(n)
        self._chk(self.lib.pic_get_fields(self._h, _ptr(n), _ptr(E), _ptr(phi)))
        return n, E, phi

    def energies(self):
        ke = np.empty(self.num_envs)
        pe = np.empty_like(ke)
        per = np.empty_like(ke)
        self._chk(self.lib.pic_get_energies(self._h, _ptr(ke), _ptr(pe), _ptr(per)))
        return ke, pe, per

    def gather_E(self):
        E = np.empty((self.num_envs, self.N), dtype=self.dtype)
        self._chk(self.lib.pic_gather_E(self._h, _ptr(E), PIC_HOST))
        return E

    def cic(self, env=0):
        jl = np.empty(self.N, dtype=np.int64)
        jr = np.empty_like(jl)
        wl = np.empty(self.N)
        wr = np.empty_like(wl)
        self._chk(self.lib.pic_get_cic(self._h, int(env), _ptr(jl), _ptr(jr), _ptr(wl), _ptr(wr)))
        return jl, jr, wl, wr

    def eval_field(self, x, E_ext=None, fields=True):
        """compute_E on arbitrary positions -> (n, E_mesh(+E_ext), 0.5*sum(E^2)*dx) per env (n, E_mesh None with fields=False:
        a caller that wants the energy alone saves two read-backs)."""
        x = self._particles_in(x)
        e = None if E_ext is None else np.ascontiguousarray(np.asarray(E_ext, dtype=np.float64).reshape(self.num_envs, self.Ng))
        n = np.empty((self.num_envs, self.Ng)) if fields else None
        E = np.empty_like(n) if fields else None
        pe = np.empty(self.num_envs)
        self._chk(self.lib.pic_eval_field(self._h, _ptr(x), PIC_HOST, _ptr(e), _ptr(n), _ptr(E), _ptr(pe)))
        return n, E, pe

    def step_stage(self, stage, E_ext=None):
        e = None if E_ext is None else np.ascontiguousarray(np.asarray(E_ext, dtype=np.float64).reshape(self.num_envs, self.Ng))
        self._chk(self.lib.pic_step_stage(self._h, int(stage), _ptr(e), PIC_HOST))

    def step_history(self, E_ext=None, nsteps=1):
        """nsteps steps; returns (KE, PE, PE_reward), each [nsteps][num_envs], the energies after every step."""
        e = None if E_ext is None else np.ascontiguousarray(np.asarray(E_ext, dtype=np.float64).reshape(self.num_envs, self.Ng))
        hist = np.empty((int(nsteps), 3, self.num_envs))
        self._chk(self.lib.pic_step_history(self._h, _ptr(e), PIC_HOST, int(nsteps), _ptr(hist)))
        return hist[:, 0], hist[:, 1], hist[:, 2]

    def step_snapshots(self, E_ext=None, nsteps=1):
        """nsteps steps; returns (x, v, KE, PE, PE_reward): particles after every step [nsteps][num_envs][N] and the
        energies [nsteps][num_envs] -- what PIC.simulate records, with one read-back at the end."""
        e = None if E_ext is None else np.ascontiguousarray(np.asarray(E_ext, dtype=np.float64).reshape(self.num_envs, self.Ng))
        snap = np.empty((int(nsteps), 2, self.num_envs, self.N), dtype=self.dtype)
        hist = np.empty((int(nsteps), 3, self.num_envs))
        self._chk(self.lib.pic_step_snapshots(self._h, _ptr(e), PIC_HOST, int(nsteps), _ptr(snap), _ptr(hist)))
        return snap[:, 0], snap[:, 1], hist[:, 0], hist[:, 1], hist[:, 2]

    def compute_E(self, x, E_ext=None, particles=True, shape=False):
        """compute_E(return_all=True) + shape bookkeeping on arbitrary positions.  Returns a dict with
        n, E_mesh (incl. E_ext), phi_mesh [env][Ng]; E, phi [env][N] if `particles`; idx (int64), w [env][3][N]
        if `shape` (rows l, r, - for CIC; l, m, r for TSC)."""
        x = self._particles_in(x)
        E_, N, Ng = self.num_envs, self.N, self.Ng
        e = None if E_ext is None else np.ascontiguousarray(np.asarray(E_ext, dtype=np.float64).reshape(E_, Ng))
        out = {"n": np.empty((E_, Ng)), "E_mesh": np.empty((E_, Ng)), "phi_mesh": np.empty((E_, Ng))}
        if particles:
            out["E"] = np.empty((E_, N), dtype=self.dtype)
            out["phi"] = np.empty((E_, N), dtype=self.dtype)
        if shape:
            out["idx"] = np.empty((E_, 3, N), dtype=np.int64)
            out["w"] = np.empty((E_, 3, N))
        self._chk(self.lib.pic_compute_E(self._h, _ptr(x), PIC_HOST, _ptr(e), _ptr(out.get("E")), _ptr(out.get("phi")),
                                         _ptr(out["n"]), _ptr(out["E_mesh"]), _ptr(out["phi_mesh"]),
                                         _ptr(out.get("idx")), _ptr(out.get("w"))))
        return out

    def solve_poisson(self, rhs):
        """Periodic 3-point Poisson solve of rhs [env][Ng] (zero sum) -> (phi with zero mean, E_mesh = -grad phi)."""
        b = np.ascontiguousarray(np.asarray(rhs, dtype=np.float64).reshape(self.num_envs, self.Ng))
        phi = np.empty_like(b)
        E = np.empty_like(b)
        self._chk(self.lib.pic_solve_poisson(self._h, _ptr(b), _ptr(phi), _ptr(E)))
        return phi, E

    def device_ptrs(self):
        ps = [C.c_void_p() for _ in range(8)]
        ld = C.c_int64()
        self._chk(self.lib.pic_device_ptrs(self._h, C.byref(ps[0]), C.byref(ps[1]), C.byref(ld), C.byref(ps[2]),
                                           C.byref(ps[3]), C.byref(ps[4]), C.byref(ps[5]), C.byref(ps[6]),
                                           C.byref(ps[7])))
        names = ("x", "v", "n", "E_mesh", "phi", "KE", "PE", "PE_reward")
        out = {k: p.value for k, p in zip(names, ps)}
        out["ld"] = ld.value
        return out

    def profile(self, enable=True):
        self._chk(self.lib.pic_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        ms = (C.c_double * 8)()
        cnt = (C.c_int64 * 8)()
        self._chk(self.lib.pic_profile_read(self._h, ms, cnt))
        return {KIND_NAMES[i]: (ms[i], cnt[i]) for i in range(7) if cnt[i]}

    def set_actuator(self, basis_cos, basis_sin):
        bc = np.ascontiguousarray(basis_cos, dtype=np.float64)
        bs = np.ascontiguousarray(basis_sin, dtype=np.float64)
        if bc.shape != bs.shape or bc.shape[0] != self.Ng:
            raise ValueError("basis tables must be [Ng, max_mode]")
        self.max_mode = int(bc.shape[1])
        self._chk(self.lib.pic_set_actuator(self._h, self.max_mode, _ptr(bc), _ptr(bs)))

    def step_actions(self, actions, nsteps=1):
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(self.num_envs, 2 * self.max_mode))
        self._chk(self.lib.pic_step_actions(self._h, _ptr(a), PIC_HOST, int(nsteps)))

    def step_actions_device(self, actions_ptr, nsteps=1):
        self._chk(self.lib.pic_step_actions(self._h, _ptr(int(actions_ptr)), PIC_DEVICE, int(nsteps)))

    def _hist_out(self, nsteps, history):
        return np.empty((int(nsteps), 3, self.num_envs)) if history else None

    def step_actions_traj(self, actions, history=False):
        """One step per row of actions [nsteps][num_envs][2*max_mode], all in one call (pic_step_actions_traj).
        history=True: returns (KE, PE, PE_reward), each [nsteps][num_envs]; otherwise asynchronous, returns None."""
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64))
        a = a.reshape(-1, self.num_envs, 2 * self.max_mode)
        hist = self._hist_out(a.shape[0], history)
        self._chk(self.lib.pic_step_actions_traj(self._h, _ptr(a), PIC_HOST, a.shape[0], _ptr(hist)))
        return None if hist is None else (hist[:, 0], hist[:, 1], hist[:, 2])

    def step_actions_traj_device(self, actions_ptr, nsteps):
        """actions: device pointer to [nsteps][num_envs][2*max_mode] float64; asynchronous on the handle's stream."""
        self._chk(self.lib.pic_step_actions_traj(self._h, _ptr(int(actions_ptr)), PIC_DEVICE, int(nsteps), None))

    def step_ext_traj(self, E_ext_traj, history=False, snapshots=False):
        """One step per row of E_ext_traj [nsteps][num_envs][Ng] (pic_step_ext_traj): PIC.simulate(E_external_traj).
        Returns None, (KE, PE, PE_reward), or (x, v, KE, PE, PE_reward) with snapshots=True."""
        e = np.ascontiguousarray(np.asarray(E_ext_traj, dtype=np.float64)).reshape(-1, self.num_envs, self.Ng)
        k = e.shape[0]
        hist = self._hist_out(k, history or snapshots)
        snap = np.empty((k, 2, self.num_envs, self.N), dtype=self.dtype) if snapshots else None
        self._chk(self.lib.pic_step_ext_traj(self._h, _ptr(e), PIC_HOST, k, _ptr(hist), _ptr(snap)))
        if snapshots:
            return snap[:, 0], snap[:, 1], hist[:, 0], hist[:, 1], hist[:, 2]
        return None if hist is None else (hist[:, 0], hist[:, 1], hist[:, 2])

    def step_feedback(self, nsteps, actions=False, history=False):
        """nsteps of the linear feedback loop on the device (pic_step_feedback; the actuator's max_mode).  Returns a dict with
        "actions" [nsteps][num_envs][2*max_mode] and / or "KE", "PE", "PE_reward" [nsteps][num_envs] as asked for; with
        neither the call is asynchronous and returns None."""
        k = int(nsteps)
        act = np.empty((k, self.num_envs, 2 * self.max_mode)) if actions else None
        hist = self._hist_out(k, history)
        self._chk(self.lib.pic_step_feedback(self._h, self.max_mode, k, _ptr(act), _ptr(hist)))
        out = {}
        if act is not None:
            out["actions"] = act
        if hist is not None:
            out.update(KE=hist[:, 0], PE=hist[:, 1], PE_reward=hist[:, 2])
        return out or None

    def step_observe(self, E_ext=None, actions=None, nsteps=1, particles=True, out=None):
        """One Gym-style iteration in one call with one synchronisation (pic_step_observe): nsteps steps under E_ext
        [num_envs][Ng] or actions [num_envs][2*max_mode] (or neither), then -> (x, v, KE, PE, PE_reward) of the new state
        (x, v None with particles=False).  out: (x, v) C-contiguous arrays of num_envs * N elements of the particle dtype each
        to receive the particles (e.g. the two halves of one observation buffer) instead of fresh ones."""
        if E_ext is not None:
            E_ext = np.ascontiguousarray(np.asarray(E_ext, dtype=np.float64).reshape(self.num_envs, self.Ng))
        if actions is not None:
            actions = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(self.num_envs, 2 * self.max_mode))
        if out is not None:
            x, v = out
            for a in (x, v):
                if a.dtype != self.dtype or a.size != self.num_envs * self.N or not a.flags.c_contiguous:
                    raise ValueError("out: two C-contiguous arrays of num_envs * N elements of the particle dtype")
        else:
            x = np.empty((self.num_envs, self.N), dtype=self.dtype) if particles else None
            v = np.empty_like(x) if particles else None
        en = np.empty((3, self.num_envs))
        self._chk(self.lib.pic_step_observe(self._h, _ptr(E_ext), _ptr(actions), int(nsteps), _ptr(x), _ptr(v),
                                            _ptr(en[0]), _ptr(en[1]), _ptr(en[2])))
        return x, v, en[0], en[1], en[2]

    def set_stream(self, hip_stream):
        """hip_stream: integer hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; 0 = the default stream)."""
        self._chk(self.lib.pic_set_stream(self._h, C.c_void_p(int(hip_stream)) if hip_stream else None))

    def own_stream(self):
        self._chk(self.lib.pic_own_stream(self._h))

    def modes_device(self, max_mode, re_ptr, im_ptr):
        self._chk(self.lib.pic_get_modes(self._h, int(max_mode), _ptr(int(re_ptr)), _ptr(int(im_ptr)), PIC_DEVICE))

    def modes(self, max_mode):
        re = np.empty((self.num_envs, int(max_mode)))
        im = np.empty_like(re)
        self._chk(self.lib.pic_get_modes(self._h, int(max_mode), _ptr(re), _ptr(im), PIC_HOST))
        return re + 1j * im

    def phase_histogram(self, nbins, vmin, vmax):
        counts = np.zeros((self.num_envs, int(nbins), int(nbins)), dtype=np.uint32)
        self._chk(self.lib.pic_phase_histogram(self._h, int(nbins), float(vmin), float(vmax), _ptr(counts)))
        return counts

    def phase_kl(self, feq, vmin, vmax):
        """KL cost of every environment's phase-space density against feq [nbins][nbins] (pic_phase_kl) -> [num_envs]."""
        f = np.ascontiguousarray(np.asarray(feq, dtype=np.float64))
        if f.ndim != 2 or f.shape[0] != f.shape[1]:
            raise ValueError("feq must be [nbins, nbins]")
        kl = np.empty(self.num_envs)
        self._chk(self.lib.pic_phase_kl(self._h, int(f.shape[0]), float(vmin), float(vmax), _ptr(f), _ptr(kl)))
        return kl

    def stream_probe(self, repeats=10):
        g = C.c_double()
        self._chk(self.lib.pic_stream_probe(self._h, int(repeats), C.byref(g)))
        return g.value

    def bad_count(self):
        c = C.c_int64()
        self._chk(self.lib.pic_bad_count(self._h, C.byref(c)))
        return c.value
