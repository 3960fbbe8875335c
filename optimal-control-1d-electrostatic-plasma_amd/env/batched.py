"""Batched rollout environments: `num_envs` independent PIC systems stepped by one
libpicstep handle on one MI355X (BASELINE configs 2-5).  Nothing couples the
environments inside a step; across GPUs they are sharded rank-wise
(`.sharded.ShardedPIC`), never split.
"""
from typing import Optional

import numpy as np

from .. import _abi


class _DeviceView:
    """Minimal __cuda_array_interface__ carrier so torch can alias library-owned memory."""

    def __init__(self, ptr, shape, typestr, strides=None):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": strides}


class BatchedPIC:
    def __init__(self, num_envs: int, N: int, N_mesh: int, n0: float = 1.0, L: float = 50.0, dt: float = 0.1,
                 gamma: float = 5.0, interpol: str = "CIC", device: int = 0, dtype="float64", accum_dtype=None,
                 blocks_per_env: int = 0, verbose: bool = False, env_index_base: int = 0, position_dtype=None,
                 placement: str = "auto", placement_ms: int = 0):
        self.num_envs, self.N, self.N_mesh = int(num_envs), int(N), int(N_mesh)
        self.n0, self.L, self.gamma, self.interpol = n0, L, gamma, interpol
        self.dx = L / N_mesh
        # CFL clamp of PIC.initialize (src/env/pic.py:71-73)
        self.dt = dt
        if self.dt > 2 / np.sqrt(self.N / self.L):
            self.dt = 2 / np.sqrt(self.N / self.L)
            if verbose:
                print("CFL condtion invalid: change dt = {:.4f}".format(self.dt))
        self.device = device
        self.dtype = np.dtype(dtype)
        self._h = _abi.Handle(self.N, self.N_mesh, self.num_envs, L, n0, self.dt, gamma, self.dtype, accum_dtype,
                              interpol, device, blocks_per_env, env_index_base, position_dtype, placement, placement_ms)
        self.fixed_positions = self._h.fixed_positions

    # reset(x0, v0): x0, v0 are [num_envs, N] with any velocity perturbation already applied
    def reset(self, x0, v0):
        self._h.reset(x0, v0)

    def reset_sampled(self, kind: str = "bump-on-tail", a: float = 0.2, v0: float = 3.0, sigma: float = 1.0,
                      A: float = 0.1, n_mode: int = 2, seed: int = 0):
        """`reinit()` for every environment with the sample drawn on the device: the distributions of the
        reference's TwoStream / BumpOnTail (same particle ordering), velocity perturbation included.  A new
        `seed` gives a new ensemble; environment e always differs from environment e'.  Not NumPy's RNG
        stream -- use `reset(x0, v0)` with the host samplers when the reference's exact particles are wanted."""
        self._h.reset_sampled(kind, a, v0, sigma, A, n_mode, seed)

    def reset_device(self, x_ptr, v_ptr):
        self._h.reset_device(x_ptr, v_ptr)

    def step(self, E_external: Optional[np.ndarray] = None, nsteps: int = 1):
        """nsteps x update_state for every environment; asynchronous (call sync() or a getter)."""
        self._h.step(E_external, nsteps)

    def simulate(self, nsteps: int, E_external: Optional[np.ndarray] = None):
        """The energy traces of PIC.simulate (pic.py:175-223) for every environment, without its particle
        snapshots: -> (E, PE), each [nsteps + 1, num_envs], entry 0 = the state before the first step, as the
        reference records it.  The steps run back to back on the device; one read-back at the end."""
        ke0, pe0, _ = self.energies()
        ke, pe, _ = self._h.step_history(E_external, nsteps)
        return np.concatenate([(ke0 + pe0)[None], ke + pe]), np.concatenate([pe0[None], pe])

    def simulate_snapshots(self, nsteps: int, E_external: Optional[np.ndarray] = None):
        """PIC.simulate's snapshots for every environment: (x, v) after each step, each [nsteps, num_envs, N] of the
        particle dtype, and (KE, PE, PE_reward) [nsteps, num_envs]; one read-back for all steps (mind the size:
        nsteps x 2 x num_envs x N values are held on the device until the end of the call)."""
        return self._h.step_snapshots(E_external, nsteps)

    def step_history(self, E_external: Optional[np.ndarray] = None, nsteps: int = 1):
        """nsteps x update_state; -> (KE, PE, PE_reward) after every step, each [nsteps, num_envs]."""
        return self._h.step_history(E_external, nsteps)

    def step_device(self, E_ext_ptr=0, nsteps: int = 1):
        self._h.step_device(E_ext_ptr, nsteps)

    def sync(self):
        self._h.sync()

    def get_state(self):
        """[num_envs, 2N] float64: per environment the reference's get_state() column, flattened."""
        x, v = self._h.particles()
        return np.concatenate([x, v], axis=1, dtype=np.float64)

    def particles(self):
        return self._h.particles()

    def fields(self):
        return self._h.fields()

    def energies(self):
        """(KE, PE, PE_reward), each [num_envs]."""
        return self._h.energies()

    def rewards(self):
        """max(1 - PE_reward, 0) per environment (reward.py:72 with r_pe_n = 1)."""
        return np.maximum(1.0 - self._h.energies()[2], 0.0)

    def gather_E(self):
        return self._h.gather_E()

    def eval_field(self, x, E_ext=None):
        return self._h.eval_field(x, E_ext)

    # -- control loop on the device ---------------------------------------
    def set_actuator(self, actuator):
        """Upload an `E_field`'s basis tables (its linspace mesh included) for `step_actions`."""
        self._h.set_actuator(actuator.basis_cos, actuator.basis_sin)
        self.max_mode = actuator.max_mode

    def step_actions(self, actions, nsteps: int = 1):
        """actions [num_envs, 2*max_mode] (cos coefficients, then sin): E_ext is built on the device
        (actuator.py:54-63) and held for `nsteps` steps."""
        self._h.step_actions(actions, nsteps)

    def step_observe(self, E_external: Optional[np.ndarray] = None, actions: Optional[np.ndarray] = None, nsteps: int = 1):
        """One iteration of a Gym-style loop for every environment in ONE call with ONE synchronisation (pic_step_observe;
        ddpg.py:421-468: update_state -> get_state -> reward): nsteps steps under E_external [num_envs, Ng] or actions
        [num_envs, 2*max_mode] (or neither) -> (state [num_envs, 2N] float64, (KE, PE, PE_reward) each [num_envs])."""
        x, v, ke, pe, per = self._h.step_observe(E_external, actions, nsteps)
        return np.concatenate([x, v], axis=1, dtype=np.float64), (ke, pe, per)

    def step_actions_device(self, actions_ptr, nsteps: int = 1):
        self._h.step_actions_device(actions_ptr, nsteps)

    def step_actions_traj(self, actions, history: bool = False):
        """A rollout with a new action every step in ONE call (pic_step_actions_traj): actions
        [nsteps, num_envs, 2*max_mode]; step s runs under actions[s] -- the inner loop of the trainers
        (src/control/rl/ddpg.py:421-468) once the actions are known.  history=True -> (KE, PE, PE_reward), each
        [nsteps, num_envs], after every step; otherwise asynchronous."""
        return self._h.step_actions_traj(actions, history)

    def step_actions_traj_torch(self, actions):
        """The same from a float64 CUDA tensor [nsteps, num_envs, 2*max_mode], consumed in stream order."""
        if not (actions.is_cuda and actions.dtype.is_floating_point and actions.element_size() == 8 and actions.is_contiguous()
                and actions.dim() == 3 and tuple(actions.shape[1:]) == (self.num_envs, 2 * self.max_mode)):
            raise ValueError("actions must be a contiguous float64 CUDA tensor [nsteps, num_envs, 2*max_mode]")
        shared = getattr(self, "_torch_stream", None) is not None
        if not shared:
            import torch
            torch.cuda.current_stream(self.device).synchronize()
        self._h.step_actions_traj_device(actions.data_ptr(), int(actions.shape[0]))
        if not shared:
            self._h.sync()

    def step_ext_traj(self, E_ext_traj, history: bool = False, snapshots: bool = False):
        """PIC.simulate's E_external_traj (pic.py:175-223) for every environment in one call: E_ext_traj
        [nsteps, num_envs, N_mesh]; step s runs under E_ext_traj[s]."""
        return self._h.step_ext_traj(E_ext_traj, history, snapshots)

    def step_feedback(self, nsteps: int, actions: bool = False, history: bool = False):
        """nsteps iterations of run_feedback.py:130-168's loop body on the device (pic_step_feedback): before each step the
        actuator coefficients become (-Re E_k, +Im E_k), k = 1..max_mode, of the current E_mesh.  Bit for bit the host loop
        `step_actions(feedback_actions(max_mode))`.  Returns a dict with "actions" [nsteps, num_envs, 2*max_mode] and / or
        "KE", "PE", "PE_reward" [nsteps, num_envs] as asked for (None, and asynchronous, with neither)."""
        return self._h.step_feedback(nsteps, actions, history)

    def modes(self, max_mode: int):
        """Complex [num_envs, max_mode]: rows 1..max_mode of compute_E_k_spectrum for the current E_mesh."""
        return self._h.modes(max_mode)

    def feedback_actions(self, max_mode: int):
        """The linear-feedback / behaviour-cloning action of run_feedback.py:133-135 and ddpg.py:369-371:
        cos coefficients -Re(E_k), sin coefficients +Im(E_k), k = 1..max_mode."""
        ek = self.modes(max_mode)
        return np.concatenate([-ek.real, ek.imag], axis=1)

    # -- stream-ordered operation next to torch (no host synchronisation in the loop) ------------------
    def use_torch_stream(self, stream=None):
        """Run this environment's kernels on torch's current (or the given) stream, so that torch ops
        on that stream and environment steps are ordered by the stream alone."""
        import torch
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        self._h.set_stream(st.cuda_stream)
        self._torch_stream = st

    def use_own_stream(self):
        self._h.own_stream()
        self._torch_stream = None

    def step_actions_torch(self, actions, nsteps: int = 1):
        """actions: float64 CUDA tensor [num_envs, 2*max_mode] on this device; consumed in stream order."""
        if not (actions.is_cuda and actions.dtype.is_floating_point and actions.element_size() == 8
                and actions.is_contiguous() and tuple(actions.shape) == (self.num_envs, 2 * self.max_mode)):
            raise ValueError("actions must be a contiguous float64 CUDA tensor [num_envs, 2*max_mode]")
        shared = getattr(self, "_torch_stream", None) is not None
        if not shared:      # different streams: order them through the host
            import torch
            torch.cuda.current_stream(self.device).synchronize()
        self._h.step_actions_device(actions.data_ptr(), nsteps)
        if not shared:
            self._h.sync()

    def feedback_actions_torch(self, max_mode: int):
        """`feedback_actions` with everything on the device: returns a float64 CUDA tensor [num_envs, 2*max_mode]."""
        import torch
        dev = f"cuda:{self.device}"
        re = torch.empty((self.num_envs, max_mode), dtype=torch.float64, device=dev)
        im = torch.empty_like(re)
        shared = getattr(self, "_torch_stream", None) is not None
        if not shared:
            torch.cuda.current_stream(self.device).synchronize()
        self._h.modes_device(max_mode, re.data_ptr(), im.data_ptr())
        if not shared:
            self._h.sync()
        return torch.cat([-re, im], dim=1)

    def _ordered_views(self):
        """The zero-copy views, safe to read on torch's current stream: on a shared stream (use_torch_stream) the stream
        orders the read behind the steps; otherwise the handle's own stream is drained first."""
        if not hasattr(self, "_views"):
            self._views = self.torch_views()
        if getattr(self, "_torch_stream", None) is None:
            self._h.sync()
        return self._views

    def energy_views_torch(self):
        """{"KE", "PE", "PE_reward"}: the zero-copy [num_envs] views of the energies the last step left, safe to read on torch's
        current stream (no kernel is launched: the handle's own stream is drained unless it is shared with torch)."""
        v = self._ordered_views()
        return {k: v[k] for k in ("KE", "PE", "PE_reward")}

    def rewards_torch(self):
        """max(1 - PE_reward, 0) per environment as a CUDA tensor (reward.py:72), read from the zero-copy view (after a
        sync of the handle's stream unless it is shared with torch: use_torch_stream)."""
        import torch
        return torch.clamp(1.0 - self._ordered_views()["PE_reward"], min=0.0)

    def trainer_rewards_torch(self, actions=None, alpha: float = 1.0, beta: float = 1.0, r_pe_n: float = 1.0,
                              r_ie_n: Optional[float] = None):
        """Reward.compute_reward (src/control/rl/reward.py:71-76) for every environment, on the device:
        alpha max(1 - PE_r / r_pe_n, 0) + beta max(1 - (sum a^2 L / 4) / r_ie_n, 0), with PE_r the field energy of the
        CURRENT state (read before stepping, it is the pre-step reward the trainers use, ddpg.py:455) and `actions` the
        [num_envs, A] CUDA tensor about to be applied.  r_ie_n defaults to the reference's normaliser, the input energy of
        an all-ones action of the same length (reward.py:26)."""
        import torch
        r = alpha * torch.clamp(1.0 - self._ordered_views()["PE_reward"] / r_pe_n, min=0.0)
        if actions is not None and beta != 0.0:
            a = actions.to(torch.float64)
            ie = (a * a).sum(dim=1) * (self.L * 0.25)
            if r_ie_n is None:
                r_ie_n = a.shape[1] * self.L * 0.25
            r = r + beta * torch.clamp(1.0 - ie / r_ie_n, min=0.0)
        return r

    def phase_density(self, nbins: int, vmin: float = -25.0, vmax: float = 25.0):
        """estimate_f (src/control/objective.py:8-14) for every environment, [num_envs, nbins, nbins]:
        the histogram is counted on the device, the normalisation n0/dx/dv/N applied here."""
        counts = self._h.phase_histogram(nbins, vmin, vmax).astype(np.float64)
        dx, dv = self.L / nbins, (vmax - vmin) / nbins
        counts *= self.n0 / dx / dv / self.N
        return counts

    def kl_divergence(self, feq, vmin: float = -25.0, vmax: float = 25.0):
        """Reward.compute_kl_divergence (reward.py:43-46) of every environment against `feq` [nbins, nbins]: histogram and
        reduction on the device, one value per environment read back (pic_phase_kl)."""
        return self._h.phase_kl(feq, vmin, vmax)

    def stream_probe(self, repeats=10):
        """GB/s of a read-2-arrays / write-2-arrays copy with the sweeps' grid on this device."""
        return self._h.stream_probe(repeats)

    def bad_count(self):
        return self._h.bad_count()

    def profile(self, enable=True):
        self._h.profile(enable)

    def profile_read(self):
        return self._h.profile_read()

    def torch_views(self):
        """Zero-copy torch tensors over the device state: x, v [num_envs, N] (strided), n, E_mesh,
        phi [num_envs, Ng], KE, PE, PE_reward [num_envs].  Call sync() before reading them on
        another stream.  A write to x or v must be followed by invalidate() or refresh().
        With position_dtype="fixed32" the zero-copy position view is `x_fixed` (int32 bit pattern of the
        uint32 u, x = u L / 2^32) and `x` is a float64 COPY computed from it."""
        import torch

        p = self._h.device_ptrs()
        ts = "<f8" if self.dtype == np.float64 else "<f4"
        isz = self.dtype.itemsize
        dev = f"cuda:{self.device}"
        out = {}
        for k in ("x", "v"):
            t = "<i4" if (k == "x" and self.fixed_positions) else ts
            view = _DeviceView(p[k], (self.num_envs, self.N), t, (p["ld"] * isz, isz))
            out[k] = torch.as_tensor(view, device=dev)
        if self.fixed_positions:
            out["x_fixed"] = out["x"]
            out["x"] = (out["x_fixed"].to(torch.int64) & 0xFFFFFFFF).to(torch.float64) * (self.L / 2.0 ** 32)
        for k in ("n", "E_mesh", "phi"):
            out[k] = torch.as_tensor(_DeviceView(p[k], (self.num_envs, self.N_mesh), "<f8"), device=dev)
        for k in ("KE", "PE", "PE_reward"):
            out[k] = torch.as_tensor(_DeviceView(p[k], (self.num_envs,), "<f8"), device=dev)
        return out

    def invalidate(self):
        """After a write to x / v through `torch_views()`: drop the cached first deposit of the next step."""
        self._h.invalidate()

    def refresh(self):
        """update_density + update_E_field on the current particles (also valid after a write through the views)."""
        self._h.refresh()

    def close(self):
        self._h.close()
