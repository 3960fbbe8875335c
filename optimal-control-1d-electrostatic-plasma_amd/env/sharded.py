"""Environment sharding across the GPUs of one node: one process per GPU, each owning a contiguous
block of the rollout environments in its own `BatchedPIC` handle.

Nothing couples environments inside a step, so there is NO collective on the step path.  The only
exchanges are (i) the all-gather of per-environment returns / rewards (<= 1 KB per rank, latency
bound on xGMI) and (ii) optionally the broadcast of actions from the rank that runs the policy.
`torch.distributed` backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
from typing import Callable, Optional

import numpy as np


def shard_range(rank: int, world: int, total_envs: int):
    """Contiguous block of environments owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(total_envs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def check_one_device_per_rank(placements):
    """placements: one (host name, device ordinal) per rank.  Two ranks on one device would step their shards one after the
    other on the same GPU while the job reports itself as N-way parallel: refuse."""
    seen = {}
    for rank, place in enumerate(placements):
        if place in seen:
            raise RuntimeError(f"ranks {seen[place]} and {rank} both drive device {place[1]} of host {place[0]}: "
                               "one rank per GPU (set the device from LOCAL_RANK)")
        seen[place] = rank


class ShardedPIC:
    def __init__(self, total_envs: int, N: int, N_mesh: int, env_factory: Optional[Callable] = None,
                 device: Optional[int] = None, **env_kwargs):
        import torch.distributed as dist

        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.total_envs = int(total_envs)
        self.lo, self.hi = shard_range(self.rank, self.world, self.total_envs)
        self.num_local = self.hi - self.lo
        self.counts = [shard_range(r, self.world, self.total_envs) for r in range(self.world)]
        env_factory_is_default = env_factory is None
        if env_factory is None:
            from .batched import BatchedPIC

            def env_factory(num_envs, N, N_mesh, **kw):
                return BatchedPIC(num_envs, N, N_mesh, **kw)
        # The device is resolved ONCE, here: the ordinal that is checked is the ordinal the handle is created on.  (Left to the
        # factory's default a rank that had only called torch.cuda.set_device(LOCAL_RANK) would pass the check below with its
        # current device and then build its handle on device 0 -- the very sharing the check refuses.)
        nccl = dist.is_initialized() and dist.get_backend() == "nccl"
        if device is not None:
            env_kwargs["device"] = int(device)
        elif "device" not in env_kwargs and nccl:
            import torch
            env_kwargs["device"] = int(torch.cuda.current_device())
        if nccl and self.world > 1:
            import socket
            mine = (socket.gethostname(), int(env_kwargs["device"]))
            every = [None] * self.world
            dist.all_gather_object(every, mine)
            check_one_device_per_rank(every)
        self.device = env_kwargs.get("device", 0)
        if env_factory_is_default:
            env_kwargs.setdefault("env_index_base", self.lo)      # device sampler keyed by the GLOBAL environment index
        self.env = env_factory(self.num_local, N, N_mesh, **env_kwargs)
        self.N, self.N_mesh = N, N_mesh

    def _backend_device(self):
        import torch
        if self.dist.is_initialized() and self.dist.get_backend() == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def local_slice(self, global_array):
        """Rows of a [total_envs, ...] array that belong to this rank."""
        return np.asarray(global_array)[self.lo:self.hi]

    def reset(self, x0, v0, is_global: bool = False):
        if is_global:
            x0, v0 = self.local_slice(x0), self.local_slice(v0)
        self.env.reset(x0, v0)

    def step(self, E_external=None, nsteps: int = 1, is_global: bool = False):
        if E_external is not None and is_global:
            E_external = self.local_slice(E_external)
        self.env.step(E_external, nsteps)

    # -- tensor path: the exchange stays on the collective's device (RCCL: the GPU; gloo: the host) ----------------
    def broadcast_actions_tensor(self, actions=None, src: int = 0, width: Optional[int] = None, dtype=None):
        """[total_envs, A] actions decided on rank `src` -> this rank's rows [num_local, A], as a tensor on the
        collective's device and without a NumPy hop.  Ranks other than `src` may pass None together with `width`
        (= A).  With RCCL the result is a CUDA tensor that `env.step_actions_torch` consumes in stream order."""
        import torch
        dev = self._backend_device()
        dtype = dtype or torch.float64
        if actions is not None:
            t = torch.as_tensor(actions, dtype=dtype).to(dev).contiguous()
        else:
            if width is None:
                raise ValueError("a rank that passes no actions must pass `width`")
            t = torch.empty((self.total_envs, int(width)), dtype=dtype, device=dev)
        if self.world > 1:
            self.dist.broadcast(t, src=src)
        return t[self.lo:self.hi].contiguous()

    def gather_tensor(self, local):
        """All-gather of a per-environment tensor [num_local, ...] -> [total_envs, ...] in global environment order,
        identical on every rank, on the collective's device (ragged shards are padded to the largest one)."""
        import torch
        dev = self._backend_device()
        loc = local.to(dev)
        if self.world == 1:
            return loc
        tail = tuple(loc.shape[1:])
        pad = max(hi - lo for lo, hi in self.counts)
        buf = torch.zeros((pad,) + tail, dtype=loc.dtype, device=dev)
        buf[: self.num_local] = loc
        out = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(out, buf)
        return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, self.counts)], dim=0)

    def gather_returns_tensor(self):
        """max(1 - PE_r, 0) of every environment of every rank (reward.py:72); with RCCL neither the reward nor the
        gathered result leaves the device."""
        import torch
        if hasattr(self.env, "rewards_torch") and self._backend_device().type == "cuda":
            self.env.sync()
            return self.gather_tensor(self.env.rewards_torch())
        return self.gather_tensor(torch.as_tensor(np.ascontiguousarray(self.env.rewards(), dtype=np.float64)))

    # -- NumPy path (thin wrappers over the tensor path) -----------------------------------------------------------
    def broadcast_actions(self, actions, src: int = 0):
        """[total_envs, A] actions decided on `src` -> every rank (then `local_slice`)."""
        import torch
        t = torch.as_tensor(np.ascontiguousarray(actions, dtype=np.float64), device=self._backend_device())
        if self.world > 1:
            self.dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def gather(self, local_values):
        """All-gather of a per-environment quantity: [num_local, ...] -> [total_envs, ...] in global
        environment order, identical on every rank."""
        import torch
        loc = np.ascontiguousarray(local_values, dtype=np.float64)
        if self.world == 1:
            return loc
        return self.gather_tensor(torch.as_tensor(loc)).cpu().numpy()

    def gather_returns(self):
        """Per-environment reward of the current state, max(1 - PE_r, 0) (reward.py:72), for all ranks."""
        return self.gather(self.env.rewards())

    def gather_energies(self):
        ke, pe, per = self.env.energies()
        return self.gather(np.stack([ke, pe, per], axis=1))

    def close(self):
        self.env.close()
