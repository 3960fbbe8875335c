"""world_size-2 CPU test of the env-sharding layer (gloo).  The GPU library cannot run here, so each
rank's shard is stepped by an oracle-backed stand-in with BatchedPIC's surface; what is under test
is the partition, the action broadcast and the all-gather order -- the code the 8-GPU run relies on."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N, NG, L, TOTAL = 600, 32, 50.0, 5          # 5 envs over 2 ranks: shards of 3 and 2 (ragged on purpose)


class OracleBatch:
    """BatchedPIC surface on top of the NumPy oracle (test double, CPU only)."""

    def __init__(self, num_envs, N, N_mesh, L=50.0, dt=0.1, **_):
        self.num_envs, self.N, self.N_mesh, self.L, self.dt = num_envs, N, N_mesh, L, dt
        self.sims = []

    def reset(self, x0, v0):
        from oracle import pic_oracle as po
        self.sims = [po.OraclePIC(x0[e], v0[e], self.N_mesh, L=self.L, dt=self.dt, perturb=False, faithful=False)
                     for e in range(self.num_envs)]

    def step(self, E_external=None, nsteps=1):
        for _ in range(nsteps):
            for e, s in enumerate(self.sims):
                s.update_state(None if E_external is None else np.asarray(E_external[e]).reshape(-1, 1))

    def energies(self):
        from oracle import pic_oracle as po
        ke = np.array([s.kinetic_energy() for s in self.sims])
        pe = np.array([s.get_electric_energy() for s in self.sims])
        per = np.array([po.reward_electric_energy(s.get_state(), None, s.N_mesh, s.L, s.n0) for s in self.sims])
        return ke, pe, per

    def rewards(self):
        return np.maximum(1.0 - self.energies()[2], 0.0)

    def close(self):
        pass


def _inputs():
    from oracle import pic_oracle as po
    xs, vs = zip(*[po.synthetic_two_stream(N, L, seed=40 + e) for e in range(TOTAL)])
    return np.stack(xs), np.stack(vs)


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ocplasma_amd
    from ocplasma_amd.env.sharded import ShardedPIC
    from ocplasma_amd.control.actuator import E_field

    x0, v0 = _inputs()
    env = ShardedPIC(TOTAL, N, NG, env_factory=OracleBatch, L=L, dt=0.1)
    env.reset(x0, v0, is_global=True)
    act = E_field(L, NG, 2)
    rng = np.random.default_rng(7)          # only rank 0's draw matters: it is broadcast
    actions = rng.uniform(-1, 1, (TOTAL, 4)) if rank == 0 else np.zeros((TOTAL, 4))
    drawn = actions
    actions = env.broadcast_actions(actions, src=0)
    # the tensor path: rank 0 hands over its tensor, the others only say how wide it is; everybody gets its own rows
    mine = env.broadcast_actions_tensor(torch.as_tensor(drawn) if rank == 0 else None, src=0, width=4)
    assert isinstance(mine, torch.Tensor) and tuple(mine.shape) == (env.num_local, 4)
    assert np.array_equal(mine.numpy(), actions[env.lo:env.hi])
    env.step(act.compute_E_batched(actions), nsteps=2, is_global=True)
    ke, pe, per = env.env.energies()
    gathered = env.gather_tensor(torch.as_tensor(np.stack([ke, pe, per], axis=1)))      # ragged: 3 + 2 rows
    res = {"range": (env.lo, env.hi), "returns": env.gather_returns(), "energies": env.gather_energies(),
           "actions": actions, "returns_tensor": env.gather_returns_tensor().numpy(), "energies_tensor": gathered.numpy()}
    out[rank] = res
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_partitions_every_env_once():
    from ocplasma_amd.env.sharded import shard_range
    for total in (1, 5, 64, 1024, 7):
        for world in (1, 2, 3, 8):
            spans = [shard_range(r, world, total) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sharded_rollout_matches_single_process():
    from ocplasma_amd.control.actuator import E_field
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0]["range"] == (0, 3) and out[1]["range"] == (3, 5)
    # every rank holds the same gathered arrays, in global env order
    for k in ("returns", "energies", "actions", "returns_tensor", "energies_tensor"):
        assert np.array_equal(out[0][k], out[1][k]), k
    assert np.array_equal(out[0]["returns_tensor"], out[0]["returns"])            # tensor path == NumPy path
    assert np.array_equal(out[0]["energies_tensor"], out[0]["energies"]) and out[0]["energies_tensor"].shape == (TOTAL, 3)
    # and they equal an unsharded run
    x0, v0 = _inputs()
    ref = OracleBatch(TOTAL, N, NG, L=L, dt=0.1)
    ref.reset(x0, v0)
    ref.step(E_field(L, NG, 2).compute_E_batched(out[0]["actions"]), nsteps=2)
    ke, pe, per = ref.energies()
    assert np.allclose(out[0]["energies"], np.stack([ke, pe, per], 1), rtol=1e-13)
    assert np.allclose(out[0]["returns"], ref.rewards(), rtol=1e-13)
    assert out[0]["actions"].any()           # rank 1 really received rank 0's actions


def test_two_ranks_on_one_device_are_refused():
    """Under RCCL every rank must drive a GPU of its own (ShardedPIC checks it at construction); the rule itself is plain data."""
    from ocplasma_amd.env.sharded import check_one_device_per_rank
    check_one_device_per_rank([("node", 0), ("node", 1), ("other", 0)])
    with pytest.raises(RuntimeError, match="ranks 0 and 2 both drive device 0"):
        check_one_device_per_rank([("node", 0), ("node", 1), ("node", 0)])
