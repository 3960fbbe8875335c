"""Sharded rollouts: one process per GPU, each stepping its block of independent environments; the only
collective is the all-gather of per-environment returns at the end of the episode.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        examples/sharded_rollout.py --envs 512 --particles 1000000 --mesh 256 --steps 100

Every rank draws its environments on its own device (`reset_sampled`; ShardedPIC hands the handle the global index
of its first environment, so the ensemble does not depend on the number of ranks), runs the linear feedback controller of
run_feedback.py:130-168 on the device, accumulates max(1 - PE_r, 0) per step (reward.py:72) and rank 0 prints
the episode returns of ALL environments.  `--backend gloo --same-device` rehearses the multi-rank path on a
one-GPU box (collectives on host copies).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16, help="environments in total, over all ranks")
    ap.add_argument("--particles", type=int, default=100_000)
    ap.add_argument("--mesh", type=int, default=128)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--modes", type=int, default=3)
    ap.add_argument("--gain", type=float, default=1.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-device", action="store_true", help="every rank on cuda:0 (one-GPU rehearsal)")
    ap.add_argument("--out", default=None, help="rank 0 saves the gathered returns and energies here (.npz, full precision)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ocplasma_amd
    from ocplasma_amd import E_field, ShardedPIC

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device}"))
        else:
            dist.init_process_group("gloo")

    L = 50.0
    sh = ShardedPIC(args.envs, args.particles, args.mesh, device=device, L=L, dt=0.1)
    env = sh.env                                           # this rank's BatchedPIC: environments [sh.lo, sh.hi)
    env.set_actuator(E_field(L, args.mesh, args.modes))
    env.reset_sampled("two-stream", v0=3.0, sigma=1.0, A=0.1, n_mode=2, seed=1000)   # keyed by the global env index
    # The control loop stays on the device and on one stream: modes -> action (torch) -> actuator + step -> reward view.
    env.use_torch_stream()
    returns = torch.zeros(sh.num_local, dtype=torch.float64, device=f"cuda:{device}")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        returns += env.rewards_torch()                     # reward of the pre-step state (ddpg.py:455)
        env.step_actions_torch(args.gain * env.feedback_actions_torch(args.modes))
    torch.cuda.current_stream().synchronize()
    elapsed = time.perf_counter() - t0
    # the one collective: [envs] on every rank, on the collective's device (RCCL: no host hop)
    all_returns = sh.gather_tensor(returns).cpu().numpy()
    env.use_own_stream()
    all_energy = sh.gather_energies()
    if rank == 0:
        rate = args.envs * args.particles * args.steps / elapsed
        print(f"{world} rank(s) x {sh.num_local} envs: {args.steps} controlled steps in {elapsed:.3f} s "
              f"({rate:.3e} particle-steps/s incl. the per-step feedback law, no host synchronisation in the loop)")
        print("episode returns:", np.array2string(all_returns, precision=3, max_line_width=120))
        print(f"final field energy per env: min {all_energy[:, 2].min():.3e} max {all_energy[:, 2].max():.3e}")
        if args.out:
            np.savez(args.out, returns=all_returns, energies=all_energy)
    sh.close()
    if world > 1:
        dist.destroy_process_group()
    return all_returns


if __name__ == "__main__":
    main()
