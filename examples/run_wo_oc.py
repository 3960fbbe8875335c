"""Uncontrolled Vlasov-Poisson run, the shape of the reference's baseline driver (run_wo_oc.py:20-158)
on the MI355X environment: same command-line knobs and defaults, same per-step loop
(`update_state(None)`, `get_energy`, `get_electric_energy`, snapshots, KL and field-energy cost).
Plots are left out; with --is_save the trajectory goes to an .npz next to the reference's key names.

    python examples/run_wo_oc.py --simcase bump-on-tail --num_particle 10000 --num_mesh 128 --t_max 10
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocplasma_amd
from ocplasma_amd.control.reward import Reward
from ocplasma_amd.env.dist import BumpOnTail, TwoStream
from ocplasma_amd.env.pic import PIC


def parsing(argv=None):
    p = argparse.ArgumentParser(description="Vlasov-Poisson plasma kinetic simulation without E-field control")
    p.add_argument("--simcase", type=str, default="two-stream", choices=["two-stream", "bump-on-tail"])
    p.add_argument("--interpol", type=str, default="CIC", choices=["CIC", "TSC"])
    p.add_argument("--gamma", type=float, default=5.0)
    p.add_argument("--save_file", type=str, default="./dataset/")
    p.add_argument("--is_save", action="store_true")
    p.add_argument("--num_particle", type=int, default=5000)
    p.add_argument("--num_mesh", type=int, default=250)
    p.add_argument("--t_min", type=float, default=0)
    p.add_argument("--t_max", type=float, default=50)
    p.add_argument("--dt", type=float, default=0.1)
    p.add_argument("--L", type=float, default=50)
    p.add_argument("--n0", type=float, default=1.0)
    p.add_argument("--vb", type=float, default=3.0)
    p.add_argument("--vth", type=float, default=1.0)
    p.add_argument("--A", type=float, default=0.1)
    p.add_argument("--n_mode", type=int, default=2)
    p.add_argument("--a", type=float, default=0.2)
    p.add_argument("--device", type=int, default=0)
    return vars(p.parse_args(argv))


def main(argv=None, quiet=False):
    args = parsing(argv)
    if args["simcase"] == "two-stream":
        dist = TwoStream(v0=args["vb"], sigma=args["vth"], n_samples=args["num_particle"], L=args["L"])
    else:
        dist = BumpOnTail(a=args["a"], v0=args["vb"], sigma=args["vth"], n_samples=args["num_particle"], L=args["L"])
    sim = PIC(N=args["num_particle"], N_mesh=args["num_mesh"], n0=args["n0"], L=args["L"], dt=args["dt"],
              tmin=args["t_min"], tmax=args["t_max"], gamma=args["gamma"], A=args["A"], n_mode=args["n_mode"],
              interpol=args["interpol"], init_dist=dist, device=args["device"])
    Nt = int(np.ceil((args["t_max"] - args["t_min"]) / args["dt"]))
    reward = Reward(sim.init_dist.get_init_state(), args["num_mesh"], args["L"], -25.0, 25.0, args["n0"], 1.0)

    pos, vel, E_list, PE_list, cost_kl, cost_ee = [], [], [], [], [], []
    t0 = time.perf_counter()
    for _ in range(Nt):
        sim.update_state(None)
        E_list.append(sim.get_energy())
        PE_list.append(sim.get_electric_energy())
        pos.append(sim.x.copy())
        vel.append(sim.v.copy())
        state = sim.get_state()
        cost_kl.append(reward.compute_kl_divergence(state))
        cost_ee.append(reward.compute_electric_energy(state))
    wall = time.perf_counter() - t0
    snapshot = np.concatenate([np.concatenate(pos, axis=1), np.concatenate(vel, axis=1)], axis=0)
    out = {"snapshot": snapshot, "E": np.array(E_list), "PE": np.array(PE_list), "N": args["num_particle"],
           "N_mesh": args["num_mesh"], "n0": args["n0"], "L": args["L"], "dt": sim.dt, "tmin": args["t_min"],
           "tmax": args["t_max"], "n_mode": args["n_mode"], "A": args["A"], "vth": args["vth"], "vb": args["vb"],
           "a": args["a"], "J_KL": np.array(cost_kl), "J_ee": np.array(cost_ee)}
    if not quiet:
        print(f"{args['simcase']}: {Nt} steps of N={args['num_particle']}, Ng={args['num_mesh']} in {wall:.2f} s "
              f"({wall / Nt * 1e3:.2f} ms per loop iteration incl. snapshots and costs)")
        print(f"total energy {out['E'][0]:.6f} -> {out['E'][-1]:.6f}  (relative drift {abs(out['E'][-1] / out['E'][0] - 1):.2e});"
              f" field energy cost J_ee {out['J_ee'][0]:.4e} -> max {out['J_ee'].max():.4e};  J_KL end {out['J_KL'][-1]:.4e}")
    if args["is_save"]:
        path = os.path.join(args["save_file"], args["simcase"], "wo-oc")
        os.makedirs(path, exist_ok=True)
        np.savez_compressed(os.path.join(path, "data.npz"), **out)
    sim.close()
    return out


if __name__ == "__main__":
    main()
