#!/usr/bin/env python3
"""One bench.py line per regime (unprofiled; kernel times from HIP events on the library's stream), as a markdown table:
    python profiles/regimes.py <outdir> > table.md
BASELINE configs 1-5 at their single-GPU share, the Infinity-Cache regime and the reference's small-environment shape."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [
    ("config 1: N=1e4, Ng=128, 1 env, fp64 (sweeps)", "--config 1 --steps 2000 --warmup 200"),
    ("config 2: N=1e6, Ng=256, 64 envs, fp64", "--config 2 --steps 200 --warmup 20"),
    ("config 3 as specified: two-stream, N=1e6, Ng=512, 128 envs, fp32 + fixed-point positions, a new random action every step (one pic_step_actions_traj call)", "--config 3 --steps 50 --warmup 5"),
    ("config 3, one pic_step_actions call per step (a trainer's loop)", "--config 3 --steps 50 --warmup 5 --per-step-calls"),
    ("config 3 in plain fp32 (float32 positions: BASELINE's wording; the preset above is the recommended float32 format)", "--config 3 --positions float --steps 50 --warmup 5"),
    ("config 3 shape without control (one pic_step call for all steps)", "--config 3 --actions 0 --steps 50 --warmup 5"),
    ("config 4 share: N=4e6, Ng=1024, 64 envs, fp64", "--config 4 --steps 20 --warmup 3 --steady-steps 50"),
    ("config 5 share: N=1e7, Ng=256, 128 envs, fp32 push / fp64 mesh", "--config 5 --steps 10 --warmup 2 --steady-steps 20"),
    ("config 5 share, fixed-point positions", "--config 5 --positions fixed32 --steps 10 --warmup 2 --steady-steps 20"),
    ("Infinity-Cache regime: N=1e6, Ng=256, 12 envs, fp64", "--steps 200 --warmup 20 --envs 12"),
    ("four large environments: N=1e6, Ng=256, fp64", "--steps 400 --warmup 40 --envs 4"),
    ("two large environments", "--steps 400 --warmup 40 --envs 2"),
    ("one large environment: N=1e6, Ng=256, fp64", "--steps 600 --warmup 60 --envs 1"),
    ("one environment of N=1e5", "--steps 1000 --warmup 100 --envs 1 --particles 100000"),
    ("reference shape x 1: N=5000, Ng=250, fp64 (resident)", "--steps 500 --warmup 50 --envs 1 --particles 5000 --mesh 250"),
    ("reference shape x 64 (resident)", "--steps 500 --warmup 50 --envs 64 --particles 5000 --mesh 250"),
    ("reference shape x 256 (resident)", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250"),
    ("reference shape x 1024 (resident)", "--steps 500 --warmup 50 --envs 1024 --particles 5000 --mesh 250"),
    ("reference shape x 1, every step's refresh made and recorded (pic_step_history)", "--steps 500 --warmup 50 --envs 1 --particles 5000 --mesh 250 --history"),
    ("reference shape x 64, every step's refresh made and recorded", "--steps 500 --warmup 50 --envs 64 --particles 5000 --mesh 250 --history"),
    ("reference shape x 256, every step's refresh made and recorded", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --history"),
    ("reference shape x 1024, every step's refresh made and recorded", "--steps 500 --warmup 50 --envs 1024 --particles 5000 --mesh 250 --history"),
    ("reference shape x 256, a new action every step, one call (resident)", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --actions 5"),
    ("reference shape x 256, one pic_step_actions call per step (resident)", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --actions 5 --per-step-calls"),
    ("reference shape x 256, sweeps (--blocks-per-env 2)", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --blocks-per-env 2"),
    ("reference shape x 256, fp32 + fixed-point positions (resident)", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --dtype float32 --positions fixed32"),
    ("64 x N=8000, float32 (resident, 16 particles per lane)", "--steps 500 --warmup 50 --envs 64 --particles 8000 --mesh 250 --dtype float32"),
    ("64 x N=5000, float32, TSC (resident)", "--steps 500 --warmup 50 --envs 64 --particles 5000 --mesh 250 --dtype float32 --interpol TSC"),
    ("64 x N=20000, Ng=128, fp64 (sweeps)", "--steps 500 --warmup 50 --envs 64 --particles 20000 --mesh 128"),
]


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    print("`refresh`: whether the post-step refresh of SURVEY 8d's step (density, E_mesh, phi, KE / PE: pic.py:145-146) is made after EVERY "
          "step of a call or only after the last one (a plain pic_step(nsteps) call of the resident schedule skips the refreshes nothing "
          "can observe; the streaming schedule, pic_step_history and one-step calls make them all).\n")
    print("| workload | schedule | refresh | particle-steps/s | us/step | steady-state us/step | first steps of the handle, us/step | kernels (avg us per launch) | roofline of the dominant kernel | copy probe GB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for k, (name, args) in enumerate(CASES):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args.split(), capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(f"| {name} | FAILED | | | | | | {r.stderr[-200:]!r} | | |")
            continue
        open(os.path.join(out, f"case{k:02d}.json"), "w").write(lines[-1] + "\n")
        d = json.loads(lines[-1])
        kern = ", ".join(f"{n} {v['avg_ms'] * 1e3:.1f}" + (f" ({v['steps_per_launch']} steps per launch)" if "steps_per_launch" in v else "")
                         for n, v in d["kernels"].items())
        rf = d["roofline"]
        if rf["bound"] == "hbm":
            roof = f"{rf['kernel']}: {rf['achieved'] / 1e3:.2f} of 8 TB/s = {rf['frac']:.3f} (whole step, moved bytes: {d['hbm_frac_of_step']:.3f})"
        elif rf["achieved"] is not None:
            roof = (f"VALU issue: {rf['valu_wave_insts_per_particle_step'] * 64:.0f} instructions per particle-step, {rf['achieved']:.0f} of "
                    f"{rf['peak']:.0f} Gwave-inst/s = {rf['frac']:.3f} of the chip, {rf['frac_of_the_CUs_in_use']:.3f} of the CUs in use")
        else:
            roof = "VALU issue + latency (no counter pass for this format)"
        ss = d.get("steady_state") or {}
        cs = d.get("cold_start") or {}
        print(f"| {name} | {d['config'].get('schedule', '')} | {d['config'].get('refresh', '')} | {d['value']:.3e} | {d['ms_per_step'] * 1e3:.1f} | "
              f"{ss.get('ms_per_step', float('nan')) * 1e3:.1f} | {cs.get('ms_per_step', float('nan')) * 1e3:.1f} | {kern} | {roof} | "
              f"{rf['measured_inplace_copy_GBs']:.0f} |", flush=True)


if __name__ == "__main__":
    main()
