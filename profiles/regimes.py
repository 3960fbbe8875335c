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
    ("config 1: N=1e4, Ng=128, 1 env, fp64 (sweeps)", "--steps 2000 --warmup 200 --envs 1 --particles 10000 --mesh 128"),
    ("config 2: N=1e6, Ng=256, 64 envs, fp64", "--steps 50 --warmup 5"),
    ("config 3: two-stream, N=1e6, Ng=512, 128 envs, fp32 x and v, a new random action every step", "--steps 50 --warmup 5 --envs 128 --mesh 512 --dtype float32 --init two-stream --actions 3"),
    ("config 3 with fixed-point positions", "--steps 50 --warmup 5 --envs 128 --mesh 512 --dtype float32 --positions fixed32 --init two-stream --actions 3"),
    ("config 3 shape without control (one pic_step call for all steps)", "--steps 50 --warmup 5 --envs 128 --mesh 512 --dtype float32 --positions fixed32"),
    ("config 4 share: N=4e6, Ng=1024, 64 envs, fp64", "--steps 20 --warmup 3 --particles 4000000 --mesh 1024"),
    ("config 5 share: N=1e7, Ng=256, 128 envs, fp32 push / fp64 mesh", "--steps 10 --warmup 2 --envs 128 --particles 10000000 --dtype float32"),
    ("config 5 share, fixed-point positions", "--steps 10 --warmup 2 --envs 128 --particles 10000000 --dtype float32 --positions fixed32"),
    ("Infinity-Cache regime: N=1e6, Ng=256, 12 envs, fp64", "--steps 200 --warmup 20 --envs 12"),
    ("one large environment: N=1e6, Ng=256, fp64", "--steps 300 --warmup 30 --envs 1"),
    ("reference shape x 64: N=5000, Ng=250, fp64 (resident)", "--steps 500 --warmup 50 --envs 64 --particles 5000 --mesh 250"),
    ("reference shape x 256 (resident)", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250"),
    ("reference shape x 1024 (resident)", "--steps 500 --warmup 50 --envs 1024 --particles 5000 --mesh 250"),
    ("reference shape x 256, sweeps (--blocks-per-env 2)", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --blocks-per-env 2"),
    ("reference shape x 256, fp32 + fixed-point positions (resident)", "--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --dtype float32 --positions fixed32"),
]


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    print("| workload | schedule | particle-steps/s | us/step | kernels (avg us per launch) | moved-bytes fraction of 8 TB/s | copy probe GB/s |")
    print("|---|---|---|---|---|---|---|")
    for k, (name, args) in enumerate(CASES):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args.split(), capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(f"| {name} | FAILED | | | {r.stderr[-200:]!r} | | |")
            continue
        open(os.path.join(out, f"case{k:02d}.json"), "w").write(lines[-1] + "\n")
        d = json.loads(lines[-1])
        kern = ", ".join(f"{n} {v['avg_ms'] * 1e3:.1f}" + (f" ({v['steps_per_launch']} steps in one launch)" if "steps_per_launch" in v else "")
                         for n, v in d["kernels"].items())
        print(f"| {name} | {d['config'].get('schedule', '')} | {d['value']:.3e} | {d['ms_per_step'] * 1e3:.1f} | {kern} | "
              f"{d['hbm_frac_of_step']:.3f} | {d['roofline']['measured_inplace_copy_GBs']:.0f} |", flush=True)


if __name__ == "__main__":
    main()
