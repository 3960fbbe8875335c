#!/usr/bin/env python3
"""In-kernel timeline of the streaming sweeps (and of the resident kernel) from the DIAGNOSTIC build of the library
(profiles/timeline/build.sh -> profiles/bin/libpicstep_timeline.so; the product library has no stamps).

    python profiles/timeline/timeline.py [--envs 1 --particles 1000000 --mesh 256 --steps 6] > table.md

Per launch: when the first / last workgroup started and ended (100 MHz wall clock, so workgroups can be compared), the gap
to the previous launch, and the median / max over workgroups of every interval between two hooks (shader clock converted
with the clock ratio measured in the same launch).  Stamping perturbs: read intervals against each other."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ocplasma_amd  # noqa: E402
from ocplasma_amd import _abi, _build  # noqa: E402

DIAG = os.path.join(ROOT, "profiles", "bin", "libpicstep_timeline.so")
_build.LIB = DIAG            # this process loads the stamped library instead of csrc/libpicstep.so
_build.needs_build = lambda: False
TL = (64, 1024, 32)

SWEEP_SLOTS = [(0, 2, "entry -> first tile requested"), (2, 3, "accumulator row arrived, b in LDS (+barrier)"),
               (3, 4, "scan by wave 0 (+barrier)"), (4, 5, "field tile (+barrier)"), (5, 6, "clear LDS mesh (+barrier)")]
RES_SLOTS = [(0, 2, "entry -> particle loads issued"), (2, 3, "q1 row / entry deposit ready (+barrier)")]


def us(ticks, ghz):
    return ticks / (ghz * 1e3)


def analyse(rec, nl, resident, out):
    prev_end = None
    for l in range(nl):
        r = rec[l]
        live = r[:, 26] != 0
        if not live.any():
            continue
        w = r[live]
        t0r, t1r = w[:, 1].astype(np.int64), w[:, 27].astype(np.int64)
        dur_r = (t1r - t0r)
        clk = np.median((w[:, 26].astype(np.int64) - w[:, 0].astype(np.int64)) / np.maximum(dur_r, 1)) * 0.1   # GHz
        first, last_start, first_end, last_end = t0r.min(), t0r.max(), t1r.min(), t1r.max()
        gap = (first - prev_end) * 0.01 if prev_end is not None else float("nan")
        prev_end = last_end
        gx, gy = int(w[0, 29] >> 32), int(w[0, 29] & 0xFFFFFFFF)
        out.append(f"\n### launch {l}: grid ({gx}, {gy}), {live.sum()} stamped workgroups, shader clock {clk:.2f} GHz\n")
        out.append(f"gap since the previous launch's last exit {gap:.2f} us; workgroup starts spread over {(last_start - first) * 0.01:.2f} us; "
                   f"first exit at {(first_end - first) * 0.01:.2f}, last exit at {(last_end - first) * 0.01:.2f} us after the first start\n")
        out.append("| interval | median us | p90 | max |\n|---|---|---|---|\n")

        def row(a, b, name):
            ok = (w[:, a] != 0) & (w[:, b] != 0)
            if not ok.any():
                return
            d = us((w[ok, b].astype(np.int64) - w[ok, a].astype(np.int64)).astype(float), clk)
            out.append(f"| {name} | {np.median(d):.2f} | {np.percentile(d, 90):.2f} | {d.max():.2f} |\n")

        slots = RES_SLOTS if resident else SWEEP_SLOTS
        for a, b, name in slots:
            row(a, b, name)
        if resident:
            for k, nm in enumerate(("B", "C", "D")):
                row(3 if k == 0 else 7 + 2 * k, 8 + 2 * k, f"first step, sub-stage {nm}: field phase")
                row(8 + 2 * k, 9 + 2 * k, f"first step, sub-stage {nm}: particle phase (+barrier)")
            row(20, 14, "last step, sub-stage B: field phase + the previous step's refresh riding with it")
            for k, nm in enumerate(("B", "C", "D")):
                if k:
                    row(13 + 2 * k, 14 + 2 * k, f"last step, sub-stage {nm}: field phase")
                row(14 + 2 * k, 15 + 2 * k, f"last step, sub-stage {nm}: particle phase (+barrier)")
            row(24, 25, "post-step refresh of the last step")
            row(25, 26, "particle stores issued -> all memory operations done")
        else:
            prev = 6
            for t in range(8):
                if not (w[:, 8 + 2 * t] != 0).any():
                    break
                row(prev, 8 + 2 * t, f"tile {t}: wait for its loads (and the previous stores)")
                row(8 + 2 * t, 9 + 2 * t, f"tile {t}: push + stores issued")
                prev = 9 + 2 * t
            row(prev, 24, "loop exit (+barrier: slowest wave of the workgroup)")
            row(24, 25, "flush: atomics issued")
            row(25, 26, "all memory operations done")
        row(0, 26, "whole workgroup")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--particles", type=int, default=1_000_000)
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--steps", type=int, default=4, help="environment steps recorded (one pic_step call)")
    ap.add_argument("--calls", type=int, default=1, help="number of pic_step calls of --steps steps each")
    ap.add_argument("--blocks-per-env", type=int, default=0)
    ap.add_argument("--dtype", default="float64")
    args = ap.parse_args()
    from ocplasma_amd.env.batched import BatchedPIC
    lib = _abi.load()
    lib.pic_timeline_reset.restype = C.c_int
    lib.pic_timeline_read.restype = C.c_int
    lib.pic_timeline_read.argtypes = [C.c_void_p]
    env = BatchedPIC(args.envs, args.particles, args.mesh, L=50.0, dt=0.1, dtype=args.dtype, blocks_per_env=args.blocks_per_env)
    env.reset_sampled("bump-on-tail", seed=11)
    for _ in range(3):
        env.step(None, nsteps=50)
    env.sync()
    assert lib.pic_timeline_reset() == 0
    for _ in range(args.calls):
        env.step(None, nsteps=args.steps)
    env.sync()
    rec = np.zeros(TL, dtype=np.uint64)
    nl = lib.pic_timeline_read(rec.ctypes.data_as(C.c_void_p))
    assert nl >= 0
    resident = env._h.schedule() == "resident"
    out = [f"## timeline: {args.envs} x N={args.particles}, Ng={args.mesh}, {args.dtype}, {env._h.schedule()} schedule, "
           f"{args.calls} call(s) of {args.steps} step(s), {min(nl, TL[0])} launches recorded\n"]
    analyse(rec, min(nl, TL[0]), resident, out)
    print("".join(out))
    env.close()


if __name__ == "__main__":
    main()
