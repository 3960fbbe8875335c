#!/usr/bin/env python3
"""VALU wave-instructions per particle-step of the resident kernel from a `rocprofv3 --pmc SQ_INSTS_VALU` pass of
`bench.py --envs E --particles N --mesh Ng --steps K --warmup K --steady-steps 0 --profile-steps 0` (every launch K steps):
    python profiles/resident_valu.py <tag> <pmc_dir> <key> <N> <E> <K>
adds `resident_valu_wave_insts_per_particle_step[key]` to profiles/<tag>_summary.json (bench.py prices the resident schedule's
roofline with it: achieved = this x particle-steps/s)."""
import csv, glob, json, os, sys

tag, d, key, N, E, K = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
here = os.path.dirname(os.path.abspath(__file__))
vals = []
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "resident_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            vals.append(float(r["Counter_Value"]))
if not vals:
    sys.exit("no resident_kernel rows with SQ_INSTS_VALU under " + d)
w = sum(vals) / len(vals) / (N * E * K)
path = os.path.join(here, f"{tag}_summary.json")
out = json.load(open(path)) if os.path.exists(path) else {"tag": tag, "kernels": {}}
out.setdefault("resident_valu_wave_insts_per_particle_step", {})[key] = w
json.dump(out, open(path, "w"), indent=1)
print(f"{key}: {len(vals)} launches, {w:.4f} VALU wave-instructions per particle-step ({w * 64:.1f} per particle and lane)")
