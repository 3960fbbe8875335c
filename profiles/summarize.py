#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace --stats run + two --pmc passes) into the summary files
committed under profiles/.  Usage:
    python profiles/summarize.py <tag> <kernel_trace_dir> <pmc_fetch_dir> <pmc_write_dir>
writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.json and profiles/<tag>_summary.md.
HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide (16 B/lane) coalesced read, so it is doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import sys

STAGES = {"0": "sweep_A", "1": "sweep_B", "2": "sweep_C", "3": "sweep_D", "4": "sweep_refresh", "5": "sweep_probe",
          "6": "sweep_B2", "7": "sweep_D2"}      # B2 / D2: sweeps B and D of the inner steps of a multi-step call


def short(name):
    if "sweep_kernel<" in name:
        args = [a.strip() for a in name.split("sweep_kernel<")[1].split(">")[0].split(",")]
        return STAGES.get(args[3], "sweep_" + args[3])
    if "resident_kernel<" in name:
        return "resident"
    if "field_solve_kernel" in name:
        return "field_solve"
    if "stream_probe" in name:
        return "stream_probe"
    return None


def one(pattern):
    fs = glob.glob(pattern, recursive=True)
    return fs[0] if fs else None


def main():
    tag, kt, pf, pw = sys.argv[1:5]
    here = os.path.dirname(os.path.abspath(__file__))
    out = {"tag": tag, "kernels": {}}
    stats = one(os.path.join(kt, "**", "*_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if short(r["Name"])]
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel"] + list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            w.writerow({"Kernel": short(r["Name"]), **r})
    for r in keep:
        out["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                            "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                            "pct_of_gpu_time": float(r["Percentage"])}
    for ctr, d, factor in (("FETCH_SIZE", pf, 2.0), ("WRITE_SIZE", pw, 1.0)):
        f = one(os.path.join(d, "**", "*_counter_collection.csv"))
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == ctr:
                agg[k].append(float(r["Counter_Value"]) * 1024.0 * factor)
        for k, v in agg.items():
            out["kernels"].setdefault(k, {})["hbm_%s_bytes_per_launch" % ("read" if ctr == "FETCH_SIZE" else "write")] = \
                sum(v) / len(v)
    for k, d in out["kernels"].items():
        if "hbm_read_bytes_per_launch" in d and "hbm_write_bytes_per_launch" in d:
            d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    json.dump(out, open(os.path.join(here, f"{tag}_summary.json"), "w"), indent=1)
    with open(os.path.join(here, f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}`\n\n{os.environ.get('SUMMARY_WORKLOAD', 'bench.py config 2 (N=1e6, Ng=256, 64 envs, fp64)')}. Durations: "
                "`--kernel-trace --stats`; HBM bytes: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, "
                "FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes.\n\n"
                "| kernel | calls | avg us | min us | max us | HBM read MB | HBM write MB | HBM total MB |\n|---|---|---|---|---|---|---|---|\n")
        for k in sorted(out["kernels"]):
            d = out["kernels"][k]
            g = lambda n, s=1.0: ("%.1f" % (d[n] / s)) if n in d else "-"
            f.write(f"| {k} | {d.get('calls', '-')} | {g('avg_us')} | {g('min_us')} | {g('max_us')} | "
                    f"{g('hbm_read_bytes_per_launch', 1e6)} | {g('hbm_write_bytes_per_launch', 1e6)} | {g('hbm_bytes_per_launch', 1e6)} |\n")
    print(open(os.path.join(here, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main()
