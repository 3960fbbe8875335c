#!/bin/bash
# Same-box A/B of the working tree against the round-1 tree (profiles/ab/r1, git archive of b4d101c with its built
# library): config 2, the Infinity-Cache regime (12 envs) and the reference's small-environment shape, interleaved.
# usage (on the GPU box, from the repo root): bash profiles/ab_r1.sh <outdir>
set -e
out=${1:-gpurun_out/ab}
mkdir -p $out
here=$(pwd)
run() {  # tag dir args...
  tag=$1; dir=$2; shift 2
  (cd $dir && python bench.py "$@" --no-cpu-baseline) > $out/$tag.json 2> $out/$tag.err || { echo "FAILED $tag"; tail -5 $out/$tag.err; exit 1; }
}
for r in 1 2; do
  run cfg2_new_$r . --steps 50 --warmup 5
  run cfg2_r1_$r profiles/ab/r1 --steps 50 --warmup 5
done
run env12_new . --steps 200 --warmup 20 --envs 12
run env12_r1 profiles/ab/r1 --steps 200 --warmup 20 --envs 12
for E in 64 256 1024; do
  run small${E}_new . --steps 500 --warmup 50 --envs $E --particles 5000 --mesh 250
  run small${E}_r1 profiles/ab/r1 --steps 500 --warmup 50 --envs $E --particles 5000 --mesh 250
done
run cfg1_new . --steps 2000 --warmup 200 --envs 1 --particles 10000 --mesh 128
run cfg1_r1 profiles/ab/r1 --steps 2000 --warmup 200 --envs 1 --particles 10000 --mesh 128
run cfg3_float . --steps 50 --warmup 5 --envs 128 --mesh 512 --dtype float32
run cfg3_fixed32 . --steps 50 --warmup 5 --envs 128 --mesh 512 --dtype float32 --positions fixed32
run cfg3_r1 profiles/ab/r1 --steps 50 --warmup 5 --envs 128 --mesh 512 --dtype float32
run cfg2_f64acc . --steps 50 --warmup 5 --accum float64
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = {n: round(v["avg_ms"] * 1e3, 1) for n, v in d["kernels"].items()}
    print(f"{os.path.basename(f):24s} {d['value']:.3e} p-steps/s  {d['ms_per_step']*1e3:9.1f} us/step  {k}  copy {d['roofline']['measured_inplace_copy_GBs']:.0f} GB/s")
PY
