#!/bin/bash
# float32 particles with FLOAT positions at config 3's shape: sweep D read 344 us in round 2 and 364-366 in round 3.
# Interleaved same-box runs of the round-2 tree, the working tree and a variant whose float32 sweep D stores plainly instead
# of write-through (profiles/mk_ab.sh r2 783b15d / head WORK / plainD WORK + the two-line patch of experiments_r4.md 4).
# usage: bash profiles/ab_f32d.sh <outdir> [rounds]
out=$1; rounds=${2:-3}
mkdir -p $out
args="--no-cpu-baseline --steps 100 --warmup 20 --envs 128 --mesh 512 --dtype float32 --positions float --init two-stream --actions 3"
for r in $(seq 1 $rounds); do for t in r2 head plainD; do
  extra="--steady-steps 0"; [ $t = r2 ] && extra=""
  (cd profiles/ab/$t && python3 bench.py $args $extra) > $out/f32d_${t}_$r.json 2> $out/f32d_${t}_$r.err || echo "FAILED $t $r"
done; done
python3 - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "f32d_*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f"{os.path.basename(f):28s} {d['ms_per_step']*1e3:9.1f} us/step  " + str({n: round(v['avg_ms'] * 1e3, 1) for n, v in d['kernels'].items()}))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
