#!/bin/bash
# resident-schedule A/B of two exported trees on the reference's environment shape: bash profiles/ab_small.sh <outdir> <treeA> <treeB>
out=$1; A=$2; B=$3
mkdir -p $out
for r in 1 2; do for E in 1 64 256 1024; do for t in $A $B; do
  (cd profiles/ab/$t && python bench.py --no-cpu-baseline --steps 500 --warmup 50 --envs $E --particles 5000 --mesh 250) > $out/res${E}_${t}_$r.json 2>/dev/null || echo FAILED
done; done; done
for t in $A $B; do (cd profiles/ab/$t && python bench.py --no-cpu-baseline --steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --dtype float32 --positions fixed32) > $out/res256fx_${t}_1.json 2>/dev/null; done
python - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "res*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f"{os.path.basename(f):28s} {d['ms_per_step']*1e3:9.1f} us/step")
    except Exception:
        print(os.path.basename(f), "unreadable")
PY
