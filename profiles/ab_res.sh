#!/bin/bash
# resident-schedule A/B of exported trees, multi-step calls and single-step calls:  bash profiles/ab_res.sh <outdir> <rounds> <tree> ...
out=$1; rounds=$2; shift 2
mkdir -p $out
cases=(
 "res1|--steps 500 --warmup 50 --envs 1 --particles 5000 --mesh 250"
 "res64|--steps 500 --warmup 50 --envs 64 --particles 5000 --mesh 250"
 "res256|--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250"
 "res1024|--steps 500 --warmup 50 --envs 1024 --particles 5000 --mesh 250"
 "res256fx|--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --dtype float32 --positions fixed32"
 "res256act|--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --actions 3"
 "res256act1|--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --actions 3 --per-step-calls"
)
for r in $(seq 1 $rounds); do for c in "${cases[@]}"; do tag=${c%%|*}; args=${c#*|}; for t in "$@"; do
  (cd profiles/ab/$t && python bench.py --no-cpu-baseline --steady-steps 0 $args) > $out/${tag}_${t}_$r.json 2>/dev/null || (cd profiles/ab/$t && python bench.py --no-cpu-baseline ${args/ --per-step-calls/}) > $out/${tag}_${t}_$r.json 2>/dev/null || echo "FAILED $tag $t"
done; done; done
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f"{os.path.basename(f):28s} {d['ms_per_step']*1e3:9.2f} us/step  " + str({n: round(v['avg_ms'] * 1e3, 1) for n, v in d['kernels'].items()}))
    except Exception as e:
        print(os.path.basename(f), "unreadable")
PY
