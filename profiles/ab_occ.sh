#!/bin/bash
# Sweeps B and D had grown to 106 scalar registers in round 3 (the control inputs): 7 waves per SIMD = 3 workgroups per CU instead
# of 4.  Interleaved same-box A/B of two exported trees (profiles/mk_ab.sh) over the regimes the sweeps run in:
#   bash profiles/ab_occ.sh <outdir> <treeA> <treeB> [rounds]
out=$1; A=$2; B=$3; rounds=${4:-3}
mkdir -p $out
cases=(
 "cfg2|--steps 100 --warmup 20"
 "cfg3float|--config 3 --positions float --steps 100 --warmup 20"
 "cfg3fixed|--config 3 --steps 100 --warmup 20"
 "cfg4|--config 4 --steps 30 --warmup 5"
 "env12|--steps 200 --warmup 20 --envs 12"
 "env1|--steps 600 --warmup 60 --envs 1"
 "cfg1|--config 1 --steps 2000 --warmup 200"
 "s256seq|--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --blocks-per-env 2"
)
for r in $(seq 1 $rounds); do for c in "${cases[@]}"; do tag=${c%%|*}; args=${c#*|}; for t in $A $B; do
  (cd profiles/ab/$t && python3 bench.py --no-cpu-baseline --steady-steps 200 $args) > $out/${tag}_${t}_$r.json 2>/dev/null || echo "FAILED $tag $t"
done; done; done
python3 - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f"{os.path.basename(f):28s} {d['ms_per_step']*1e3:9.1f} us/step (steady {d['steady_state']['ms_per_step']*1e3:9.1f})  " + str({n: round(v['avg_ms'] * 1e3, 1) for n, v in d['kernels'].items()}))
    except Exception as e:
        print(os.path.basename(f), "unreadable")
PY
