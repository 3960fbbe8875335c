#!/bin/bash
# streaming schedule under control, interleaved A/B of exported trees:  bash profiles/ab_control.sh <outdir> <rounds> <tree> ...
out=$1; rounds=$2; shift 2
mkdir -p $out
cases=(
 "cfg3traj|--config 3 --steps 60 --warmup 10 --steady-steps 100"
 "cfg3calls|--config 3 --steps 60 --warmup 10 --steady-steps 100 --per-step-calls"
 "cfg3free|--config 3 --actions 0 --steps 60 --warmup 10 --steady-steps 100"
 "cfg2act|--config 2 --actions 5 --steps 60 --warmup 10 --steady-steps 100"
 "s256act|--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --blocks-per-env 2 --actions 5 --steady-steps 0"
 "cfg1act|--config 1 --actions 3 --steps 2000 --warmup 200 --steady-steps 0"
)
for r in $(seq 1 $rounds); do for c in "${cases[@]}"; do tag=${c%%|*}; args=${c#*|}; for t in "$@"; do
  (cd profiles/ab/$t && python bench.py --no-cpu-baseline $args) > $out/${tag}_${t}_$r.json 2>/dev/null || echo "FAILED $tag $t"
done; done; done
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        ss = (d.get("steady_state") or {}).get("ms_per_step")
        print(f"{os.path.basename(f):28s} {d['ms_per_step']*1e3:9.1f} us/step  steady {ss * 1e3 if ss else float('nan'):9.1f}")
    except Exception as e:
        print(os.path.basename(f), "unreadable")
PY
