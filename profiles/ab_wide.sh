#!/bin/bash
# Interleaved same-box A/B of exported trees (profiles/mk_ab.sh) over every regime:  bash profiles/ab_wide.sh <outdir> <rounds> <tree> ...
out=$1; rounds=$2; shift 2
mkdir -p $out
cases=(
 "one1e6|--steps 600 --warmup 60 --envs 1"
 "two1e6|--steps 400 --warmup 40 --envs 2"
 "three1e6|--steps 400 --warmup 40 --envs 3"
 "four1e6|--steps 400 --warmup 40 --envs 4"
 "cfg1|--steps 2000 --warmup 200 --envs 1 --particles 10000 --mesh 128"
 "one1e5|--steps 1000 --warmup 100 --envs 1 --particles 100000"
 "env12|--steps 200 --warmup 20 --envs 12"
 "cfg2|--steps 100 --warmup 10"
 "cfg3fx|--steps 50 --warmup 5 --envs 128 --mesh 512 --dtype float32 --positions fixed32"
 "cfg3f32|--steps 50 --warmup 5 --envs 128 --mesh 512 --dtype float32"
 "s256seq|--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250 --blocks-per-env 2"
 "res256|--steps 500 --warmup 50 --envs 256 --particles 5000 --mesh 250"
 "res1024|--steps 500 --warmup 50 --envs 1024 --particles 5000 --mesh 250"
)
for r in $(seq 1 $rounds); do for c in "${cases[@]}"; do tag=${c%%|*}; args=${c#*|}; for t in "$@"; do
  (cd profiles/ab/$t && python bench.py --no-cpu-baseline --steady-steps 0 $args) > $out/${tag}_${t}_$r.json 2>/dev/null || (cd profiles/ab/$t && python bench.py --no-cpu-baseline $args) > $out/${tag}_${t}_$r.json 2>/dev/null || echo "FAILED $tag $t"
done; done; done
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f"{os.path.basename(f):28s} {d['ms_per_step']*1e3:9.1f} us/step  " + str({n: round(v['avg_ms'] * 1e3, 1) for n, v in d['kernels'].items()}))
    except Exception as e:
        print(os.path.basename(f), "unreadable")
PY
