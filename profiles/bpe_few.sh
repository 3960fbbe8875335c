#!/bin/bash
# workgroups per environment for one to four large environments: bash profiles/bpe_few.sh <outdir> [tree]
out=$1; tree=${2:-.}
mkdir -p $out
for E in 1 2 4; do for bpe in 0 61 122 245 489 977; do
  (cd $tree && python bench.py --no-cpu-baseline --steady-steps 0 --steps 400 --warmup 40 --envs $E --blocks-per-env $bpe) > $out/e${E}_bpe${bpe}.json 2>/dev/null || echo FAILED
done; done
python - $out <<'PY'
import json, sys, glob, os, re
for f in sorted(glob.glob(os.path.join(sys.argv[1], "e*_bpe*.json")), key=lambda f: [int(v) for v in re.findall(r"\d+", os.path.basename(f))]):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f"{os.path.basename(f):22s} {d['ms_per_step']*1e3:8.1f} us/step  " + str({n: round(v['avg_ms'] * 1e3, 1) for n, v in d['kernels'].items()}))
    except Exception:
        print(os.path.basename(f), "unreadable")
PY
