#!/bin/bash
# workgroups per environment vs step time, for a given ensemble: bash profiles/bpe_sweep.sh <outdir> <envs> <particles> <mesh> <steps> "<bpe list>" [extra bench args]
out=$1; E=$2; N=$3; Ng=$4; K=$5; list=$6; shift 6
mkdir -p $out
for r in 1 2; do for b in $list; do
  python bench.py --no-cpu-baseline --steps $K --warmup $((K/10+2)) --envs $E --particles $N --mesh $Ng --blocks-per-env $b "$@" > $out/bpe_${E}x${N}_${b}_$r.json 2>/dev/null || echo "failed $b"
done; done
python - $out $E $N <<'PY'
import json, sys, glob, os, re
out, E, N = sys.argv[1:4]
rows = {}
for f in sorted(glob.glob(os.path.join(out, f"bpe_{E}x{N}_*.json"))):
    b = int(re.search(r"_(\d+)_\d\.json$", f).group(1))
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    rows.setdefault(b, []).append((d["ms_per_step"] * 1e3, {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels"].items()}))
for b in sorted(rows):
    print(f"bpe {b:4d}: " + "  ".join(f"{t:8.1f} us/step {k}" for t, k in rows[b]))
PY
