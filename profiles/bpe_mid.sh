#!/bin/bash
# step time of 1..16 large environments with the automatic workgroup count, for an exported tree:  bash profiles/bpe_mid.sh <tree> [<tree> ...]
for r in 1 2; do for E in 1 2 3 4 5 6 7 8 10 12 16; do for t in "$@"; do
  (cd profiles/ab/$t && python bench.py --no-cpu-baseline --steady-steps 0 --steps 300 --warmup 30 --envs $E 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t E=$E', round(d['ms_per_step']*1e3,1), {k: (round(v['avg_ms']*1e3,1), v['launches']) for k,v in d['kernels'].items() if k=='sweep_B'})")
done; done; done
