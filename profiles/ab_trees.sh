#!/bin/bash
# bash profiles/ab_trees.sh "<trees>" "<bench args>" [rounds]: config-2-style bench of several exported trees, interleaved
trees=$1; args=$2; rounds=${3:-2}
for r in $(seq 1 $rounds); do for t in $trees; do
  (cd profiles/ab/$t && python bench.py --no-cpu-baseline --steps 100 --warmup 10 --profile-steps 20 $args 2>/dev/null) | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-12s' % '$t', '$r', round(d['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d.get('placement'))"
done; done
