#!/bin/bash
# config 2 only, alternating order: bash profiles/ab_cfg2.sh <treeA> <treeB> [rounds]
A=$1; B=$2; rounds=${3:-4}
for r in $(seq 1 $rounds); do
  if [ $((r % 2)) = 1 ]; then order="$A $B"; else order="$B $A"; fi
  for t in $order; do
    (cd profiles/ab/$t && python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null) | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$t $r', round(d['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, round(d['roofline']['measured_inplace_copy_GBs']))"
  done
done
