#!/bin/bash
# usage: profiles/run_runtiles.sh "<run tiles list (0 = one contiguous chunk per workgroup)>" "<bpe list>" [extra bench args]
for B in $2; do for T in $1; do
  PICSTEP_RUN_TILES=$T timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --blocks-per-env $B $3 2>/dev/null | T=$T B=$B python -c "
import sys, json, os
d = json.loads(sys.stdin.readline()); k = d['kernels']
g = lambda n: k[n]['avg_ms'] if n in k else float('nan')
print('run_tiles=%-3s bpe=%-4s ms/step=%.4f ps/s=%.3e frac=%.3f B=%.4f C=%.4f D=%.4f solve=%.4f drift=%.1e bad=%d' % (os.environ['T'], os.environ['B'], d['ms_per_step'], d['value'], d['hbm_frac_of_step'], g('sweep_B'), g('sweep_C'), g('sweep_D'), g('field_solve'), d['energy_drift'], d['bad_positions']))"
done; done
