#!/bin/bash
# usage: profiles/run_bpe.sh "<blocks-per-env list>" "<env counts>" [extra bench args]
for E in $2; do for B in $1; do
  timeout -k 10 120 python bench.py --envs $E --steps 40 --warmup 5 --no-cpu-baseline --blocks-per-env $B $3 2>/dev/null | B=$B python -c "
import sys, json, os
d = json.loads(sys.stdin.readline()); k = d['kernels']
g = lambda n: k[n]['avg_ms'] if n in k else float('nan')
print('bpe=%-4s E=%3d ms/step=%.4f ps/s=%.3e frac=%.3f A=%.4f B=%.4f C=%.4f D=%.4f solve=%.4f drift=%.1e' % (os.environ['B'], d['config']['envs_per_gpu'], d['ms_per_step'], d['value'], d['hbm_frac_of_step'], g('sweep_A'), g('sweep_B'), g('sweep_C'), g('sweep_D'), g('field_solve'), d['energy_drift']))"
done; done
