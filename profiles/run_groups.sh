#!/bin/bash
# usage: profiles/run_groups.sh "<group MB list>" "<streams list>" "<blocks-per-env list>" [extra bench args]
for S in $2; do for G in $1; do for B in $3; do
  PICSTEP_GROUP_MB=$G PICSTEP_STREAMS=$S timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --blocks-per-env $B $4 2>/dev/null | G=$G S=$S B=$B python -c "
import sys, json, os
d = json.loads(sys.stdin.readline()); k = d['kernels']
g = lambda n: k[n]['avg_ms'] if n in k else float('nan')
print('groupMB=%-5s streams=%s bpe=%-4s ms/step=%.4f ps/s=%.3e frac=%.3f B=%.4f C=%.4f D=%.4f solve=%.4f drift=%.1e' % (os.environ['G'], os.environ['S'], os.environ['B'], d['ms_per_step'], d['value'], d['hbm_frac_of_step'], g('sweep_B'), g('sweep_C'), g('sweep_D'), g('field_solve'), d['energy_drift']))"
done; done; done
