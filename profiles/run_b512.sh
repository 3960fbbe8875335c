#!/bin/bash
# usage: profiles/run_b512.sh "<bpe list>" "<max R list>"  -- 512-thread build vs blocks per env and mesh replicas
D="$GRAFT_REPO_ROOT/optimal-control-1d-electrostatic-plasma_amd/csrc"
for B in $1; do for R in $2; do
  PICSTEP_LIB=$D/exp/libpicstep_b512.so PICSTEP_MAX_R=$R timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --blocks-per-env $B 2>/dev/null | B=$B R=$R python -c "
import sys, json, os
d = json.loads(sys.stdin.readline()); k = d['kernels']
g = lambda n: k[n]['avg_ms'] if n in k else float('nan')
print('b512 bpe=%-4s maxR=%s ms/step=%.4f ps/s=%.3e frac=%.3f B=%.4f C=%.4f D=%.4f solve=%.4f' % (os.environ['B'], os.environ['R'], d['ms_per_step'], d['value'], d['hbm_frac_of_step'], g('sweep_B'), g('sweep_C'), g('sweep_D'), g('field_solve')))"
done; done
