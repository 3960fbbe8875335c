#!/bin/bash
# usage: profiles/run_variants.sh "<variant names>" "<env counts>" [extra bench args] -- bench.py per experiment build
D="$GRAFT_REPO_ROOT/optimal-control-1d-electrostatic-plasma_amd/csrc"
for E in $2; do for V in $1; do
  if [ "$V" = base ]; then unset PICSTEP_LIB; else export PICSTEP_LIB=$D/exp/libpicstep_$V.so; fi
  timeout -k 10 120 python bench.py --envs $E --steps 40 --warmup 5 --no-cpu-baseline $3 2>/dev/null | V=$V python -c "
import sys, json, os
d = json.loads(sys.stdin.readline()); k = d['kernels']
g = lambda n: k[n]['avg_ms'] if n in k else float('nan')
print('%-12s E=%3d ms/step=%.4f ps/s=%.3e frac=%.3f A=%.4f B=%.4f C=%.4f D=%.4f solve=%.4f copy=%.0f' % (os.environ['V'], d['config']['envs_per_gpu'], d['ms_per_step'], d['value'], d['hbm_frac_of_step'], g('sweep_A'), g('sweep_B'), g('sweep_C'), g('sweep_D'), g('field_solve'), d['roofline']['measured_inplace_copy_GBs']))"
done; done
