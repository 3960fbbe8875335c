#!/bin/bash
# Workgroups per environment for the two largest BASELINE shares (round 4): the automatic choice aimed at 8192 workgroups in all,
# which at config 5's share is 156 000 particles -- 340 us -- per workgroup and a last partial round of workgroups of the same length.
#   bash profiles/bpe_big.sh
for rep in 1 2; do for cfg in "4 128" "4 160" "4 192" "4 256" "4 384" "4 512" "5 64" "5 256" "5 512" "5 768" "5 1024" "5 2048"; do set -- $cfg
  python3 bench.py --no-cpu-baseline --config $1 --blocks-per-env $2 --steps 20 --warmup 3 --steady-steps 60 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config $1 bpe $2:', round(d['ms_per_step']*1e3,1), 'steady', round(d['steady_state']['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d['placement']['outcome'], round(d['placement']['kept_GBs']))"
done; done
