#!/bin/bash
# timelines of the latency-bound regimes with the diagnostic library: bash profiles/timeline/run_all.sh <outdir>
out=${1:-gpurun_out/timeline}; mkdir -p $out
T="python profiles/timeline/timeline.py"
$T --envs 1 --particles 1000000 --mesh 256 --steps 3 > $out/one_1e6.md 2> $out/one_1e6.err || echo "one_1e6 failed"
$T --envs 4 --particles 1000000 --mesh 256 --steps 3 > $out/four_1e6.md 2> $out/four_1e6.err || echo "four_1e6 failed"
$T --envs 1 --particles 10000 --mesh 128 --steps 3 > $out/config1.md 2> $out/config1.err || echo "config1 failed"
$T --envs 1 --particles 100000 --mesh 256 --steps 3 > $out/one_1e5.md 2> $out/one_1e5.err || echo "one_1e5 failed"
$T --envs 1 --particles 5000 --mesh 250 --steps 1 --calls 6 > $out/res_single.md 2> $out/res_single.err || echo "res_single failed"
$T --envs 1 --particles 5000 --mesh 250 --steps 10 --calls 2 > $out/res_ten.md 2> $out/res_ten.err || echo "res_ten failed"
$T --envs 256 --particles 5000 --mesh 250 --steps 1 --calls 6 > $out/res256_single.md 2> $out/res256_single.err || echo "res256_single failed"
ls -la $out
