#!/bin/bash
# Kernel trace + HBM traffic (two PMC passes) of BASELINE config 3 with fixed-point positions, as collect.sh does for
# config 2:  bash profiles/collect_cfg3.sh   -> profiles/r2cfg3_{kernel_stats.csv,summary.json,summary.md}
set -o pipefail
TAG=r2cfg3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
ARGS="--envs 128 --mesh 512 --dtype float32 --positions fixed32 --init two-stream --actions 3 --no-cpu-baseline"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 3 $ARGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python3 $ROOT/bench.py --steps 20 --warmup 3 $ARGS \
  > $OUT/${TAG}_bench_under_rocprofv3.json 2> $OUT/${TAG}_kt.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o pf -- python3 $ROOT/bench.py --steps 5 --warmup 1 $ARGS > /dev/null 2> $OUT/${TAG}_pf.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o pw -- python3 $ROOT/bench.py --steps 5 --warmup 1 $ARGS > /dev/null 2> $OUT/${TAG}_pw.err || exit 1
cd $ROOT && SUMMARY_WORKLOAD="bench.py $ARGS (BASELINE config 3: two-stream, N=1e6, Ng=512, 128 envs, float32 velocities + 32-bit fixed-point positions, a new action every step)" \
  python3 profiles/summarize.py $TAG $OUT/${TAG}_kt $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_summary.json profiles/${TAG}_summary.md $OUT/
