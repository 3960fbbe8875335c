#!/bin/bash
# Collect the judged artifact set on the GPU box (one gpurun call):
#   bash profiles/collect.sh r1
# writes gpurun_out/<tag>_* ; copy the summaries into profiles/ afterwards (the kernel trace is trimmed by summarize.py).
set -o pipefail
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. unprofiled bench line (with the CPU baseline leg)
python3 $ROOT/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
# 2. kernel trace + stats of the same command (its own JSON line is kept: the profiled process runs slower)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline \
  > $OUT/${TAG}_bench_under_rocprofv3.json 2> $OUT/${TAG}_kt.err || exit 1
# 3. HBM traffic: one counter per pass, nothing else enabled
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o pf -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline \
  > /dev/null 2> $OUT/${TAG}_pf.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o pw -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline \
  > /dev/null 2> $OUT/${TAG}_pw.err || exit 1
cd $ROOT && python3 profiles/summarize.py $TAG $OUT/${TAG}_kt $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
cp "$(ls $OUT/${TAG}_kt/*/*_kernel_trace.csv $OUT/${TAG}_kt/*_kernel_trace.csv 2>/dev/null | head -1)" $OUT/${TAG}_kernel_trace.csv
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_summary.json profiles/${TAG}_summary.md $OUT/ 2>/dev/null
ls -la $OUT | tail -20
