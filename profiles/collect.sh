#!/bin/bash
# Collect the judged artifact set on the GPU box (one gpurun call):
#   bash profiles/collect.sh r4 [a|b]      (a: profiler passes + bench line, b: regime counters, timelines, regime table, CPU legs;
#                                           nothing = both -- about 20 minutes, more than one gpurun call may take)
# writes gpurun_out/<tag>_* and profiles/<tag>_{kernel_stats.csv,summary.json,summary.md}; the other files are copied
# into profiles/ by hand afterwards (the raw kernel trace is large).
set -o pipefail
TAG=${1:-r4}
PART=${2:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [[ $PART == *a* ]]; then
# 2. kernel trace + stats of the same command (its own JSON line is kept: the profiled process runs slower)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python3 $ROOT/bench.py --no-cpu-baseline \
  > $OUT/${TAG}_bench_under_rocprofv3.json 2> $OUT/${TAG}_kt.err || exit 1
echo "kernel trace done"
# 3. HBM traffic: one counter per pass, nothing else enabled
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o pf -- python3 $ROOT/bench.py --steps 5 --warmup 1 --steady-steps 0 --order cold --no-cpu-baseline \
  > /dev/null 2> $OUT/${TAG}_pf.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o pw -- python3 $ROOT/bench.py --steps 5 --warmup 1 --steady-steps 0 --order cold --no-cpu-baseline \
  > /dev/null 2> $OUT/${TAG}_pw.err || exit 1
echo "pmc passes done"
cd $ROOT && python3 profiles/summarize.py $TAG $OUT/${TAG}_kt $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
cp "$(ls $OUT/${TAG}_kt/*/*_kernel_trace.csv $OUT/${TAG}_kt/*_kernel_trace.csv 2>/dev/null | head -1)" $OUT/${TAG}_kernel_trace.csv
# 4. the other regimes under the same profiler: Infinity-Cache resident (12 envs) and the reference's small environments
cd /tmp
SUMMARY_WORKLOAD="bench.py --envs 12 (N=1e6, Ng=256, fp64: particles resident in the 256 MB Infinity Cache)" \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt_env12 -o kt -- python3 $ROOT/bench.py --steps 100 --warmup 10 --envs 12 --steady-steps 200 --no-cpu-baseline \
  > $OUT/${TAG}_env12_under_rocprofv3.json 2> $OUT/${TAG}_kt_env12.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt_small -o kt -- python3 $ROOT/bench.py --steps 200 --warmup 20 --envs 256 --particles 5000 --mesh 250 --steady-steps 400 --no-cpu-baseline \
  > $OUT/${TAG}_small256_under_rocprofv3.json 2> $OUT/${TAG}_kt_small.err || exit 1
for d in env12 small; do cp "$(ls $OUT/${TAG}_kt_$d/*/*_kernel_stats.csv $OUT/${TAG}_kt_$d/*_kernel_stats.csv 2>/dev/null | head -1)" $OUT/${TAG}_kernel_stats_$d.csv; done
echo "regime traces done"
cd $ROOT
# 4b. resident schedule: VALU wave-instructions per particle-step (bench.py prices its roofline with them), float64 and fixed32
cd /tmp
for fmt in "float64 float" "float32 fixed32"; do set -- $fmt
  for hist in "" "--history"; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/${TAG}_pmc_res_$1$hist -o pr -- python3 $ROOT/bench.py --no-cpu-baseline \
    --steady-steps 0 --profile-steps 0 --order cold --envs 256 --particles 5000 --mesh 250 --steps 100 --warmup 100 --dtype $1 --positions $2 $hist > /dev/null 2> $OUT/${TAG}_pmc_res_$1$hist.err \
    && python3 $ROOT/profiles/resident_valu.py $TAG $OUT/${TAG}_pmc_res_$1$hist "$1/$2/N=5000/Ng=250${hist:+/history}" 5000 256 100
  done
done
echo "resident pmc done"
cd $ROOT
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_summary.json profiles/${TAG}_summary.md $OUT/ 2>/dev/null
# 1. unprofiled bench line (with the 1-core CPU baseline leg); run after the counter passes so that `roofline.traffic` quotes this
#    collection's own summary
python3 $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
echo "bench done"
fi
if [[ $PART == *b* ]]; then
cd $ROOT
# 4c. SQ counters of the few-large-environment regime (one environment of N=1e6) and its in-kernel timeline
bash profiles/pmc_regime.sh gpurun_out/pmc_${TAG} env1 --envs 1 --steps 100 --steady-steps 0 > $OUT/${TAG}_pmc_env1.log 2>&1
cp gpurun_out/pmc_${TAG}/env1_pmc.md $OUT/${TAG}_pmc_env1.md 2>/dev/null
python3 profiles/timeline/timeline.py --envs 1 --steps 3 > $OUT/${TAG}_timeline_one_1e6.md 2>/dev/null
python3 profiles/timeline/timeline.py --envs 4 --steps 3 > $OUT/${TAG}_timeline_four_1e6.md 2>/dev/null
python3 profiles/timeline/timeline.py --envs 1 --particles 5000 --mesh 250 --steps 1 --calls 4 > $OUT/${TAG}_timeline_res_single.md 2>/dev/null
python3 profiles/timeline/timeline.py --envs 1 --particles 5000 --mesh 250 --steps 10 --calls 2 > $OUT/${TAG}_timeline_res_ten.md 2>/dev/null
echo "regime counters and timelines done"
# 5. the regime table: unprofiled bench lines (HIP-event kernel times) at every BASELINE configuration share and regime
python3 profiles/regimes.py $OUT/${TAG}_regimes > $OUT/${TAG}_regimes.md 2> $OUT/${TAG}_regimes.err
# 6. CPU comparators of SURVEY 8d: (a) one core at config 1, (b) one process per core at config 2
python3 bench.py --steps 200 --warmup 20 --envs 1 --particles 10000 --mesh 128 > $OUT/${TAG}_bench_config1.json 2> /dev/null
python3 bench.py --cpu-procs 16 > $OUT/${TAG}_bench_cpu16.json 2> /dev/null
python3 profiles/pyloop_rl.py > $OUT/${TAG}_pyloop_rl.log 2> /dev/null
python3 profiles/gym_breakdown.py > $OUT/${TAG}_gym_breakdown.log 2> /dev/null
fi
ls -la $OUT | tail -30
