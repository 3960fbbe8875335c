#!/bin/bash
# SQ counters of the sweeps for one bench.py workload, one rocprofv3 --pmc pass per counter group.
# Round 2 also tried a TA_* group: rocprofiler_create_counter_config failed with error 38 ("Request exceeds the capabilities of the
# hardware to collect": more counters of one block than it has slots), rocprofv3 raised SIGABRT at the first dispatch and the
# process then sat silent until gpurun's 7-minute watchdog killed it (gpurun_out/pmc/env12_g3.err) -- an over-subscribed group, not a
# hang of the pool.  Groups below stay within 8 SQ counters per pass; every pass runs under its own `timeout`, so an abort cannot
# sit silent.
#   bash profiles/pmc_regime.sh <outdir> <tag> [bench args...]      e.g.  ... gpurun_out/pmc env12 --envs 12 --steps 20
out=$1; tag=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_SALU"
)
g=0
for grp in "${groups[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$out/${tag}_g$g -o p -- python3 $ROOT/bench.py --no-cpu-baseline --warmup 2 --profile-steps 0 "$@" > /dev/null 2> $ROOT/$out/${tag}_g$g.err || echo "group $g failed: $(tail -2 $ROOT/$out/${tag}_g$g.err)"
  g=$((g+1))
done
cd $ROOT && python3 - $out $tag <<'PY'
import collections, csv, glob, os, sys
out, tag = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, f"{tag}_g*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "sweep_kernel<" in n:
            k = "sweep_" + ("A", "B", "C", "D", "R", "P", "B2", "D2")[int(n.split("sweep_kernel<")[1].split(">")[0].split(",")[3])]
        elif "resident_kernel" in n: k = "resident"
        elif "stream_probe" in n: k = "stream_probe"
        else: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, f"{tag}_pmc.md"), "w") as fo:
    names = sorted({c for k in agg for c in agg[k]})
    ks = sorted(agg)
    fo.write(f"# PMC counters per launch (mean), {tag}\n\n| counter | " + " | ".join(ks) + " |\n|---|" + "---|" * len(ks) + "\n")
    for c in names:
        fo.write(f"| {c} | " + " | ".join(f"{sum(agg[k][c]) / len(agg[k][c]):.4g}" if agg[k][c] else "-" for k in ks) + " |\n")
print(open(os.path.join(out, f"{tag}_pmc.md")).read())
PY
