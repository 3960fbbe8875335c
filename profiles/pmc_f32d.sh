#!/bin/bash
# SQ counters of float32 / float-position sweep D at config 3's shape in the round-2 tree and in the working tree (profiles/mk_ab.sh
# r2 783b15d / head WORK): same instruction mix in the loop, 353 against 366-372 us on the same box -- where do the cycles go?
#   bash profiles/pmc_f32d.sh <outdir>
out=$1; here=$(pwd); mkdir -p $out
shape="--steps 6 --envs 128 --mesh 512 --dtype float32 --positions float --init two-stream"
for t in r2 head; do for ctl in free act; do
  extra=""; [ $t = head ] && extra="--steady-steps 0 --order cold"
  [ $ctl = act ] && extra="$extra --actions 3"
  (cd profiles/ab/$t && GRAFT_REPO_ROOT=$(pwd) bash $here/profiles/pmc_regime.sh pmc_out ${t}_$ctl $shape $extra) > $out/pmc_${t}_$ctl.log 2>&1
  cp profiles/ab/$t/pmc_out/${t}_${ctl}_pmc.md $out/ 2>/dev/null
done; done
tail -n +1 $out/*_pmc.md
