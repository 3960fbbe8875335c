#!/bin/bash
# Export a tree (a git revision, or WORK = the working tree) with a freshly built library into profiles/ab/<name>,
# for same-box A/B runs through `cd profiles/ab/<name> && python bench.py ...`.  profiles/ab/ is git-ignored.
# usage: bash profiles/mk_ab.sh <name> <rev|WORK>
set -e
name=$1; rev=$2
dst=profiles/ab/$name
rm -rf $dst; mkdir -p $dst
if [ "$rev" = WORK ]; then
  tar -c --exclude='*.so' --exclude=__pycache__ bench.py ocplasma_amd.py optimal-control-1d-electrostatic-plasma_amd oracle include | tar -x -C $dst
else
  git archive $rev bench.py ocplasma_amd.py optimal-control-1d-electrostatic-plasma_amd oracle include | tar -x -C $dst
fi
(cd $dst && python -c "import ocplasma_amd; from ocplasma_amd import _build; print(_build.build_library(force=True))")
