#!/bin/bash
# round 4 evidence run: full gpu suite, smoke, profiles of the default bench and of -s (kernel-trace stats + PMC passes)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash profiles/r04/run_profile.sh $OUT/prof_dense > /dev/null 2>&1; head -14 $OUT/prof_dense/summary.txt | cut -c1-200
bash profiles/r04/run_profile.sh $OUT/prof_sel --sel-aln > /dev/null 2>&1; head -16 $OUT/prof_sel/summary.txt | cut -c1-200
