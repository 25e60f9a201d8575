#!/bin/bash
# phase timers of the packed list kernel (-DQM_TIMING build as rapmap_amd/variants/timing.so), then PMC passes of -s
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/timing.so timeout 600 python bench.py --sel-aln --no-cpu-baseline --no-other-configs --no-side-legs --steps 1 --warmup 1 > $OUT/timing.json 2> $OUT/timing.err
grep "qm timing" $OUT/timing.err | tail -30
bash profiles/r04/run_profile.sh $OUT/prof_sel --sel-aln > /dev/null 2>&1
grep -A22 "qm_h2m_pack" $OUT/prof_sel/summary.txt | head -30
