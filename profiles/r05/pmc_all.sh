#!/bin/bash
# round 5: the counter passes behind profiles/pmc_traffic.json -- for the default leg, the compact -p leg and the -s leg (unsplit, so
# that a dispatch is 10 M pairs, and with the plan kernels on the alignment kernels' stream, QM_SEL_SERIAL, so that every kernel's duration is its own): kernel-trace stats + FETCH_SIZE / WRITE_SIZE / TCC (separate --pmc passes) + the SQ instruction
# counters.  usage on the GPU box: bash profiles/r05/pmc_all.sh gpurun_out/<dir>; then python profiles/r05/make_pmc_traffic.py <dir> <tag>
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (QM_SPLIT=1: one launch per step, a dispatch is the whole batch of 10 M pairs; the bench itself maps a batch as two parts in flight)
QM_SPLIT=1 bash profiles/r05/run_profile.sh $OUT/dense > /dev/null 2>&1
QM_SPLIT=1 bash profiles/r05/run_profile.sh $OUT/ph_compact --perfect-hash --ph-compact > /dev/null 2>&1
QM_SPLIT=1 QM_SEL_SERIAL=1 bash profiles/r05/run_profile.sh $OUT/sel --sel-aln > /dev/null 2>&1
for d in dense ph_compact sel; do echo "== $d"; grep -A4 "^\"Name\"" $OUT/$d/summary.txt | cut -c1-140; done
# the JSON and the summaries that are committed, made here (the per-dispatch csv files are too large to travel back)
mkdir -p $OUT/commit && python profiles/r05/make_pmc_traffic.py $OUT ${2:-r05} $OUT/commit
for d in dense ph_compact sel; do find $OUT/$d -name "*.csv" -delete; done
