#!/bin/bash
# phase timers of the wide list kernel on 2 x 250 bp (a -DQM_TIMING build of qm_kernels / qm_host as rapmap_amd/variants/timing.so)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 250 150; do
QM_SPLIT=1 QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/timing.so timeout 600 python bench.py --sel-aln --read-len $L --no-cpu-baseline --no-other-configs --no-side-legs --steps 1 --warmup 1 > $OUT/timing$L.json 2> $OUT/timing$L.err
echo "== $L bp"; grep "qm timing pack\|qm timing h2m" $OUT/timing$L.err | tail -16
done
