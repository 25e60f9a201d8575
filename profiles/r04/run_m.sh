#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python profiles/r04/ingest_hyp.py affinity > $OUT/ingest_affinity.log 2> $OUT/ingest_after.err; cat $OUT/ingest_affinity.log; tail -3 $OUT/ingest_after.err
