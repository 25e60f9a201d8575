#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python profiles/r04/ingest_hyp.py > $OUT/ingest_hyp.log 2> $OUT/ingest_hyp.err; cat $OUT/ingest_hyp.log; tail -3 $OUT/ingest_hyp.err; which perf
