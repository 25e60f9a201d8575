#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python profiles/r04/ingest_hyp.py after > $OUT/ingest_after2.log 2> $OUT/ingest_after.err; cat $OUT/ingest_after2.log; tail -3 $OUT/ingest_after.err
