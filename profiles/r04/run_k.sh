#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python profiles/r04/ingest_numa.py 20000000 > $OUT/ingest_numa.log 2> $OUT/ingest_numa.err; grep -v amdgpu $OUT/ingest_numa.log; tail -8 $OUT/ingest_numa.err
