#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python profiles/r04/compat_probe.py 2000000 1,4,8,16,32 > $OUT/compat_probe.txt 2>&1; tail -6 $OUT/compat_probe.txt
QM_BLOCKING_SYNC=1 python profiles/r04/compat_probe.py 2000000 8,32 > $OUT/compat_probe_blocking.txt 2>&1; tail -3 $OUT/compat_probe_blocking.txt
GPU_MAX_HW_QUEUES=8 python profiles/r04/compat_probe.py 2000000 8,32 > $OUT/compat_probe_q8.txt 2>&1; tail -3 $OUT/compat_probe_q8.txt
nproc; lscpu | grep -i "numa\|socket\|model name" | head
