#!/bin/bash
# oversubscribed stage-A grids as the default (QM_GRID_OVERSUB=4), parts only with -s: full gpu suite, then the three head lines and a few alternatives
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
run() { # name, env, flags
  env $2 timeout 600 python bench.py $3 --no-cpu-baseline --no-other-configs --no-side-legs --steps 10 --warmup 3 2>$OUT/$1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))"
}
run dense_default "X=1" ""
run dense_over6 "QM_GRID_OVERSUB=6" ""
run sel_default "X=1" "--sel-aln"
run sel_split2_over2 "QM_SPLIT=2 QM_GRID_OVERSUB=2" "--sel-aln"
run sel_split1 "QM_SPLIT=1" "--sel-aln"
run sel_split3 "QM_SPLIT=3" "--sel-aln"
run ph_default "X=1" "--perfect-hash --ph-compact"
run ph_split2 "QM_SPLIT=2" "--perfect-hash --ph-compact"
run ph_split3_over2 "QM_SPLIT=3 QM_GRID_OVERSUB=2" "--perfect-hash --ph-compact"
run phx_default "X=1" "--perfect-hash"
