#!/bin/bash
# oversubscribed alignment grids: -s with QM_ALIGN_OVERSUB 1 / 2 / 4 / 8 (with and without parts), then the full gpu suite
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, env, flags
  env $2 timeout 600 python bench.py $3 --no-cpu-baseline --no-other-configs --no-side-legs --steps 10 --warmup 3 2>$OUT/$1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))"
}
for a in 1 2 4 8; do run sel_align_over$a "QM_ALIGN_OVERSUB=$a" "--sel-aln"; done
for a in 1 4; do run sel_split1_align_over$a "QM_SPLIT=1 QM_ALIGN_OVERSUB=$a" "--sel-aln"; done
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
