#!/bin/bash
# -s parts: stage A one part after the other (QM_SPLIT_STAGGER) against all parts at once, 2 / 3 / 4 parts
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, env, flags
  env $2 timeout 600 python bench.py $3 --no-cpu-baseline --no-other-configs --no-side-legs --steps 10 --warmup 3 2>$OUT/$1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))"
}
for k in 2 3 4 6; do for g in 0 1; do run sel_parts${k}_stagger$g "QM_SPLIT=$k QM_SPLIT_STAGGER=$g" "--sel-aln"; done; done
timeout 900 python -m pytest tests -m gpu -q -k "device_resident or sel" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
