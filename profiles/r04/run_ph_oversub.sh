#!/bin/bash
# compact -p kernel <2,6,1>: grid oversubscription sweep
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for G in 4 2 6 8 12 16 4; do
  QM_GRID_OVERSUB=$G timeout 600 python bench.py --perfect-hash --ph-compact --no-cpu-baseline --no-other-configs --no-side-legs --steps 10 --warmup 2 2>$OUT/e.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('compact -p oversub $G: %.2f M pairs/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
