#!/bin/bash
# -s on longer reads: grid oversubscription of the stage-A / alignment kernels (defaults 4 / 4)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 250 150; do for G in 2 4 8; do for A in 4 8; do
  QM_GRID_OVERSUB=$G QM_ALIGN_OVERSUB=$A timeout 600 python bench.py --sel-aln --read-len $L --no-cpu-baseline --no-other-configs --no-side-legs --steps 3 --warmup 1 2>$OUT/e.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL $L bp grid-oversub $G align-oversub $A: %.2f M pairs/s %.1f ms' % (d['value'], d['ms_per_step']))"
done; done; done
