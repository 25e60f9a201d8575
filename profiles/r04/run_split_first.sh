#!/bin/bash
# -s, two parts in flight of unequal size (QM_SPLIT_FIRST = per cent of the batch in the first part; 0 / unset: halves)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for F in 0 30 40 45 55 60 70 0; do
  QM_SPLIT_FIRST=$F timeout 600 python bench.py --sel-aln --no-cpu-baseline --no-other-configs --no-side-legs --steps 5 --warmup 1 2>$OUT/e.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL 100 bp first part $F %%: %.2f M pairs/s %.1f ms' % (d['value'], d['ms_per_step']))"
done
