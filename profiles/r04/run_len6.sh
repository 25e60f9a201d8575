#!/bin/bash
# 2 x 150 bp -s: narrow list kernel first (default) against the wide edition directly (QM_SEL_WIDE_FROM=128)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in 192 128; do
  QM_SEL_WIDE_FROM=$W timeout 900 python bench.py --sel-aln --read-len 150 --no-other-configs --no-side-legs --no-cpu-baseline --steps 3 --warmup 1 2>$OUT/w$W.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL 150 bp wide-from $W: %.2f M pairs/s %.1f ms' % (d['value'], d['ms_per_step']))"
done
QM_SEL_WIDE_FROM=128 QM_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats150 -o s -- python bench.py --sel-aln --read-len 150 --no-cpu-baseline --no-other-configs --no-side-legs --steps 2 --warmup 1 > $OUT/stats150.log 2>&1
f=$(find $OUT/stats150 -name "*kernel_stats.csv" | head -1); grep "qm::" "$f" | grep -v "build_\|rocprim" | sed 's/"\(void \)\{0,1\}qm::\([a-z_0-9A-Z<>, ]*\).*",\([0-9]*\),\([0-9]*\),\([0-9.]*\),.*/\2 calls \3 total_ns \4 avg_ns \5/' | head -8
