#!/bin/bash
# -s by read length: kernel times of the 2 x 250 bp and 2 x 150 bp steps (unsplit, so that one launch = the whole batch), then parts in flight A/B
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kst() { f=$(find $1 -name "*kernel_stats.csv" | head -1); grep "qm::" "$f" | grep -v "build_\|rocprim" | sed 's/"\(void \)\{0,1\}qm::\([a-z_0-9A-Z<>, ]*\).*",\([0-9]*\),\([0-9]*\),\([0-9.]*\),.*/\2 calls \3 total_ns \4 avg_ns \5/' | head -14; }
for L in 250 150; do
  QM_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$L -o s -- python bench.py --sel-aln --read-len $L --no-cpu-baseline --no-other-configs --no-side-legs --steps 2 --warmup 1 > $OUT/stats$L.log 2>&1
  echo "== $L bp, QM_SPLIT=1"; tail -1 $OUT/stats$L.log | cut -c1-200; kst $OUT/stats$L
done > $OUT/kernel_times.txt 2>&1
for L in 250 150; do for S in 1 2 3; do
  QM_SPLIT=$S timeout 600 python bench.py --sel-aln --read-len $L --no-cpu-baseline --no-other-configs --no-side-legs --steps 3 --warmup 1 2>$OUT/len${L}_$S.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL $L bp split $S: %.2f M pairs/s %.1f ms' % (d['value'], d['ms_per_step']))"
done; done > $OUT/split_ab.txt 2>&1
cat $OUT/kernel_times.txt $OUT/split_ab.txt
