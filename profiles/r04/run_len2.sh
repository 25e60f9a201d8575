#!/bin/bash
# wide packed list kernel: -s tests, then -s by read length (with parity) and the kernel times of the 2 x 250 bp step
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "sel or stage or long or device_resident or longer" > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
for L in 100 150 250; do
  timeout 900 python bench.py --sel-aln --read-len $L --no-other-configs --no-side-legs --steps 3 --warmup 1 --cpu-seconds 8 2>$OUT/len$L.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL $L bp: %.2f M pairs/s %.1f ms' % (d['value'], d['ms_per_step']), d.get('parity'))"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats250 -o s -- python bench.py --sel-aln --read-len 250 --no-cpu-baseline --no-other-configs --no-side-legs --steps 2 --warmup 1 > $OUT/stats250.log 2>&1
f=$(find $OUT/stats250 -name "*kernel_stats.csv" | head -1); grep "qm::" "$f" | grep -v "build_\|rocprim" | sed 's/"\(void \)\{0,1\}qm::\([a-z_0-9A-Z<>, ]*\).*",\([0-9]*\),\([0-9]*\),\([0-9.]*\),.*/\2 calls \3 total_ns \4 avg_ns \5/' | head -12
