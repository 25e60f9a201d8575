#!/bin/bash
# end-of-round evidence run (GPU box): full gpu test suite, smoke, bench lines of configs 2 / 4 / 5, kernel-trace stats of config 5
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py > $OUT/bench_c2.log 2>&1
python bench.py --perfect-hash > $OUT/bench_c4.log 2>&1
python bench.py --sel-aln --steps 2 > $OUT/bench_c5.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sel_stats -o s -- python bench.py --sel-aln --steps 2 --no-cpu-baseline > $OUT/sel_stats.log 2>&1
f=$(find $OUT/sel_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-260 > $OUT/sel_kernel_stats.txt
tail -2 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; for c in c2 c4 c5; do tail -1 $OUT/bench_$c.log | cut -c1-180; done; cat $OUT/sel_kernel_stats.txt
