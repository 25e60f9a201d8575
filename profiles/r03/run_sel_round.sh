#!/bin/bash
# -s measurement round (GPU box): the -s GPU tests, the bench line with parity, kernel-trace stats at 2x100 and 2x150 bp
# usage: bash profiles/r03/run_sel_round.sh gpurun_out/<dir>
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "selective or ksw or sel" > $OUT/pytest_sel.log 2>&1; tail -2 $OUT/pytest_sel.log
timeout 900 python bench.py --sel-aln --no-other-configs --no-side-legs --steps 3 --warmup 1 > $OUT/bench_sel.json 2> $OUT/bench_sel.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats100 -o s -- python bench.py --sel-aln --no-other-configs --no-side-legs --no-cpu-baseline --steps 3 --warmup 1 > $OUT/stats100.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats150 -o s -- python bench.py --sel-aln --read-len 150 --no-other-configs --no-side-legs --no-cpu-baseline --steps 3 --warmup 1 > $OUT/stats150.log 2>&1
for d in stats100 stats150; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo "# $d"; [ -n "$f" ] && head -9 "$f" | cut -c1-160; tail -1 $OUT/$d.log | cut -c1-330; done | tee $OUT/summary.txt
python - <<PY
import json
d=json.loads(open("$OUT/bench_sel.json").read().strip().splitlines()[-1])
print("bench -s:", d["value"], d["ms_per_step"], d.get("parity"))
PY
