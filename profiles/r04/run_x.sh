#!/bin/bash
# the packed list kernel of -s (qm_selpack.inl) on the GPU: parity tests that go through -s, then the -s bench with and without it
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "sel or stage or long or compat or golden" > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
for pk in 1 0; do
  QM_SEL_PACK=$pk timeout 600 python bench.py --sel-aln --no-other-configs --no-side-legs --steps 5 --warmup 2 > $OUT/bench_sel_pack$pk.json 2> $OUT/bench_sel_pack$pk.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_sel_pack$pk.json").read().strip().splitlines()[-1])
print("pack=$pk", d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["parity"])
PY
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --sel-aln --no-cpu-baseline --no-other-configs --no-side-legs --steps 3 --warmup 1 > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
