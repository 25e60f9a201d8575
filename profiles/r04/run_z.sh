#!/bin/bash
# packed list kernel, second edition (register chaining for one-diagonal groups): -s parity tests, then -s bench with the tree's
# build and with variants (rapmap_amd/variants/*.so: other occupancy targets)
set -u
OUT=$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "sel or stage or long" > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
for v in main "$@"; do
  if [ "$v" = main ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/$v.so; fi
  timeout 600 python bench.py --sel-aln --no-cpu-baseline --no-other-configs --no-side-legs --steps 5 --warmup 2 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["parity"])
PY
done
unset QM_LIB_OVERRIDE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --sel-aln --no-cpu-baseline --no-other-configs --no-side-legs --steps 3 --warmup 1 > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); grep "h2m\|Name" "$f" | cut -c1-160
