#!/bin/bash
# batches in parts (map_device_split): device-resident tests, then the default and -s bench lines with QM_SPLIT unset / 1 / other counts
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "device_resident or list_kernels" > $OUT/pytest_split.log 2>&1; tail -5 $OUT/pytest_split.log
run() { # name, env, flags
  env $2 timeout 600 python bench.py $3 --no-cpu-baseline --no-other-configs --no-side-legs --steps 10 --warmup 3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$1.json").read().strip().splitlines()[-1])
print("$1", d["value"], d["ms_per_step"], d["config"].get("map_kernel_ms"))
PY
}
run dense_default "X=1" ""
run dense_split1 "QM_SPLIT=1" ""
run dense_split2 "QM_SPLIT=2" ""
run dense_split4 "QM_SPLIT=4" ""
run sel_default "X=1" "--sel-aln"
run sel_split1 "QM_SPLIT=1" "--sel-aln"
run sel_split3 "QM_SPLIT=3" "--sel-aln"
run ph_default "X=1" "--perfect-hash --ph-compact"
run ph_split1 "QM_SPLIT=1" "--perfect-hash --ph-compact"
