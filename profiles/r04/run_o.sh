#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_native_io.py tests/test_cli.py -m gpu -q -x > $OUT/pytest_io.log 2>&1; tail -3 $OUT/pytest_io.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -4 $OUT/bench_default.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["parity"])
for k,v in d.get("other_configs",{}).items():
    if isinstance(v, dict): print(k[:44], v["value"], v["ms_per_step"], v["kernel_ms"], v["roofline"]["frac"], v["roofline"].get("traffic"), v["parity"])
print("pcie", d.get("pcie_inclusive",{}).get("value"))
print("e2e", json.dumps(d.get("end_to_end"))[:700])
print("compat_face", json.dumps(d.get("compat_face", {}).get("by_host_threads")))
PY
