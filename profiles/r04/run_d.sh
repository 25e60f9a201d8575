#!/bin/bash
# round 4: first-probe prefetch (default kernel), staged sanext entries (-s collector), sharded HitCounters (compat face)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rapmap_compat.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; tail -4 $OUT/pytest_parity.log
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["parity"])
for k,v in d.get("other_configs",{}).items():
    if isinstance(v, dict): print(k[:44], v["value"], v["ms_per_step"], v["kernel_ms"], v["roofline"]["frac"], v["parity"])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("pcie_inclusive","end_to_end")})
print("compat_face", json.dumps(d.get("compat_face", {}).get("by_host_threads")))
PY
timeout 600 python profiles/r04/compat_probe.py 2000000 1,8,32,64 > $OUT/compat_probe.txt 2>&1; tail -5 $OUT/compat_probe.txt
