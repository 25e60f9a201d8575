#!/bin/bash
# round 4: batching service with the linger policy; default / -p compact (SPEC 2) after the revert of the prefetch experiments
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rapmap_compat.py -m gpu -q -x > $OUT/pytest_compat.log 2>&1; tail -3 $OUT/pytest_compat.log
timeout 600 python profiles/r04/compat_probe.py 4000000 1,8,16,32,64 > $OUT/compat_probe.txt 2>&1; tail -6 $OUT/compat_probe.txt
QMAP_COMPAT_LINGER_US=1000 timeout 600 python profiles/r04/compat_probe.py 4000000 32 > $OUT/compat_probe_l1000.txt 2>&1; tail -1 $OUT/compat_probe_l1000.txt
QMAP_COMPAT_CONTEXTS=4 timeout 600 python profiles/r04/compat_probe.py 4000000 32 > $OUT/compat_probe_c4.txt 2>&1; tail -1 $OUT/compat_probe_c4.txt
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
