#!/bin/bash
# round 4 validation: full gpu suite, smoke, the driver-style bench line (N=1, every leg)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
t0=$(date +%s); timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench.py (no flags) took $(( $(date +%s) - t0 )) s"; tail -3 $OUT/bench_default.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["parity"])
for k,v in d.get("other_configs",{}).items():
    if isinstance(v, dict): print(k[:44], v["value"], v["ms_per_step"], v["kernel_ms"], v["roofline"]["frac"], v["parity"])
print("pcie", d.get("pcie_inclusive",{}).get("value"))
e=d.get("end_to_end") or {}; print("e2e", e.get("value"), e.get("pairs"), e.get("without_read_names"))
print("compat_face", json.dumps({k:(v.get("value"), v.get("bit_identical_joint_hits")) for k,v in (d.get("compat_face", {}).get("by_host_threads") or {}).items()}))
PY
