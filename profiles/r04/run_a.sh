#!/bin/bash
# round 4, first GPU call: the new tests (RCCL with one rank, the compat face under threads), the full bench line with the new
# compat_face leg (baseline before any work on it), HBM traffic counters of the -s kernels.
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_distributed_gpu.py tests/test_rapmap_compat.py tests/test_native_io.py -m gpu -q -x > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
python bench.py --gpus 1 --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
bash profiles/r04/pmc_traffic_sel.sh $OUT/sel_traffic > $OUT/sel_traffic.txt 2>&1; cat $OUT/sel_traffic.txt | head -60
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
for k,v in d.get("other_configs",{}).items():
    if isinstance(v, dict): print(k[:44], v["value"], v["ms_per_step"], v["roofline"]["frac"], v["parity"])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("pcie_inclusive","end_to_end")})
print("compat_face", json.dumps(d.get("compat_face")))
PY
