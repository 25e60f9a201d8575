#!/bin/bash
# round 4, second GPU call: qm_fetch_stages / the compat face on it, the strided probe of the -s collector, the bench line with compat_face
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_rapmap_compat.py tests/test_compat_header.py tests/test_abi.py -m gpu -q -x > $OUT/pytest_compat.log 2>&1; tail -5 $OUT/pytest_compat.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stage_entry or selective or sel or chain" > $OUT/pytest_sel.log 2>&1; tail -5 $OUT/pytest_sel.log
python bench.py --gpus 1 --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
for k,v in d.get("other_configs",{}).items():
    if isinstance(v, dict): print(k[:44], v["value"], v["ms_per_step"], v["kernel_ms"], v["roofline"]["frac"], v["parity"])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("pcie_inclusive","end_to_end")})
print("compat_face", json.dumps(d.get("compat_face")))
PY
