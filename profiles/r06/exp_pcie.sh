#!/bin/bash
# round 6: qm_map_pairs on pageable host buffers with (default) / without (QM_HOST_STAGE=0) the staged parallel upload; usage on the GPU box: bash profiles/r06/exp_pcie.sh
run() { echo -n "$* : "; env "$@" python bench.py --no-cpu-baseline --no-other-configs --no-input-variants --e2e-copies 1 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], (d.get('pcie_inclusive') or {}).get('value'), (d.get('end_to_end') or {}).get('value'))"; }
run QM_HOST_STAGE=0
run QM_HOST_STAGE=1
run QM_HOST_STAGE=0
run QM_HOST_STAGE=1
