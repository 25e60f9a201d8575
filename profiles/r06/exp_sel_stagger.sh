#!/bin/bash
# round 6: -s, the parts' stage A one after the other (QM_SPLIT_STAGGER) so that the scalar-bound collector of a part runs beside the VALU-bound
# list / plan / alignment kernels of the part before it; usage on the GPU box: bash profiles/r06/exp_sel_stagger.sh
run() { echo -n "$* : "; env "$@" python bench.py --sel-aln --no-cpu-baseline --no-other-configs --no-side-legs --no-input-variants --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))"; }
run QM_SPLIT=2
run QM_SPLIT=2 QM_SPLIT_STAGGER=1
run QM_SPLIT=4
run QM_SPLIT=4 QM_SPLIT_STAGGER=1
run QM_SPLIT=8 QM_SPLIT_STAGGER=1
run QM_SPLIT=3 QM_SPLIT_STAGGER=1
