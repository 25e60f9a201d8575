cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-other-configs --no-side-legs --no-input-variants --steps 10 --warmup 2"
run() { echo "== $*"; env "$@" python bench.py $A 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],'M pairs/s',d['ms_per_step'],'ms/step kernel',d['config'].get('map_kernel_ms'))"; }
run QM_DUO_PARTS=-1
run QM_DUO_PARTS=1
run QM_DUO_PARTS=0
run QM_DUO_PARTS=-1 QM_SPLIT=1
