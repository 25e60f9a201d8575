cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-other-configs --no-side-legs --steps 10 --warmup 2"
run() { echo "== $*"; env "$@" python bench.py $A 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],'M pairs/s',d['ms_per_step'],'ms/step kernel',d['roofline'].get('kernel_ms'),'parity',d.get('parity',{}).get('bit_identical_to_oracle'))"; }
run QM_DUO_PARTS=-1
run QM_DUO_PARTS=1
run QM_DUO_PARTS=1 QM_SPLIT_FIRST=40
run QM_DUO_PARTS=1 QM_SPLIT_FIRST=60
run QM_DUO_PARTS=0
run QM_DUO_PARTS=5 QM_SPLIT=4
run QM_DUO_PARTS=1 QM_SPLIT=3
run QM_DUO_PARTS=3 QM_SPLIT=3
run QM_DUO_PARTS=1 QM_SPLIT=2
