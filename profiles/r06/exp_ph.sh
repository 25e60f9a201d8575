cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-other-configs --no-side-legs --steps 6 --warmup 2"
run() { echo "== $*"; env "$@" python bench.py $A --perfect-hash --ph-compact 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],'M pairs/s',d['ms_per_step'],'ms/step kernel',d['config'].get('map_kernel_ms'),'parity',d.get('parity',{}).get('bit_identical_to_oracle'))"; }
run QM_DUO_PARTS=0
run QM_DUO_PARTS=-1
run QM_DUO_PARTS=1
