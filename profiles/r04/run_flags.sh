#!/bin/bash
# code-generation switches on the two-slot stage-A unit (qm_kernels_ns2: default, -s collector, compact -p): variants under rapmap_amd/variants against the tree
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in tree ns2_ilp ns2_trk ns2_nosink tree ns2_ilp ns2_trk; do
  if [ $V = tree ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/$V.so; fi
  timeout 900 python bench.py --no-side-legs --no-cpu-baseline --steps 10 --warmup 2 2>$OUT/$V.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
o = d.get('other_configs', {})
print('$V: default %.1f (kernel %.2f ms)' % (d['value'], d['roofline']['kernel_ms']), ' '.join('%s %.1f' % (k.split()[1][:3] + k.split()[-1][:9], v['value']) for k, v in o.items() if isinstance(v, dict)))"
done
