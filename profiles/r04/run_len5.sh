#!/bin/bash
# list kernels after a change: -s GPU tests, -s by read length with parity against the oracle, then the wide kernel's phase timers (timing.so)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "sel or stage or long or device_resident or longer" > $OUT/pytest_sel.log 2>&1; tail -2 $OUT/pytest_sel.log
for L in 100 150 250; do
  timeout 900 python bench.py --sel-aln --read-len $L --no-other-configs --no-side-legs --steps 3 --warmup 1 --cpu-seconds 8 2>$OUT/len$L.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL $L bp: %.2f M pairs/s %.1f ms' % (d['value'], d['ms_per_step']), d.get('parity'))"
done
for L in 250 100; do
QM_SPLIT=1 QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/timing.so timeout 600 python bench.py --sel-aln --read-len $L --no-cpu-baseline --no-other-configs --no-side-legs --steps 1 --warmup 1 > $OUT/timing$L.json 2> $OUT/timing$L.err
echo "== $L bp"; grep "qm timing pack" $OUT/timing$L.err | tail -8
done
