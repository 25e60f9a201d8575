#!/bin/bash
# collector slabs overlapped (qm_read_kernel.inl): the whole gpu suite, then -s by read length with parity
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
for L in 100 150 250; do
  timeout 900 python bench.py --sel-aln --read-len $L --no-other-configs --no-side-legs --steps 3 --warmup 1 --cpu-seconds 8 2>$OUT/len$L.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL $L bp: %.2f M pairs/s %.1f ms' % (d['value'], d['ms_per_step']), d.get('parity'))"
done
