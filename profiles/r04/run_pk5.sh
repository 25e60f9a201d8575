#!/bin/bash
# narrow list kernel built for 5 waves per SIMD (96 VGPRs, no spills: rapmap_amd/variants/pk5.so) against the tree's 6 (80 VGPRs, 9 spilled): 2 x 100 bp -s
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in tree pk5 tree pk5; do
  if [ $V = tree ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/$V.so; fi
  timeout 900 python bench.py --sel-aln --no-other-configs --no-side-legs --no-cpu-baseline --steps 5 --warmup 1 2>$OUT/$V.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL 100 bp $V: %.2f M pairs/s %.1f ms' % (d['value'], d['ms_per_step']))"
done
