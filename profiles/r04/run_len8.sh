#!/bin/bash
# 2 x 150 bp -s: the three-slot collector built for 8 waves per SIMD (rapmap_amd/variants/ns3w8.so) against the tree's (6 by its registers, 7 resident)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in tree ns3w8 tree ns3w8; do
  if [ $V = tree ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/$V.so; fi
  timeout 900 python bench.py --sel-aln --read-len 150 --no-other-configs --no-side-legs --steps 3 --warmup 1 --cpu-seconds 6 2>$OUT/$V.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SEL 150 bp $V: %.2f M pairs/s %.1f ms kernel %s' % (d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms')), d.get('parity'))"
done
