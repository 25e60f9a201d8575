#!/bin/bash
# A/B of a stage-A change: the in-tree library against profiles/variants/libqmap_base.so (the previous source), alternating, same box
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-other-configs --no-side-legs --no-cpu-baseline --steps 10 --warmup 2"
line() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))" $1 "$2"; }
for r in 1 2; do
  timeout 600 python bench.py $B > $OUT/dense_new$r.json 2> $OUT/err.log; line $OUT/dense_new$r.json "dense new"
  QM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/profiles/variants/libqmap_base.so timeout 600 python bench.py $B > $OUT/dense_base$r.json 2> $OUT/err.log; line $OUT/dense_base$r.json "dense base"
done
timeout 600 python bench.py $B --sel-aln > $OUT/sel_new.json 2> $OUT/err.log; line $OUT/sel_new.json "sel new"
QM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/profiles/variants/libqmap_base.so timeout 600 python bench.py $B --sel-aln > $OUT/sel_base.json 2> $OUT/err.log; line $OUT/sel_base.json "sel base"
timeout 600 python bench.py $B --perfect-hash --ph-compact > $OUT/ph_new.json 2> $OUT/err.log; line $OUT/ph_new.json "ph new"
QM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/profiles/variants/libqmap_base.so timeout 600 python bench.py $B --perfect-hash --ph-compact > $OUT/ph_base.json 2> $OUT/err.log; line $OUT/ph_base.json "ph base"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; tail -2 $OUT/pytest_parity.log
