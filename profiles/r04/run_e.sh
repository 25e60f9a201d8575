#!/bin/bash
# round 4: the batching service behind the compat header; A/B of the first-probe prefetch (QM_TUNE=1 off) and of the staged sanext
# entries (QM_TUNE=2 off); BooPHF levels per round (QM_PH_SPEC 1 / 2 / 3) on the compact -p image
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rapmap_compat.py tests/test_compat_header.py -m gpu -q -x > $OUT/pytest_compat.log 2>&1; tail -4 $OUT/pytest_compat.log
timeout 600 python profiles/r04/compat_probe.py 4000000 1,8,32,64 > $OUT/compat_probe.txt 2>&1; tail -5 $OUT/compat_probe.txt
B="--no-other-configs --no-side-legs --no-cpu-baseline --steps 10 --warmup 2"
line() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))" $1 "$2"; }
for t in 0 1; do QM_TUNE=$t timeout 600 python bench.py $B > $OUT/dense_tune$t.json 2> $OUT/dense_tune$t.err; line $OUT/dense_tune$t.json "dense QM_TUNE=$t"; done
for t in 0 1 2 3; do QM_TUNE=$t timeout 600 python bench.py $B --sel-aln > $OUT/sel_tune$t.json 2> $OUT/sel_tune$t.err; line $OUT/sel_tune$t.json "sel QM_TUNE=$t"; done
timeout 600 python bench.py $B --perfect-hash --ph-compact > $OUT/ph_spec3.json 2> $OUT/ph_spec3.err; line $OUT/ph_spec3.json "ph compact SPEC=3"
for n in 1 2; do QM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/profiles/variants/libqmap_spec$n.so timeout 600 python bench.py $B --perfect-hash --ph-compact > $OUT/ph_spec$n.json 2> $OUT/ph_spec$n.err; line $OUT/ph_spec$n.json "ph compact SPEC=$n"; done
