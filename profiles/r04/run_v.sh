#!/bin/bash
# -s list kernel: occupancy variants (QM_SEL_SMALL records in the LDS edition of the scratch, waves per SIMD it is built for)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-other-configs --no-side-legs --no-cpu-baseline --steps 5 --warmup 2 --sel-aln"
for v in base 48_5 48_4 32_6 base; do
  if [ $v = base ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/profiles/variants/libqmap_h2m_$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$v -o s -- python bench.py $B > $OUT/sel_$v.log 2>&1
  f=$(find $OUT/st_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(tail -1 $OUT/sel_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  grep "qm_h2m_kernel\|qm_read_kernel" $f | cut -d, -f1,2,4 | cut -c1-120
done
unset QM_LIB_OVERRIDE
for L in 150 250; do for v in base 48_5; do
  if [ $v = base ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/profiles/variants/libqmap_h2m_$v.so; fi
  timeout 600 python bench.py $B --read-len $L --pairs 4000000 > $OUT/sel_${v}_L$L.json 2> $OUT/err.log
  echo "== $v 2x$L: $(tail -1 $OUT/sel_${v}_L$L.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))")"
done; done
