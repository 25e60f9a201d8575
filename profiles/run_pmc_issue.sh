#!/bin/bash
# instruction-mix / issue counters of the mapping kernel (separate --pmc passes); usage: profiles/run_pmc_issue.sh <outdir> [bench flags]
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --steps 1 --warmup 0 $*"
pass() { name=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
pass i1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES
pass i2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
for d in i1 i2; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$d" <<'PY'
import csv, sys, collections
tot = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'qm_read_kernel' in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(tot.items()):
    print("%-4s %-28s %18.0f  per pair %12.3f" % (sys.argv[2], k, v, v / 1e7))
PY
done | tee $OUT/summary.txt
