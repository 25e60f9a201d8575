#!/bin/bash
# instruction-cache and LDS-conflict counters of the stage-A kernel (separate --pmc passes; the TCP pass of
# run_pmc_icache.sh hung a box once and is left out); usage: profiles/run_pmc_lds_icache.sh <outdir>
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --steps 1 --warmup 0 $*"
pass() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
pass ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
pass flat SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES
for d in ic1 lds flat; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$d" <<'PY'
import csv, sys, collections
tot = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'qm_read_kernel' in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(tot.items()):
    print("%-6s %-34s %18.0f  per pair %12.3f" % (sys.argv[2], k, v, v / 1e7))
PY
  tail -3 $OUT/$d.log | grep -i "error\|invalid\|not" | head -2
done | tee $OUT/summary.txt
