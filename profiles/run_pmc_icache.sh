#!/bin/bash
# instruction-fetch / issue-stall counters of the mapping kernel; usage: profiles/run_pmc_icache.sh <outdir>
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3-avail list 2>/dev/null | grep -oE "\b(SQC?_[A-Z0-9_]+|TCP_[A-Z0-9_]+)\b" | sort -u > $OUT/avail_counters.txt
ARGS="--no-cpu-baseline --steps 1 --warmup 0 $*"
pass() { name=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
pass ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass ic2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pass ic3 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM
pass ic4 SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES
pass ic5 SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
for d in ic1 ic2 ic3 ic4 ic5 tcp; do
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
  tail -2 $OUT/$d.log | grep -i "error\|invalid" | head -2
done | tee $OUT/summary.txt
