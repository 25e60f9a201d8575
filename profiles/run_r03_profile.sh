#!/bin/bash
# round 3: kernel-trace stats + PMC passes of the config-2 bench (each pass its own rocprofv3 run, --pmc never combined with
# anything but --kernel-trace); usage on the GPU box: bash profiles/run_r03_profile.sh gpurun_out/r03prof [bench flags]
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py $ARGS --steps 3 --warmup 1 > $OUT/stats.log 2>&1
pass() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS --steps 1 --warmup 0 > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
{
  echo "# kernel-trace stats (rocprofv3 --kernel-trace --stats), bench args: $ARGS --steps 3 --warmup 1"
  f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-230
  echo "# PMC passes (one rocprofv3 --pmc run each), per launch of the stage-A kernel"
  for d in sq1 sq2 fetch write tcc; do
    f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$d" <<'PY'
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'qm_read_kernel' in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k, v in sorted(tot.items()):
    print("%-6s %-28s %18.0f  per pair %12.3f  (%d dispatch(es))" % (sys.argv[2], k, v, v / 1e7, n[k]))
PY
  done
  tail -1 $OUT/stats.log | cut -c1-400
} | tee $OUT/summary.txt
