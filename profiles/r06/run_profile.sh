#!/bin/bash
# round 6: kernel-trace stats + PMC passes of a bench.py command (each PMC pass its own rocprofv3 run, --pmc never combined with
# anything but --kernel-trace).  usage on the GPU box: bash profiles/r06/run_profile.sh gpurun_out/<dir> [bench flags]
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-other-configs --no-side-legs $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py $ARGS --steps 3 --warmup 1 > $OUT/stats.log 2>&1
pass() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS --steps 1 --warmup 1 > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS --steps 3 --warmup 1"
  f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -10 "$f" | cut -c1-230
  echo "# bench line of that run:"
  grep '^{"metric"' $OUT/stats.log | tail -1 | cut -c1-600
  echo "# PMC passes (-- python bench.py $ARGS --steps 1 --warmup 1), per dispatch; FETCH_SIZE / WRITE_SIZE in KiB as reported and in GB"
  python - $OUT <<'PY'
import csv, sys, collections, glob
tot = collections.defaultdict(collections.Counter); n = collections.defaultdict(collections.Counter)
for d in ("sq1", "sq2", "fetch", "write", "tcc"):
    for f in glob.glob(sys.argv[1] + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'qm' not in k: continue
            k = k.split('(')[0].replace('void ', '')
            tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k in sorted(tot, key=lambda k: -tot[k].get('SQ_BUSY_CYCLES', 0)):
    print(k)
    for c, v in sorted(tot[k].items()):
        per = v / n[k][c]
        extra = ""
        if c in ("FETCH_SIZE", "WRITE_SIZE"): extra = "  = %.2f GB" % (per * 1024 / 1e9)
        if c == "TCC_MISS_sum": extra = "  x64 B = %.2f GB" % (per * 64 / 1e9)
        print("   %-24s %16.0f per dispatch (%d dispatches)%s" % (c, per, n[k][c], extra))
PY
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-200 | head -60
