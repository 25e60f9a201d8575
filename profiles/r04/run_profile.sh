#!/bin/bash
# round 4 profile of one bench configuration (GPU box): rocprofv3 kernel-trace stats of `bench.py --steps 3 --warmup 1`, then PMC
# passes (each its own rocprofv3 run, --pmc with --kernel-trace only) of `--steps 1 --warmup 1` -- the warm-up step absorbs the
# first call's buffer growth and relaunch, counters are reported PER DISPATCH (average over a kernel's dispatches).
#   bash profiles/r04/run_profile.sh <outdir> [bench flags, e.g. --sel-aln]
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-other-configs --no-side-legs $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py $ARGS --steps 3 --warmup 1 > $OUT/stats.log 2>&1
pass() { name=$1; shift; timeout 500 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS --steps 1 --warmup 1 > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS --steps 3 --warmup 1"
  f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -10 "$f" | cut -c1-220
  echo "# bench line of that run:"; grep "^{" $OUT/stats.log | tail -1 | cut -c1-600
  echo "# PMC passes (-- python bench.py $ARGS --steps 1 --warmup 1), per dispatch; FETCH_SIZE / WRITE_SIZE in KiB as reported and in GB"
  python - $OUT <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(collections.Counter); disp = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-64:]
        if "qm_" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("FETCH_SIZE", 0))):
    print(k)
    for n, v in sorted(c.items()):
        d = max(1, len(disp[(k, n)])); a = v / d
        extra = "  = %.2f GB" % (a * 1024 / 1e9) if n in ("FETCH_SIZE", "WRITE_SIZE") else ("  x64 B = %.2f GB" % (a * 64 / 1e9) if n == "TCC_MISS_sum" else "")
        print("   %-20s %18.0f per dispatch (%d dispatches)%s" % (n, a, d, extra))
PY
} > $OUT/summary.txt 2>&1
head -12 $OUT/summary.txt | cut -c1-200
