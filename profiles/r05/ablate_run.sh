#!/bin/bash
# on the GPU box: instruction counts of the lean kernel cut off after each phase (differences between neighbours = the phase)
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-other-configs --no-side-legs --steps 1 --warmup 1"
for n in 1 2 3 full; do
  lib=$GRAFT_REPO_ROOT/profiles/ab/libqmap_ab$n.so; [ $n = full ] && lib=$GRAFT_REPO_ROOT/rapmap_amd/libqmap_mi355.so
  QM_LIB_OVERRIDE=$lib timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/ab$n -o p -- python bench.py $ARGS > $OUT/ab$n.log 2>&1
done
python - $OUT <<'PY' | tee $OUT/summary.txt
import csv, sys, glob, collections
print("# lean kernel cut off after phase N (1: characters -> images + staging, 2: + first probe, 3: + scan and walks, full); per pair (10 M pairs per launch)")
prev = None
for n in ("1", "2", "3", "full"):
    tot = collections.Counter(); cnt = collections.Counter()
    for f in glob.glob(sys.argv[1] + "/ab" + n + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if 'qm_lean_kernel' in r['Kernel_Name']:
                tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
    row = {k: tot[k] / cnt[k] / 1e7 for k in tot}
    print("phase<=%-4s " % n + "  ".join("%s %8.1f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), row[k]) for k in sorted(row)))
    if prev: print("   delta    " + "  ".join("%s %8.1f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), row[k] - prev.get(k, 0)) for k in sorted(row)))
    prev = row
PY
