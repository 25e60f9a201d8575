#!/bin/bash
# bench + HBM traffic passes for the perfect-hash (-p) index variant (config 4); usage: run_ph_traffic.sh <outdir>
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --perfect-hash ${QM_PH_FLAGS:-} --no-cpu-baseline > $OUT/bench_ph.log 2>&1
tail -1 $OUT/bench_ph.log | cut -c1-600
A="--perfect-hash ${QM_PH_FLAGS:-} --no-cpu-baseline --steps 1 --warmup 0"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python bench.py $A > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
tot = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'qm_read_kernel' in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(tot.items()):
    print("%-12s %18.0f  per pair %12.3f" % (k, v, v / 1e7))
PY
done | tee $OUT/summary.txt
