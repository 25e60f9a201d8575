#!/bin/bash
# profiles/ab/run_ablation.sh <outdir>: instruction counters of the stage-A kernel for the ablation variants
# (profiles/ab/ablate_patch.py, built with profiles/ab/build_variant.sh abl<n>) and for the product library.
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${QM_ABL_LIST:-abl1 abl2 abl3 abl4 abl5 abl6 main}; do
  if [ "$v" = main ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/rapmap_amd/variants/$v.so; fi
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES \
    --kernel-trace --output-format csv -d $OUT/$v -o p -- python bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/$v.log 2>&1
  f=$(find $OUT/$v -name "*counter_collection.csv" | head -1)
  k=$(find $OUT/$v -name "*kernel_trace.csv" | head -1)
  python - "$f" "$k" "$v" <<'PY'
import csv, sys, collections
tot = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'qm_read_kernel' in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value'])
ms = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in csv.DictReader(open(sys.argv[2])) if 'qm_read_kernel' in r['Kernel_Name'] ]
print("%-5s kernel %7.2f ms | per pair: " % (sys.argv[3], sum(ms) / max(1, len(ms))) + "  ".join("%s %.0f" % (k.replace('SQ_INSTS_', ''), v / 1e7) for k, v in sorted(tot.items()) if k != 'SQ_WAVES'))
PY
done | tee $OUT/summary.txt
