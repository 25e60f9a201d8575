#!/bin/bash
# kernel-trace stats of config 4 (perfect hash) and the instruction mix of the -s kernels; usage: profiles/run_extra_stats.sh <outdir>
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ph_stats -o s -- python bench.py --perfect-hash --no-cpu-baseline --steps 3 > $OUT/ph_stats.log 2>&1
f=$(find $OUT/ph_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-200 > $OUT/ph_kernel_stats.txt
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/sel_mix -o p -- python bench.py --sel-aln --no-cpu-baseline --steps 1 --warmup 0 > $OUT/sel_mix.log 2>&1
f=$(find $OUT/sel_mix -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' > $OUT/sel_mix.txt
import csv, sys, collections
tot = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    for name in ('qm_read_kernel', 'qm_sel_align_kernel', 'qm_sel_plan_kernel', 'qm_sel_finish_kernel'):
        if name in k: tot[name][r['Counter_Name']] += float(r['Counter_Value'])
for name, c in tot.items():
    for k, v in sorted(c.items()):
        print("%-22s %-18s %16.0f  per pair %10.2f" % (name, k, v, v / 1e7))
PY
cat $OUT/ph_kernel_stats.txt $OUT/sel_mix.txt
