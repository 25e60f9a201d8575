#!/bin/bash
# instruction-mix PMC passes of the -s kernels (run on the GPU box): profiles/r03/pmc_sel.sh <outdir>
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--sel-aln --no-cpu-baseline --steps 1 --warmup 0"
pass() { name=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD
python - $OUT <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(collections.Counter); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if not any(x in k for x in ("qm_sel", "qm_h2m", "qm_read")): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen: seen.add((f, r["Dispatch_Id"]))
for k, c in tot.items():
    print(k)
    for n, v in sorted(c.items()): print("   %-22s %16.0f" % (n, v))
PY
