#!/bin/bash
# HBM traffic of the -s kernels (GPU box): separate FETCH_SIZE and WRITE_SIZE passes, summed per kernel over ONE step of 10 M pairs.
#   bash profiles/r04/pmc_traffic_sel.sh <outdir> [bench args]
set -u
OUT=$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--sel-aln --no-cpu-baseline --steps 1 --warmup 0 $*"
pass() { name=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python - $OUT <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(collections.Counter); calls = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        if not any(x in k for x in ("qm_", "rocprim", "Cijk")): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("# per kernel, summed over the launches of ONE step of 10 M pairs (-s); FETCH_SIZE / WRITE_SIZE are KiB as rocprofv3 reports them")
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    print(k)
    for n, v in sorted(c.items()):
        extra = "  = %.2f GB" % (v * 1024 / 1e9) if n in ("FETCH_SIZE", "WRITE_SIZE") else ("  x64 B = %.2f GB" % (v * 64 / 1e9) if n == "TCC_MISS_sum" else "")
        print("   %-14s %16.0f  (%d launches)%s" % (n, v, len(calls[(k, n)]), extra))
PY
