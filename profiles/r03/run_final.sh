#!/bin/bash
# end-of-round-3 evidence run (GPU box): full gpu suite, smoke, the driver-style bench line (N=1, every leg), kernel-trace stats +
# PMC passes of the default bench and of the -s kernels.   usage: bash profiles/r03/run_final.sh gpurun_out/<dir>
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err
bash profiles/run_r03_profile.sh $OUT/prof --no-other-configs --no-side-legs > /dev/null 2>&1; cat $OUT/prof/summary.txt | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sel_stats -o s -- python bench.py --sel-aln --no-other-configs --no-side-legs --no-cpu-baseline --steps 3 --warmup 1 > $OUT/sel_stats.log 2>&1
f=$(find $OUT/sel_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -7 "$f" | cut -c1-200 | tee $OUT/sel_kernel_stats.txt; tail -1 $OUT/sel_stats.log > $OUT/sel_bench_line.json
bash profiles/r03/pmc_sel.sh $OUT/sel_pmc > $OUT/sel_pmc_summary.txt 2>&1
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
for k,v in d.get("other_configs",{}).items(): print(k[:44], v["value"], v["ms_per_step"], v["roofline"]["frac"], v["parity"])
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("pcie_inclusive","end_to_end")})
PY
