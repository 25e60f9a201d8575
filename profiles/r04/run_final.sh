#!/bin/bash
# end-of-round-4 evidence run (GPU box): full gpu suite, smoke, the driver-style bench line (N=1, every leg), the one-rank nccl bench,
# kernel-trace stats + PMC passes of the default bench and of -s.   usage: bash profiles/r04/run_final.sh gpurun_out/<dir>
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
t0=$(date +%s); timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench took $(( $(date +%s) - t0 )) s"
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["parity"])
for k,v in d.get("other_configs",{}).items():
    if isinstance(v, dict): print(k[:44], v["value"], v["ms_per_step"], v["kernel_ms"], v["roofline"]["frac"], v["parity"])
print("pcie", d.get("pcie_inclusive",{}).get("value"))
e=d.get("end_to_end") or {}; print("e2e", e.get("value"), e.get("pairs"), e.get("without_read_names"))
print("compat_face", json.dumps({k:(v.get("value"), v.get("bit_identical_joint_hits")) for k,v in (d.get("compat_face", {}).get("by_host_threads") or {}).items()}))
PY
bash profiles/r04/run_profile.sh $OUT/prof_dense > /dev/null 2>&1
QM_SPLIT=1 bash profiles/r04/run_profile.sh $OUT/prof_sel --sel-aln > /dev/null 2>&1      # (one launch per kernel and step: counters per 10 M pairs)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sel_parts -o s -- python bench.py --sel-aln --no-cpu-baseline --no-other-configs --no-side-legs --steps 3 --warmup 1 > $OUT/stats_sel_parts.log 2>&1   # the default: two parts in flight
grep -A3 "^\"Name\"" $OUT/prof_dense/summary.txt | cut -c1-160; grep "qm_read_kernel\|qm_h2m\|align2" $OUT/prof_sel/summary.txt | head -6 | cut -c1-160; f=$(find $OUT/stats_sel_parts -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
