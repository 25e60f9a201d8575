#!/bin/bash
# round 5: the kernels of one timed step of a bench leg with their start offsets and durations (what is in ms_per_step besides the stage-A kernel)
# usage on the GPU box: bash profiles/r05/step_timeline.sh gpurun_out/<dir> [bench flags]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --no-cpu-baseline --no-other-configs --no-side-legs --steps 2 --warmup 1 "$@" > $OUT.log 2>&1 < /dev/null
python - $OUT <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not fs: sys.exit("no trace")
rows = [r for r in csv.DictReader(open(fs[0])) if "build_" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(rows) if "qm_lean_kernel" in r["Kernel_Name"] or "qm_read_kernel" in r["Kernel_Name"]]
i = big[-1]                      # the last step's stage-A launch (of the last part)
j0 = big[-2] + 1 if len(big) > 1 else 0
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[max(j0, i - 12): i + 40]:
    print("%9.1f us +%9.1f us  q%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", ""), r["Kernel_Name"][:90]))
PY
find $OUT -name "*kernel_trace.csv" -delete
