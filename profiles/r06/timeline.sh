#!/bin/bash
# round 6: the kernels of ONE step on a time axis (start / end / stream of every dispatch of the last step) from a rocprofv3 kernel trace
# usage on the GPU box: bash profiles/r06/timeline.sh gpurun_out/<dir> [bench flags]
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu-baseline --no-other-configs --no-side-legs --no-input-variants "$@" --steps 2 --warmup 1 > $OUT/trace.log 2>&1
python - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step: from the last stage-A launch backwards to the one before
names = [r['Kernel_Name'] for r in rows]
isA = [i for i, n in enumerate(names) if 'qm_lean_kernel' in n or 'qm_duo_kernel' in n]
# group stage-A launches closer than 5 ms into one step
steps = []
for i in isA:
    if steps and int(rows[i]['Start_Timestamp']) - int(rows[steps[-1][-1]]['Start_Timestamp']) < 5e6: steps[-1].append(i)
    else: steps.append([i])
lo = steps[-1][0]
t0 = int(rows[lo]['Start_Timestamp'])
with open(sys.argv[1] + "/timeline.txt", "w") as o:
    o.write("# start_ms end_ms dur_ms queue kernel (last step of the run; t = 0 at its first stage-A launch)\n")
    for r in rows[lo:]:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        o.write("%9.3f %9.3f %8.3f  q%-3s %s\n" % (s / 1e6, e / 1e6, (e - s) / 1e6, r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0].replace('void ', '')[:90]))
print(open(sys.argv[1] + "/timeline.txt").read()[:6000])
PY
