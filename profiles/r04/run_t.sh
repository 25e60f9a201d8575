#!/bin/bash
# kernel trace of split runs: when do the parts' stage-A kernels start and end?  + part-count sweep
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu-baseline --no-other-configs --no-side-legs --steps 3 --warmup 1 > $OUT/trace.log 2>&1
python - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/trace/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "qm_read_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-9:]:
    print("stage A  start %9.3f ms  end %9.3f ms  dur %7.3f ms  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "?")))
PY
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-180
for k in 5 6 8; do
  QM_SPLIT=$k timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-side-legs --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense QM_SPLIT=$k', d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))"
done
