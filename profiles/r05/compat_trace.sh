#!/bin/bash
# round 5: kernel trace of the compat face's batching service (8 worker threads): the kernels of one pass with their gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python profiles/r05/compat_probe.py 2000000 8 > /dev/null 2>&1   # builds /tmp/cp/compat_bench, reads.bin, index in /dev/shm
IDX=$(cat /tmp/cp/idx.txt)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cp_tr -o t -- /tmp/cp/compat_bench $IDX /tmp/cp/reads.bin 2000000 100 8 10000 --repeat 2 2>&1 | tail -1 | cut -c1-200
f=$(find gpurun_out/cp_tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || exit 1; head -24 "$f" | cut -c1-150
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/cp_tr/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if "build_" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# print one pass: find a qm_read_kernel in the middle, print kernels on same stream around it
mid=[i for i,r in enumerate(rows) if "qm_read_kernel" in r["Kernel_Name"]]
i=mid[len(mid)//2]; st=rows[i]["Stream_Id"] if "Stream_Id" in rows[i] else None
q=rows[i].get("Queue_Id")
sel=[r for r in rows if r.get("Queue_Id")==q]
j=sel.index(rows[i]); t0=int(sel[j]["Start_Timestamp"])
for r in sel[max(0,j-6):j+22]:
    print("%9.1f us +%8.1f us  %s grid %s" % ((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:70],r.get("Grid_Size_X","")))
PY
find gpurun_out/cp_tr -name "*kernel_trace.csv" -delete
