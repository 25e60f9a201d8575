#!/bin/bash
# round 6: the short loop used while tuning the pair kernel -- one bench line (10 steps) and three counter passes of the default leg, batch as one launch
# usage on the GPU box: bash profiles/r06/quick_pmc.sh gpurun_out/<dir> [bench flags]
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-other-configs --no-side-legs $*"
python bench.py $ARGS --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.log
python -c "
import json;d=json.load(open('$OUT/bench.json'));print('bench:',d['value'],'M pairs/s',d['ms_per_step'],'ms/step kernel',d['roofline'].get('kernel_ms'),'parity',d.get('parity'))"
pass() { name=$1; shift; QM_SPLIT=1 timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS --steps 1 --warmup 1 > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python - $OUT <<'PY'
import csv, sys, collections, glob
tot = collections.defaultdict(collections.Counter); n = collections.defaultdict(collections.Counter); dur = collections.defaultdict(list)
for d in ("sq1", "sq2", "tcc"):
    for f in glob.glob(sys.argv[1] + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'qm' not in k: continue
            k = k.split('(')[0].replace('void ', '')
            tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for f in glob.glob(sys.argv[1] + "/" + d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'qm' in k and d == "sq1": dur[k.split('(')[0].replace('void ', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k in sorted(tot, key=lambda k: -tot[k].get('SQ_BUSY_CYCLES', 0)):
    if 'build' in k: continue
    print(k, "  ms per dispatch (counter pass):", ["%.2f" % x for x in dur.get(k, [])][:4])
    for c, v in sorted(tot[k].items()):
        per = v / n[k][c]
        print("   %-22s %14.0f per dispatch (%d)   %8.1f per pair at 10 M pairs" % (c, per, n[k][c], per / 1e7))
PY
