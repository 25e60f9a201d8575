#!/bin/bash
# round 6: the counter passes behind profiles/pmc_traffic.json.  The default leg maps a batch as two parts in flight, one on the pair kernel and one on
# qm_lean_kernel; the passes map the batch as ONE launch of either (QM_SPLIT=1; QM_NO_DUO=1 for qm_lean_kernel) so that a dispatch is 10 M pairs, and the
# default two-part command is traced once more (stats only).  Compact -p and -s as in round 5 (unsplit, -s with QM_SEL_SERIAL).
# usage on the GPU box: bash profiles/r06/pmc_all.sh gpurun_out/<dir> [tag]
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
X="--no-input-variants"
QM_SPLIT=1 bash profiles/r06/run_profile.sh $OUT/dense_pair $X > /dev/null 2>&1
QM_SPLIT=1 QM_NO_DUO=1 bash profiles/r06/run_profile.sh $OUT/dense_lean $X > /dev/null 2>&1
QM_SPLIT=1 bash profiles/r06/run_profile.sh $OUT/ph_compact $X --perfect-hash --ph-compact > /dev/null 2>&1
QM_SPLIT=1 QM_SEL_SERIAL=1 bash profiles/r06/run_profile.sh $OUT/sel $X --sel-aln > /dev/null 2>&1
# the default command, two parts in flight: kernel-trace stats only
mkdir -p $OUT/default_two_parts
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default_two_parts/stats -o s -- python bench.py --no-cpu-baseline --no-other-configs --no-side-legs $X --steps 10 --warmup 2 > $OUT/default_two_parts/stats.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-other-configs --no-side-legs --no-input-variants --steps 10 --warmup 2 (the default: two parts in flight)";
  f=$(find $OUT/default_two_parts/stats -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; echo "# bench line of that run:"; grep '^{"metric"' $OUT/default_two_parts/stats.log | tail -1 | cut -c1-500; } > $OUT/default_two_parts/summary.txt
for d in dense_pair dense_lean ph_compact sel; do echo "== $d"; grep -A4 "^\"Name\"" $OUT/$d/summary.txt | cut -c1-140; done
mkdir -p $OUT/commit && python profiles/r06/make_pmc_traffic.py $OUT ${2:-r06} $OUT/commit
for d in dense_pair dense_lean ph_compact sel default_two_parts; do find $OUT/$d -name "*.csv" -delete; done
