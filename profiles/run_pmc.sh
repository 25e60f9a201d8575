#!/bin/bash
# PMC passes for the mapping kernel (run on the GPU box through gpurun); each pass is its own rocprofv3 run.
# usage: profiles/run_pmc.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --steps 1 --warmup 0 $*"
pass() { name=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python bench.py $ARGS > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
find $OUT -name "*.csv" | head -30
