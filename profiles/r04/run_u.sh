#!/bin/bash
# does a single stage-A launch with an oversubscribed grid (QM_GRID_OVERSUB blocks per resident block) match the parts in flight?
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, env, flags
  env $2 timeout 600 python bench.py $3 --no-cpu-baseline --no-other-configs --no-side-legs --steps 10 --warmup 3 2>$OUT/$1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('map_kernel_ms'))"
}
run dense_split1_over1 "QM_SPLIT=1 QM_GRID_OVERSUB=1" ""
run dense_split1_over2 "QM_SPLIT=1 QM_GRID_OVERSUB=2" ""
run dense_split1_over3 "QM_SPLIT=1 QM_GRID_OVERSUB=3" ""
run dense_split1_over4 "QM_SPLIT=1 QM_GRID_OVERSUB=4" ""
run dense_split1_over8 "QM_SPLIT=1 QM_GRID_OVERSUB=8" ""
run dense_split3_over1 "QM_SPLIT=3 QM_GRID_OVERSUB=1" ""
run dense_split3_over2 "QM_SPLIT=3 QM_GRID_OVERSUB=2" ""
run dense_split2_over2 "QM_SPLIT=2 QM_GRID_OVERSUB=2" ""
run sel_split1_over3 "QM_SPLIT=1 QM_GRID_OVERSUB=3" "--sel-aln"
run sel_split2_over2 "QM_SPLIT=2 QM_GRID_OVERSUB=2" "--sel-aln"
run ph_split1_over3 "QM_SPLIT=1 QM_GRID_OVERSUB=3" "--perfect-hash --ph-compact"
run ph_split3_over2 "QM_SPLIT=3 QM_GRID_OVERSUB=2" "--perfect-hash --ph-compact"
