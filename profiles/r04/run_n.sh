#!/bin/bash
set -u
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
g++ -O3 -pthread profiles/microbench/pagecache_scan.cpp -o /tmp/pagecache_scan
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from rapmap_amd import synth
n = 20_000_000
seq = np.random.default_rng(1).choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n * 100)
synth.write_fastq("/tmp/scan.fq", seq, n, 100, 1)
PY
/tmp/pagecache_scan /tmp/scan.fq 8 > $OUT/pagecache_scan.txt 2>&1; cat $OUT/pagecache_scan.txt; rm -f /tmp/scan.fq
