#!/bin/bash
# on the GPU box: time each rapmap_amd/variants/*.so given on the command line with the config-2 bench
for v in "$@"; do
  QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/$v.so python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VAR %-14s %8.2f Mpairs/s  kernel %8.3f ms  step %8.3f ms' % ('$v', d['value'], d['config']['map_kernel_ms'], d['ms_per_step']))"
done
