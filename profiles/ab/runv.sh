#!/bin/bash
# usage: runv.sh "<bench flags>" variants...
FL="$1"; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset QM_LIB_OVERRIDE; else export QM_LIB_OVERRIDE=$PWD/rapmap_amd/variants/$v.so; fi
  python bench.py --no-cpu-baseline --steps 3 $FL 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VAR %-10s %-16s %8.2f Mpairs/s  kernel %8.3f ms  step %8.3f ms' % ('$v', '$FL', d['value'], d['config']['map_kernel_ms'], d['ms_per_step']))"
done
