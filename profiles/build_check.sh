#!/bin/bash
# build the library; non-zero exit when the compile fails (use before every gpurun)
set -e
cd "$(dirname "$0")/../rapmap_amd/csrc"
make -j6 2>&1 | (grep -E "error|Error" && exit 1 || true)
test ../libqmap_mi355.so -nt qm_kernels.hip -o ../libqmap_mi355.so -nt qm_host.hip
