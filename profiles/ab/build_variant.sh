#!/bin/bash
# profiles/ab/build_variant.sh <name> <patch.py>: copy rapmap_amd/csrc to a scratch dir, apply a python patch
# (receives the scratch dir as argv[1]) and build rapmap_amd/variants/<name>.so for A/B timing on the GPU box.
set -e
NAME=$1; PATCH=$2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/qm_variant_$NAME
rm -rf $W && mkdir -p $W && cp $ROOT/rapmap_amd/csrc/*.h $ROOT/rapmap_amd/csrc/*.hip $ROOT/rapmap_amd/csrc/*.inl $ROOT/rapmap_amd/csrc/*.cpp $W/
sed -i "s|#include \"../../include/qmap_mi355.h\"|#include \"$ROOT/include/qmap_mi355.h\"|" $W/qm_mapper.inl
[ -n "$PATCH" ] && python $PATCH $W
cd $W
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -Wno-unused-value -Wno-unused-result -mllvm -sink-insts-to-avoid-spills=true $QM_XFLAGS"
/opt/rocm/bin/hipcc $FL -c qm_kernels.hip -o k.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "qm_read_kernelILi2ELi[0-9]ELi0E" | grep -E "VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize" | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' '
echo " <= $NAME"
/opt/rocm/bin/hipcc $FL -c qm_host.hip -o h.o
g++ -O2 -std=c++17 -fPIC -I$ROOT/include -c qm_indexer.cpp -o i.o
sed -i "s|#include \"../../include/qmap_mi355.h\"|#include \"$ROOT/include/qmap_mi355.h\"|" qm_io.cpp
g++ -O2 -std=c++17 -fPIC -pthread -I$ROOT/include -c qm_io.cpp -o io.o
mkdir -p $ROOT/rapmap_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/rapmap_amd/variants/$NAME.so k.o h.o i.o io.o -pthread -lz
