#!/bin/bash
# profiles/ab/build_variant.sh <name> [patch.py]: copy rapmap_amd/csrc to a scratch tree, apply a python patch (it gets
# the scratch csrc directory as argv[1]) and build rapmap_amd/variants/<name>.so for A/B timing on the GPU box
# (QM_LIB_OVERRIDE=... python bench.py).  Extra compiler flags through QM_XFLAGS (e.g. -DQM_TIMING).
set -e
NAME=$1; PATCH=$2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/qm_variant_$NAME
rm -rf $W && mkdir -p $W/rapmap_amd/csrc && ln -s $ROOT/include $W/include
cp $ROOT/rapmap_amd/csrc/*.h $ROOT/rapmap_amd/csrc/*.hip $ROOT/rapmap_amd/csrc/*.inl $ROOT/rapmap_amd/csrc/*.cpp $ROOT/rapmap_amd/csrc/Makefile $W/rapmap_amd/csrc/
[ -n "$PATCH" ] && python $PATCH $W/rapmap_amd/csrc
cd $W/rapmap_amd/csrc
make -j6 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -I../../include $QM_XFLAGS" > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
mkdir -p $ROOT/rapmap_amd/variants
cp $W/rapmap_amd/libqmap_mi355.so $ROOT/rapmap_amd/variants/$NAME.so
echo "built rapmap_amd/variants/$NAME.so"
