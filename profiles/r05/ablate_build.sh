#!/bin/bash
# profiling variants of the library: the lean kernel cut off after phase 1 (characters -> images), 2 (+ first probe), 3 (+ scan and
# walks); built here (hipcc cross-compiles), they travel to the GPU box as profiles/ab/libqmap_ab<N>.so (git-ignored)
set -e
cd $(dirname $0)/../../rapmap_amd/csrc
for n in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -mllvm -sink-insts-to-avoid-spills=true -DQM_LEAN_ABLATE=$n -c qm_kernels_lean.hip -o /tmp/lean_ab$n.o
  objs=$(ls *.o | grep -v qm_kernels_lean.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/ab/libqmap_ab$n.so $objs /tmp/lean_ab$n.o -pthread -lz
done
ls -la ../../profiles/ab/*.so
