#!/bin/bash
# build the library of another git revision next to the current one (same-box A/B with scripts/ab_bench_lib.py):
#   scripts/build_rev.sh REV padne_amd/libpadne_REV.so
set -e
REV=$1; OUT=$(realpath -m $2); T=$(mktemp -d)
git archive $REV padne_amd/csrc include | tar -x -C $T
cd $T/padne_amd/csrc
for f in capi spmv spmm pcg assemble comm amg generate kkt; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value -Wno-unused-result -c $f.hip -o $f.o &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT capi.o spmv.o spmm.o pcg.o assemble.o comm.o amg.o generate.o kkt.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf $T; ls -la $OUT
