#!/bin/bash
# A/B builds of the tile engine: scripts/build_variants.sh NAME "-DTDX_VAR_..." -> taudem_amd/variants/NAME/libtaudem_amd.so
# (the tools find it through LD_LIBRARY_PATH, which precedes their RUNPATH); only the listed objects are rebuilt (third argument; default: the ones
# that instantiate relaxation kernels)
set -e
cd "$(dirname "$0")/../taudem_amd/csrc"
NAME=$1; FLAGS=$2; FILES=${3:-"pitremove d8flowdir dinfflowdir"}
OUT=../variants/$NAME
mkdir -p $OUT/obj
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include"
for f in $FILES; do /opt/rocm/bin/hipcc $HIPFLAGS $FLAGS -c $f.hip -o $OUT/obj/$f.o & done; wait
OBJ=""
for o in build/*.o; do b=$(basename $o); if [ -f $OUT/obj/$b ]; then OBJ="$OBJ $OUT/obj/$b"; else OBJ="$OBJ $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtaudem_amd.so $OBJ -lz -ldl -lpthread
rm -rf $OUT/obj
ls -la $OUT
