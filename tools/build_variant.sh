#!/bin/bash
# A second build of the library with extra compiler flags, for A/B runs inside one GPU session (REDNER_AMD_LIB selects the
# build, redner_amd/_capi.py):  tools/build_variant.sh <name> <flags...>  ->  variants/<name>.so   (git-ignored, travels)
# e.g.  tools/build_variant.sh platform_libm -DRDR_PLATFORM_LIBM
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); CSRC=$ROOT/redner_amd/csrc; OBJ=$ROOT/build/variant_$NAME
mkdir -p $OBJ $ROOT/variants
FLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -I$CSRC/hip -I$CSRC -Wno-unused-result -pthread -ffp-contract=off $*"
for f in capi.cpp scene.cpp render.cpp edges.cpp edges_gpu.cpp bvh_gpu.cpp hip/trace.hip; do
  /opt/rocm/bin/hipcc -x hip $FLAGS -c $CSRC/$f -o $OBJ/$(basename $f).o &
done
/opt/rocm/bin/hipcc $FLAGS -c $CSRC/bvh.cpp -o $OBJ/bvh.cpp.o &
gcc -c -I$ROOT/redner_amd/data $CSRC/tables.S -o $OBJ/tables.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $ROOT/variants/$NAME.so $OBJ/*.o
ls -la $ROOT/variants/$NAME.so
