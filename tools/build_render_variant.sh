#!/bin/bash
# A variant of the DEFAULT library that differs only in the unit with the stage kernels (render.cpp), for A/B runs inside one GPU
# session (REDNER_AMD_LIB selects the build): tools/build_render_variant.sh <name> <flags...>  ->  variants/<name>.so
# (the other objects are the ones __graft_entry__.build_native() left under build/hip/).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); CSRC=$ROOT/redner_amd/csrc; OBJ=$ROOT/build/hip
mkdir -p $ROOT/variants
FLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -I$CSRC/hip -I$CSRC -Wno-unused-result -pthread -ffp-contract=off -DRDR_PLATFORM_LIBM $*"
/opt/rocm/bin/hipcc -x hip $FLAGS -c $CSRC/render.cpp -o $OBJ/render_$NAME.o 2> $OBJ/render_$NAME.log || { tail -20 $OBJ/render_$NAME.log; exit 1; }
OBJS=$(ls $OBJ/*.o | grep -v "/render\|/tracevar_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $ROOT/variants/$NAME.so $OBJ/render_$NAME.o $OBJS
grep -i "warning: .*occupancy\|spill" $OBJ/render_$NAME.log | head -3 || true
ls -la $ROOT/variants/$NAME.so
