#!/bin/bash
# A variant of the DEFAULT library that differs only in hip/trace.hip: tools/build_trace_variant.sh <name> <flags...> -> variants/<name>.so
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); CSRC=$ROOT/redner_amd/csrc; OBJ=$ROOT/build/hip
mkdir -p $ROOT/variants
FLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -I$CSRC/hip -I$CSRC -Wno-unused-result -pthread -ffp-contract=off $*"
/opt/rocm/bin/hipcc -x hip $FLAGS -c $CSRC/hip/trace.hip -o $OBJ/tracevar_$NAME.o
OBJS=$(ls $OBJ/*.o | grep -v "/render_\|/tracevar_\|/trace.hip.o\|/render_exact" )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $ROOT/variants/$NAME.so $OBJ/tracevar_$NAME.o $OBJS
ls -la $ROOT/variants/$NAME.so
