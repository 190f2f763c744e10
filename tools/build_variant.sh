#!/bin/bash
# Build an instrumented / experimental copy of the library beside the product one:
#   tools/build_variant.sh NAME FILE.hip -DFLAG=..   ->  flid_amd/csrc/variants/libflid_tg_NAME.so
# (FILE.hip is recompiled with the extra flags, every other object is the product build's; load with FLID_TG_LIB=<that path>)
set -e
cd "$(dirname "$0")/../flid_amd/csrc"
NAME=$1; SRC=$2; shift 2
make -s
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -ffp-contract=off "$@" -c $SRC -o variants/${SRC%.hip}_$NAME.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS variants/${SRC%.hip}_$NAME.o -o variants/libflid_tg_$NAME.so
echo variants/libflid_tg_$NAME.so
