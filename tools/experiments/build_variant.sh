#!/bin/bash
# build_variant.sh NAME "-DFLAG ..." [file.o ...]: a second library under _exp/NAME next to the shipped one (only the objects whose
# sources are newer than the copies taken from crdr_amd/_lib -- or those named -- are recompiled); select it with CRDR_HIP_LIB
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; FLAGS=$2; shift 2 || true
mkdir -p $ROOT/_exp/$NAME
cp -p $ROOT/crdr_amd/_lib/*.o $ROOT/_exp/$NAME/
for o in "$@"; do rm -f $ROOT/_exp/$NAME/$o; done
make -s -C $ROOT/crdr_amd/csrc OUT=$ROOT/_exp/$NAME EXTRA="$FLAGS" -j2
ls -la $ROOT/_exp/$NAME/libcrdr_hip.so
