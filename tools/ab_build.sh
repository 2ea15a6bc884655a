#!/bin/bash
# Builds a variant of the library with extra compiler defines into build/ab/<tag>/libfasttrack_amd.so (FT_LIB=<that path>
# selects it in the Python driver) - for A/B measurements of compile-time choices in ONE gpurun call.
# usage: tools/ab_build.sh <tag> "-DOD_KPW=1 ..."
set -e
TAG=$1; DEFS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/fasttrack_amd/ab_$TAG
rm -rf $OUT; mkdir -p $OUT
cp $ROOT/fasttrack_amd/csrc/*.cpp $ROOT/fasttrack_amd/csrc/*.hip $ROOT/fasttrack_amd/csrc/*.h $ROOT/fasttrack_amd/csrc/*.inc $ROOT/fasttrack_amd/csrc/Makefile $OUT/
mkdir -p $OUT/../../include
make -C $OUT -j8 OUT=libfasttrack_amd.so HDRS="" CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function -Wno-unused-value -Wno-unused-result -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -I$ROOT/include -I$ROOT/fasttrack_amd/csrc $DEFS" 2>&1 | grep -E "error" || true
rm -f $OUT/*.o $OUT/*.cpp $OUT/*.hip $OUT/*.h $OUT/*.inc $OUT/Makefile
ls -la $OUT
