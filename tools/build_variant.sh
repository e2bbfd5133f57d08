#!/bin/bash
# tools/build_variant.sh NAME SOURCE.hip "-DFLAG=..."  ->  tools/_build/libpconv_hip_NAME.so
# (one translation unit recompiled with extra flags, the rest taken from the normal build)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
obj=pseudocylindrical_convolution_amd/build
mkdir -p tools/_build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude $flags -c pseudocylindrical_convolution_amd/csrc/$src -o tools/_build/$name.o 2>/dev/null
others=$(ls $obj/hip/*.o | grep -v "/$src.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_build/libpconv_hip_$name.so tools/_build/$name.o $others
echo built tools/_build/libpconv_hip_$name.so
