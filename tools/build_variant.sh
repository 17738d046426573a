#!/bin/bash
# timing experiments: libstep_hip_<tag>.so = the current objects with ONE source recompiled under extra flags
# usage: tools/build_variant.sh <tag> <source.hip> <flags...>
set -e
cd "$(dirname "$0")/.."; tag=$1; src=$2; shift 2
mkdir -p step_amd/build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip "$@" -c step_amd/csrc/$src -o step_amd/build/var/$tag.o
objs=$(ls step_amd/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o step_amd/libstep_hip_$tag.so $objs step_amd/build/var/$tag.o -ldl
