#!/bin/bash
# tools/variant.sh NAME SRC.hip FLAGS...: a variant libgeoformer_hip.so (tools/ab/NAME.so) whose SRC is rebuilt with the extra FLAGS,
# everything else from the cached objects of the default build (run `python -m geoformer_amd.build` first).  Use with GF_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p tools/ab
obj=tools/ab/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c geoformer_amd/csrc/$src -o $obj
others=$(ls geoformer_amd/csrc/_obj/*.o | grep -v "/$src\.")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/$name.so $obj $others
echo tools/ab/$name.so
