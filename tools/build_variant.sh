#!/bin/bash
# A/B build of one kernel file: tools/build_variant.sh <name> <source.hip> "<extra hipcc flags>"  ->  tools/ab/lib_<name>.so
# (load it with MK_LIB_OVERRIDE=tools/ab/lib_<name>.so; the other objects come from the regular in-tree build)
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/.."
mkdir -p tools/ab
obj=/tmp/variant_${name}.o
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -x hip -c makani_amd/csrc/$src -o $obj $flags
objs=$(ls makani_amd/csrc/build/*.o | grep -v "/${src}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ab/lib_${name}.so $objs $obj
echo tools/ab/lib_${name}.so
