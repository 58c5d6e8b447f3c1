#!/bin/bash
# Builds a library variant for same-box A/B runs:  tools/build_variant.sh <name> <source.hip> [-DFLAG ...]
# -> foundpose_amd/lib/<name>.so = the current objects with <source> recompiled under the extra flags.
# (swap it in on the GPU box with: cp foundpose_amd/lib/<name>.so foundpose_amd/lib/libfoundpose_amd.so)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
python -m foundpose_amd.build > /dev/null
obj=/tmp/variant_$name.o
timeout 900 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form \
  "$@" -x hip -c foundpose_amd/csrc/$src -o $obj
objs=$(ls foundpose_amd/lib/obj/*.o | grep -v "/${src%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o foundpose_amd/lib/$name.so $objs $obj
echo foundpose_amd/lib/$name.so
