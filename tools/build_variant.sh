#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: kivi_amd/_variants/libkivi_<name>.so with kivi_mf.hip rebuilt under the flags
# (the other objects come from kivi_amd/_build; select at run time with KIVI_HIP_LIB=...)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p kivi_amd/_variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
/opt/rocm/bin/hipcc $F "$@" -c kivi_amd/csrc/kivi_mf.hip -o kivi_amd/_variants/kivi_mf_$name.o
objs=$(ls kivi_amd/_build/*.o | grep -v kivi_mf.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kivi_amd/_variants/libkivi_$name.so $objs kivi_amd/_variants/kivi_mf_$name.o
rm kivi_amd/_variants/kivi_mf_$name.o
echo built kivi_amd/_variants/libkivi_$name.so
