#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  ->  kivi_amd/_variants/libkivi_<name>.so, every source rebuilt under the
# flags; select at run time with KIVI_HIP_LIB=<path>.  The tuning / ablation tools use
#     tools/build_variant.sh tuning -DKIVI_TUNING
# (environment knobs, losing and result-changing instantiations: none of them is in the product library).
set -e
name=$1; shift
cd "$(dirname "$0")/.."
out=kivi_amd/_variants; mkdir -p $out/obj_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
pids=""
for src in kivi_abi kivi_pack kivi_gemv_k kivi_gemv_v kivi_gemv_compat kivi_softmax kivi_layer kivi_gqa kivi_mf; do
  /opt/rocm/bin/hipcc $F "$@" -c kivi_amd/csrc/$src.hip -o $out/obj_$name/$src.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libkivi_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/libkivi_$name.so
