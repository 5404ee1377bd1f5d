#!/bin/bash
# Build controllora_amd/_build_v_<name>/libclora.so: clora_gemm.hip recompiled with extra flags, every other object taken from the
# default build (controllora_amd/_build).  For same-box A/B and diagnostic runs through CLORA_LIB_PATH.
#   tools/build_named_variant.sh hoist_all -DCLORA_HOIST_ALL
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/controllora_amd/_build_v_$name
mkdir -p "$out"
files=${CLORA_VARIANT_FILES:-clora_gemm}
objs=""
for src in "$root"/controllora_amd/csrc/*.hip; do
  b=$(basename "${src%.hip}")
  if [[ " $files " == *" $b "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$src" -o "$out/$b.o"
    objs="$objs $out/$b.o"
  else
    objs="$objs $root/controllora_amd/_build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$out/libclora.so"
echo "built $out/libclora.so ($files with $*)"
