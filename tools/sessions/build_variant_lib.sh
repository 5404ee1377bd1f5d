#!/bin/bash
# Build controllora_amd/_build_variant/libclora.so from the CURRENT sources with extra compiler flags (experiment macros such as
# -DCLORA_EPI_SINGLE_PASS), for same-box A/B runs through CLORA_LIB_PATH.  Run in the dev container before gpurun.
#   tools/build_variant_lib.sh -DCLORA_EPI_SINGLE_PASS
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/controllora_amd/_build_variant
mkdir -p "$out"
objs=""
for src in "$root"/controllora_amd/csrc/*.hip; do
  o=$out/$(basename "${src%.hip}").o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC "$@" -c "$src" -o "$o"
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$out/libclora.so"
echo "built $out/libclora.so with $*"
