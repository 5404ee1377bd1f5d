#!/bin/bash
# Build controllora_amd/_build_prev/libclora.so = the CURRENT library with some translation units taken from an older commit,
# for same-box A/B runs through CLORA_LIB_PATH (the C ABI must match the current capi.py, so whole old trees often cannot be
# used).  Run in the dev container (needs .git), before gpurun: the .so travels with the snapshot.
#   tools/build_prev_lib.sh <rev>:<file.hip> [<rev>:<file.hip> ...]
# e.g. the GroupNorm / LayerNorm / adapter-wgrad loops before the loads-in-flight rewrite (revisions from
# `git log --oneline -- controllora_amd/csrc/<file>`; the file must still match the current include/clora.h structs):
#   tools/build_prev_lib.sh 62e4530:clora_norm.hip b00c262:clora_lora.hip
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/controllora_amd/_build_prev
tmp=$(mktemp -d)
mkdir -p "$out" "$tmp/controllora_amd/csrc" "$tmp/include"
cp "$root"/controllora_amd/csrc/* "$tmp/controllora_amd/csrc/"
cp "$root"/include/clora.h "$tmp/include/"
for spec in "$@"; do
  rev=${spec%%:*}; f=${spec#*:}
  git -C "$root" show "$rev:controllora_amd/csrc/$f" > "$tmp/controllora_amd/csrc/$f"; echo "  $f <- $rev"
done
objs=""
for src in "$tmp"/controllora_amd/csrc/*.hip; do
  o=$tmp/$(basename "${src%.hip}").o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -c "$src" -o "$o"
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$out/libclora.so"
echo "built $out/libclora.so"
rm -rf "$tmp"
