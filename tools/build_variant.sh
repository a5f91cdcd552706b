#!/usr/bin/env bash
# A second build of the library with extra compiler flags on ONE source, for same-box A/B runs (KVQ_LIB_PATH=.../lib/<name>/libkvq.so):
#   tools/build_variant.sh <name> <source stem, e.g. kvq_gemm2> [flags...]   ->  kindergarten-vq-vae_amd/lib/<name>/libkvq.so
# The other objects are the product build's (run kindergarten-vq-vae_amd/build.sh first).
set -euo pipefail
name="$1"; stem="$2"; shift 2
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/../kindergarten-vq-vae_amd" && pwd)"
mkdir -p "$here/lib/$name"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno -mllvm -amdgpu-kernarg-preload-count=16 -I"$here/../include" -I"$here/csrc" -Wall -Wno-unused-function)
"$HIPCC" "${FLAGS[@]}" "$@" -c "$here/csrc/$stem.hip" -o "$here/lib/$name/$stem.o"
objs=("$here/lib/$name/$stem.o")
for src in "$here"/csrc/*.hip; do
  b="$(basename "${src%.hip}")"
  [[ "$b" == "$stem" ]] || objs+=("$here/lib/$b.o")
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$here/lib/$name/libkvq.so" "${objs[@]}"
echo "built $here/lib/$name/libkvq.so"
