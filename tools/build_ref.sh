#!/usr/bin/env bash
# libkvq.so of another commit into lib/ref/ for same-box A/B runs:  tools/build_ref.sh <commit>   then  KVQ_LIB_PATH=.../lib/ref/libkvq.so
set -euo pipefail
commit="${1:-HEAD}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
tmp="$(mktemp -d)"
git -C "$root" archive "$commit" kindergarten-vq-vae_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/kindergarten-vq-vae_amd/lib/ref"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno ${REF_EXTRA_FLAGS--mllvm -amdgpu-kernarg-preload-count=16} -I"$tmp/include" -I"$tmp/kindergarten-vq-vae_amd/csrc" -Wall -Wno-unused-function)
objs=()
for src in "$tmp"/kindergarten-vq-vae_amd/csrc/*.hip; do
  obj="$tmp/$(basename "${src%.hip}").o"
  /opt/rocm/bin/hipcc "${FLAGS[@]}" -c "$src" -o "$obj" &
  objs+=("$obj")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/kindergarten-vq-vae_amd/lib/ref/libkvq.so" "${objs[@]}"
rm -rf "$tmp"
echo "built lib/ref/libkvq.so from $commit"
