#!/usr/bin/env bash
# Build libkvq.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/lib"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno -I"$here/../include" -I"$here/csrc" -Wall -Wno-unused-function)
objs=()
for src in "$here"/csrc/*.hip; do
  obj="$here/lib/$(basename "${src%.hip}").o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "$here/csrc/kvq_common.h" -nt "$obj" || "$here/../include/kvq.h" -nt "$obj" ]]; then
    "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$obj" "$@"
  fi
  objs+=("$obj")
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$here/lib/libkvq.so" "${objs[@]}"
echo "built $here/lib/libkvq.so"
