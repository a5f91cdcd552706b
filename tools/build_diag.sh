#!/usr/bin/env bash
# Diagnostic copy of the library with in-kernel phase stamps in the GEMM family (never loaded by the product or the tests):
#   tools/build_diag.sh  ->  kindergarten-vq-vae_amd/lib/diag/libkvq.so ; select it with KVQ_LIB_DIR=.../lib/diag
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/../kindergarten-vq-vae_amd" && pwd)"
mkdir -p "$here/lib/diag"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno -I"$here/../include" -I"$here/csrc" -Wall -Wno-unused-function)
"$HIPCC" "${FLAGS[@]}" -DKVQ_G2_DIAG -c "$here/csrc/kvq_gemm2.hip" -o "$here/lib/diag/kvq_gemm2.o" &
"$HIPCC" "${FLAGS[@]}" -mllvm -amdgpu-kernarg-preload-count=16 -DKVQ_VQ_DIAG -c "$here/csrc/kvq_vq.hip" -o "$here/lib/diag/kvq_vq.o" &
"$HIPCC" "${FLAGS[@]}" -mllvm -amdgpu-kernarg-preload-count=16 -DKVQ_NN_DIAG -c "$here/csrc/kvq_nn.hip" -o "$here/lib/diag/kvq_nn.o" &
wait
objs=("$here/lib/diag/kvq_gemm2.o" "$here/lib/diag/kvq_vq.o" "$here/lib/diag/kvq_nn.o")
for src in "$here"/csrc/*.hip; do
  b="$(basename "${src%.hip}")"
  [[ "$b" == kvq_gemm2 || "$b" == kvq_vq || "$b" == kvq_nn ]] || objs+=("$here/lib/$b.o")
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$here/lib/diag/libkvq.so" "${objs[@]}"
echo "built $here/lib/diag/libkvq.so"
