#!/usr/bin/env bash
# round 5: in-kernel phase stamps of the GEMM family as it stands (diagnostic build, profiles/r05_gemm_ceiling.md)
set -uo pipefail
mkdir -p gpurun_out/r5s
bash tools/build_diag.sh > gpurun_out/r5s/build.log 2>&1 || { tail -5 gpurun_out/r5s/build.log; exit 1; }
for spec in "nt 8192 768 768 128x192" "nt 8192 768 3072 128x192" "nn 8192 768 2304 128x192" "nt 8192 2304 768 128x192h" "nt 8192 3072 768 128x192h" \
            "nt 8192 3072 768 256x192" "nt 8192 30528 768 256x256" "nt 8192 18432 768 256x256" "tn 18432 768 8192 256x256" "tn 3072 768 8192 256x256"; do
  timeout -k 10 120 python tools/gemm2_stamps.py $spec 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5s/stamps.txt
done
cat gpurun_out/r5s/stamps.txt
