#!/usr/bin/env bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in fused bf16; do
  out=gpurun_out/r5i_$tag; rm -rf $out; mkdir -p $out
  flag=""; [ $tag = fused ] && flag="--fp8"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-distance-phase $flag > $out/step.log 2>&1
  python tools/step_breakdown.py $out/step 40 > $out/breakdown.txt
  head -42 $out/breakdown.txt
done
