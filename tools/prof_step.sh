#!/usr/bin/env bash
# one rocprofv3 --kernel-trace pass over bench.py and the per-step breakdown:  tools/prof_step.sh <tag> [env assignments...]
set -euo pipefail
tag="$1"; shift
out="gpurun_out/$tag"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
for kv in "$@"; do export "$kv"; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/step" -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 --family-steps 0 --no-distance-phase > "$out/step.log" 2>&1
python tools/step_breakdown.py "$out/step" 60 > "$out/breakdown.txt"
cat "$out/breakdown.txt"
