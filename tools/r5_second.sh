#!/usr/bin/env bash
# round 5, second GPU call: the round's profile recipe (step trace, VQ counter passes, bench line with the CPU baseline), then the
# two traffic passes again with the three-kernel quantiser forward (KVQ_VQ_FUSED=0) for the fused / split comparison
set -uo pipefail
bash tools/run_profiles.sh r05 || echo "run_profiles failed"
out=gpurun_out/r05
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export KVQ_VQ_FUSED=0
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch_split" -- python tools/vq_only.py > "$out/pmc_fetch_split.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write_split" -- python tools/vq_only.py > "$out/pmc_write_split.log" 2>&1
echo "split-variant traffic passes done"
