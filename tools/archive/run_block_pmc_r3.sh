#!/usr/bin/env bash
# The HBM-streaming kernels of the step (LayerNorm forward / backward, attention forward / backward) and a plain device copy on COLD
# operands (tools/cold_stream_probe.py): durations + fabric bytes, separate --pmc passes.  Run ON the GPU box; then
#   python tools/make_block_pmc_summary.py gpurun_out/r03_blk r03
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_blk
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python tools/cold_stream_probe.py > "$out/trace.log" 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python tools/cold_stream_probe.py > "$out/fetch.log" 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python tools/cold_stream_probe.py > "$out/write.log" 2>&1; echo "write rc=$?"
