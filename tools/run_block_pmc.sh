#!/usr/bin/env bash
# PMC passes (separate runs) over tools/block_pmc.py: attention MFMA kernels + own NT GEMM.  Run on the GPU box.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/blk
rm -rf "$out"; mkdir -p "$out"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/p$i" -- python tools/block_pmc.py > "$out/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
