#!/usr/bin/env bash
# GEMM counter passes behind profiles/r02_gemm_pmc.md -- run ON the GPU box (through gpurun); counters in separate passes,
# never combined with sys / hip / hsa traces.  Also the FETCH_SIZE / WRITE_SIZE calibration (tools/ubench/fetch_calib.hip).
set -euo pipefail
tag="${1:-r02_gemm}"
out="gpurun_out/$tag"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
export KVQ_PMC_PLAN_DIR="$out"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python tools/gemm2_pmc.py > "$out/$name.log" 2>&1; echo "$name done"; }
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/calib_fetch" -- /tmp/fetch_calib > "$out/calib_fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/calib_write" -- /tmp/fetch_calib > "$out/calib_write.log" 2>&1
echo "calibration done"
