#!/usr/bin/env bash
# The profile recipe behind profiles/<tag>_summary.md -- run ON the GPU box (through gpurun), then
#   python tools/make_profile_summary.py gpurun_out/<tag> <tag>
# here.  One rocprofv3 pass for the whole step (--kernel-trace --stats), then four SEPARATE --pmc passes over the VQ kernels
# (counters in their own runs, never combined with sys / hip / hsa traces).
set -euo pipefail
tag="${1:-r01}"
out="gpurun_out/$tag"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/step" -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 --family-steps 0 --no-distance-phase > "$out/step.log" 2>&1
echo "step pass done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python tools/vq_only.py > "$out/pmc_fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python tools/vq_only.py > "$out/pmc_write.log" 2>&1
echo "traffic passes done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$out/pmc_sq" -- python tools/vq_only.py > "$out/pmc_sq.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_grbm" -- python tools/vq_only.py > "$out/pmc_grbm.log" 2>&1
echo "counter passes done"
# the large-codebook point (BASELINE.json configs[3]): fabric traffic of the quantiser kernels at K = 8192
VQ_K=8192 timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch_k8192" -- python tools/vq_only.py > "$out/pmc_fetch_k8192.log" 2>&1
VQ_K=8192 timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write_k8192" -- python tools/vq_only.py > "$out/pmc_write_k8192.log" 2>&1
echo "K = 8192 traffic passes done"
timeout -k 10 300 python bench.py > "$out/bench_line.json" 2> "$out/bench.err"
tail -c 600 "$out/bench_line.json"
