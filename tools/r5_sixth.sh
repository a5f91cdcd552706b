#!/usr/bin/env bash
# round 5, sixth GPU call: the dropout + residual epilogue (tests, then the same-box A/B by environment)
set -uo pipefail
mkdir -p gpurun_out/r5f
timeout -k 10 900 python -m pytest tests/test_gemm2_gpu.py tests/test_dropres_gpu.py tests/test_engine_small_batches_gpu.py tests/test_engine_base_shapes_gpu.py tests/test_engine_gpu.py tests/test_step_golden.py tests/test_step_bagon_golden.py tests/test_graph_nodes_gpu.py -q -x --timeout 600 > gpurun_out/r5f/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r5f/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_env.sh 4 "dropres_on|KVQ_FUSE_DROPRES=1" "dropres_off|KVQ_FUSE_DROPRES=0" "half_cu_off|KVQ_HALF_CU=0" 2>&1 | tee gpurun_out/r5f/ab_env.txt
