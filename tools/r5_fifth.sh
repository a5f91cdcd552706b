#!/usr/bin/env bash
# round 5, fifth GPU call: GEMM / engine tests with the two-per-CU routing, then same-box A/Bs by environment: the routing
# (KVQ_HALF_CU), the reductions on a side stream (KVQ_RED_STREAM), and a kernel trace of the entry point's training loop
set -uo pipefail
mkdir -p gpurun_out/r5e
timeout -k 10 600 python -m pytest tests/test_gemm2_gpu.py tests/test_engine_small_batches_gpu.py tests/test_engine_base_shapes_gpu.py tests/test_engine_gpu.py tests/test_step_golden.py -q -x --timeout 600 > gpurun_out/r5e/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r5e/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_env.sh 3 "half_cu_on|KVQ_HALF_CU=1" "half_cu_off|KVQ_HALF_CU=0" "red_stream|KVQ_RED_STREAM=1" 2>&1 | tee gpurun_out/r5e/ab_env.txt
# the training loop of the entry point under the kernel tracer (one epoch of 120 steps + validation)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
work=$PWD/gpurun_out/r5e/main; mkdir -p $work/data
export PYTHONPATH=$PWD/kindergarten-vq-vae_amd KVQ_SYNTHETIC_SENTENCES=51200 KVQ_N_EPOCHS=1 KVQ_N_EPOCHS_TO_DECODE_AFTER=1000000 KVQ_RUNS_DIR="'$work/runs'" KVQ_EXPORT_CHECKPOINT=False KVQ_WANDB_MODE="'disabled'"
export KVQ_SENTENCES_PATH="'$work/data/dSentences_sentences_clean.npy'" KVQ_LATENT_CLASSES_LABELS_PATH="'$work/data/dSentences_latent_classes_labels_clean.npy'" KVQ_LATENT_CLASSES_ONE_HOT_PATH="'$work/data/dSentences_latent_classes_one_hot_clean.npy'"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5e/main_trace -- python kindergarten-vq-vae_amd/models/shelgon3/main.py > gpurun_out/r5e/main_trace.log 2>&1; echo "main trace rc $?"
tail -5 gpurun_out/r5e/main_trace.log
