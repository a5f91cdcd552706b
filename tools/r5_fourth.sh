#!/usr/bin/env bash
# round 5, fourth GPU call: GEMM tests with the early-start prologue and the two-workgroups-per-CU tile, the tile probe, the same-box
# A/B of the prologue (lib/g2r4 = -DKVQ_G2_EARLY=0), the bucket sweep of the one-rank RCCL rehearsal
set -uo pipefail
mkdir -p gpurun_out/r5d
timeout -k 10 600 python -m pytest tests/test_gemm2_gpu.py tests/test_engine_small_batches_gpu.py tests/test_engine_base_shapes_gpu.py -q -x --timeout 600 > gpurun_out/r5d/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r5d/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/gemm2_probe_h.py 3 > gpurun_out/r5d/probe_h.txt 2>&1; cat gpurun_out/r5d/probe_h.txt
bash tools/ab_variant.sh g2r4 3 2>&1 | tee gpurun_out/r5d/ab_early.txt
for b in 64 128 256; do
  KVQ_DP_SINGLE_RANK=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --bucket-mib $b 2> gpurun_out/r5d/rccl_b$b.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rccl one rank, bucket $b MiB: %.3f ms @ %.0f MHz exposed %.3f' % (d['ms_per_step'], d['clock_mhz'], d['exposed_comm_ms_per_step']))" | tee -a gpurun_out/r5d/rccl_buckets.txt
done
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no exchange: %.3f ms @ %.0f MHz' % (d['ms_per_step'], d['clock_mhz']))" | tee -a gpurun_out/r5d/rccl_buckets.txt
