#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out/r5l
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_engine_gpu.py -q -x --timeout 600 > gpurun_out/r5l/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r5l/pytest.log
[ $rc -eq 0 ] || { grep -n "^E " gpurun_out/r5l/pytest.log | head -20; exit $rc; }
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['final_loss'], '%.3f ms @ %.0f MHz' % (d['ms_per_step'], d['clock_mhz']))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 30 --factors 9 2>/dev/null | p "bf16                        "
  python bench.py --no-cpu-baseline --steps 30 --factors 9 --fp8 2>/dev/null | p "fp8                         "
  KVQ_FP8_ADAM=0 python bench.py --no-cpu-baseline --steps 30 --factors 9 --fp8 2>/dev/null | p "fp8, conversion pass        "
  KVQ_FP8_TILE=0 python bench.py --no-cpu-baseline --steps 30 --factors 9 --fp8 2>/dev/null | p "fp8, 128x256 tile everywhere"
done 2>&1 | tee gpurun_out/r5l/ab_fp8.txt
