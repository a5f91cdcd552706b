#!/usr/bin/env bash
# run-to-run determinism of the fp8 step (final loss of N steps, repeated): gpurun -- bash tools/fp8_determinism.sh
for flag in --fp8-all --fp8; do
for n in 12 20 35; do for i in 1 2 3; do python bench.py --no-cpu-baseline --factors 9 --steps $n --warmup 0 --family-steps 0 $flag 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$flag steps $n:', d['final_loss'])"; done; done; done
