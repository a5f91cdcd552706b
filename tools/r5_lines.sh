#!/usr/bin/env bash
# round 5: the whole GPU suite once more (fp8 / Adam changes since the profile call), then the tracked bench lines of the round from one box
set -uo pipefail
mkdir -p gpurun_out/r5y
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -rf --tb=short > gpurun_out/r5y/pytest.log 2>&1; echo "pytest rc $?"
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r5y/pytest.log | tail -10
bash tools/final_lines.sh r05
