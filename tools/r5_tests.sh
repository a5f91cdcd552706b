#!/usr/bin/env bash
# the GPU suite with a short report of every failure:  tools/r5_tests.sh [pytest args]
set -uo pipefail
mkdir -p gpurun_out/r5t
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 600 -rf --tb=short "$@" > gpurun_out/r5t/pytest.log 2>&1; echo "pytest rc $?" | tee gpurun_out/r5t/pytest.rc
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r5t/pytest.log | tail -30
