#!/usr/bin/env bash
# round 5, closing call: the whole GPU suite, smoke(), then the round's profile recipe (step trace, VQ counter passes, bench line with the CPU baseline)
set -uo pipefail
mkdir -p gpurun_out/r5z
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -rf --tb=short > gpurun_out/r5z/pytest.log 2>&1; echo "pytest rc $?" | tee gpurun_out/r5z/pytest.rc
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r5z/pytest.log | tail -10
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r5z/smoke.log
bash tools/run_profiles.sh r05 > gpurun_out/r5z/profiles.log 2>&1 || echo "run_profiles failed"
tail -c 1500 gpurun_out/r05/bench_line.json
