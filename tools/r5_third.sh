#!/usr/bin/env bash
# round 5, third GPU call: throughput of the entry points beside bench.py on one box, then the one-rank RCCL rehearsal: bench
# lines with and without it and a kernel trace of the rehearsal
set -uo pipefail
mkdir -p gpurun_out/entry gpurun_out/rccl1
timeout -k 10 700 python tools/entrypoint_rate.py --out gpurun_out/entry > gpurun_out/entry/stdout.log 2> gpurun_out/entry/stderr.log; echo "entrypoint rc $?"
tail -60 gpurun_out/entry/stdout.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --family-steps 0 > gpurun_out/rccl1/bench_plain.json 2> gpurun_out/rccl1/plain.err
KVQ_DP_SINGLE_RANK=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --family-steps 0 > gpurun_out/rccl1/bench_rccl1.json 2> gpurun_out/rccl1/rccl1.err
export KVQ_DP_SINGLE_RANK=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rccl1/trace -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 --family-steps 0 --no-distance-phase > gpurun_out/rccl1/trace.log 2>&1
python tools/step_breakdown.py gpurun_out/rccl1/trace 60 > gpurun_out/rccl1/breakdown.txt
head -40 gpurun_out/rccl1/breakdown.txt
python - <<'PY'
import json
for f in ("bench_plain", "bench_rccl1"):
    try:
        j = json.loads(open(f"gpurun_out/rccl1/{f}.json").read().strip().splitlines()[-1])
        print(f, round(j["ms_per_step"], 3), "ms/step clock", round(j["clock_mhz"]), "exposed", j.get("exposed_comm_ms_per_step"), "ranks", j.get("rccl_ranks"))
    except Exception as e:
        print(f, "unreadable", e)
PY
