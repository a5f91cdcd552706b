#!/usr/bin/env bash
# round 5, first GPU call: the whole GPU suite (all failures, not -x), then bench.py with the fused quantiser forward and with
# the three-kernel one (same box), pack-in-step figure included
set -uo pipefail
mkdir -p gpurun_out/r5a
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc $?" | tee gpurun_out/r5a/pytest.rc
tail -40 gpurun_out/r5a/pytest.log
for v in 1 0 1 0; do
  KVQ_VQ_FUSED=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --pack-in-step > gpurun_out/r5a/bench_fused$v.$RANDOM.json 2> gpurun_out/r5a/bench_err.log || { echo "bench fused=$v failed"; tail -20 gpurun_out/r5a/bench_err.log; }
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5a/bench_fused*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = j["roofline"]
    print(f.split("/")[-1], "ms/step", round(j["ms_per_step"], 3), "pack-in-step", round(j.get("ms_per_step_pack_in_step", 0), 3), "clock", round(j["clock_mhz"]), "vq kernel us", round(r["avg_launch_us"], 2), "loss", j["final_loss"])
PY
