#!/usr/bin/env bash
# same-box A/B: bf16 / fp8 where it pays / fp8 everywhere, nine codebooks (BASELINE.json configs[4] on one GPU)
set -uo pipefail
out=gpurun_out/r4n; mkdir -p "$out"
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --factors 9 --steps 30 --family-steps 0 > "$out/bf16_$i.json" 2>/dev/null
  python bench.py --no-cpu-baseline --factors 9 --steps 30 --family-steps 0 --fp8 > "$out/fp8_$i.json" 2>/dev/null
  python bench.py --no-cpu-baseline --factors 9 --steps 30 --family-steps 0 --fp8-all > "$out/fp8all_$i.json" 2>/dev/null
done
python - <<PY
import json,glob
for tag in ("bf16","fp8","fp8all"):
    v=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("$out/%s_[0-9].json"%tag))]
    print(tag, ["%.3f ms @ %.0f MHz loss %.4f"%(d["ms_per_step"], d["clock_mhz"], d["final_loss"]) for d in v])
PY
