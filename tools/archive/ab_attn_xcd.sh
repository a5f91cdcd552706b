#!/usr/bin/env bash
# same-box A/B of the XCD-aware (sentence, head) order of the attention kernels:  gpurun -- bash tools/ab_attn_xcd.sh
# (the KVQ_ATTN_XCD switch was removed after this measurement: profiles/r04_rejected.md)
set -uo pipefail
out=gpurun_out/r4j; mkdir -p "$out"
for i in 1 2 3; do
  KVQ_ATTN_XCD=0 python tools/attn_bias_probe.py 3 > "$out/probe_off_$i.log" 2>&1
  KVQ_ATTN_XCD=1 python tools/attn_bias_probe.py 3 > "$out/probe_on_$i.log" 2>&1
done
grep -h "no partials\|+ q/k/v" "$out"/probe_off_*.log | sed 's/^/off /'
grep -h "no partials\|+ q/k/v" "$out"/probe_on_*.log | sed 's/^/on  /'
for i in 1 2 3; do
  KVQ_ATTN_XCD=0 python bench.py --no-cpu-baseline --steps 30 --family-steps 0 > "$out/b_off_$i.json" 2>/dev/null
  KVQ_ATTN_XCD=1 python bench.py --no-cpu-baseline --steps 30 --family-steps 0 > "$out/b_on_$i.json" 2>/dev/null
done
python - <<PY
import json,glob
for tag in ("off","on"):
    v=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("$out/b_%s_*.json"%tag))]
    print(tag, ["%.3f ms @ %.0f MHz"%(d["ms_per_step"], d["clock_mhz"]) for d in v])
PY
