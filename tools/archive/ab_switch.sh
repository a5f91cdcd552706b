#!/usr/bin/env bash
# same-box A/B of a 0/1 environment switch of the library:  gpurun -- bash tools/ab_switch.sh KVQ_ATTN_COAL [probe.py ...]
# interleaved: the attention probe (cold buffers) and bench.py, three rounds each
set -uo pipefail
var="$1"; out="gpurun_out/ab_$var"; mkdir -p "$out"
for i in 1 2 3; do
  for v in 0 1; do env "$var=$v" python tools/attn_bias_probe.py 3 > "$out/probe_${v}_$i.log" 2>&1; done
done
for v in 0 1; do echo "== $var=$v"; grep -h "attn_bwd\|attn_fwd" "$out"/probe_${v}_*.log | sort | awk '{print}' ; done
for i in 1 2 3; do
  for v in 0 1; do env "$var=$v" python bench.py --no-cpu-baseline --steps 30 --family-steps 0 > "$out/b_${v}_$i.json" 2>/dev/null; done
done
python - <<PY
import json,glob
for v in (0,1):
    d=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("$out/b_%d_*.json"%v))]
    print("$var=%d"%v, ["%.3f ms @ %.0f MHz (loss %.5f)"%(x["ms_per_step"], x["clock_mhz"], x["final_loss"]) for x in d])
PY
