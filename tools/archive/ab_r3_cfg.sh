#!/usr/bin/env bash
# the other configurations of BASELINE.json on one box, same run:  tools/ab_r3_cfg.sh <outdir>
out="gpurun_out/${1:-r3cfg}"; mkdir -p "$out"
run() { name="$1"; shift; python bench.py --no-cpu-baseline --steps 20 "$@" > "$out/$name.json" 2> "$out/$name.err"; python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} sentences/s  loss {d['final_loss']:.4f}  vq {d['roofline']['avg_launch_us']:.1f} us frac {d['roofline']['frac']:.3f}")
PY
}
run bf16_default
run bf16_9factors --factors 9
run fp8_9factors --fp8 --factors 9
run fp8_1factor --fp8
run k8192 --codes 8192
run bf16_default_again
