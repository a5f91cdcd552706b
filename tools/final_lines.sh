#!/usr/bin/env bash
# the tracked bench lines of a round, all from one box and one build:  tools/final_lines.sh <tag>   ->  gpurun_out/<tag>_lines/*.json
tag="${1:-r03}"; out="gpurun_out/${tag}_lines"; mkdir -p "$out"
python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
python bench.py --no-cpu-baseline --codes 8192 > "$out/bench_line_k8192.json" 2> "$out/k8192.err"
python bench.py --no-cpu-baseline --factors 9 > "$out/bench_line_9factors.json" 2> "$out/9f.err"
python bench.py --no-cpu-baseline --fp8 --factors 9 > "$out/bench_line_fp8_9factors.json" 2> "$out/fp8.err"
KVQ_DP_SINGLE_RANK=1 python bench.py --no-cpu-baseline > "$out/bench_line_rccl_one_rank.json" 2> "$out/rccl1.err"
KVQ_DIST_BACKEND=gloo python bench.py --no-cpu-baseline --gpus 2 > "$out/bench_line_2ranks_gloo_one_gpu.json" 2> "$out/gloo2.err"
for f in "$out"/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1].split('/')[-1]:44s} {d['ms_per_step']:.3f} ms/step {d['value']:.0f} sent/s n_gpus {d['n_gpus']} backend {d['dist_backend']} graph {d['graph']} loss {d['final_loss']:.4f} vq {d['roofline']['avg_launch_us']:.1f} us frac {d['roofline']['frac']:.3f} traffic {d['roofline']['traffic']}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
