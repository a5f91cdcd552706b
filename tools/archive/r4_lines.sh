#!/usr/bin/env bash
# Round-4 bench lines and step traces for profiles/ -- run ON the GPU box:  gpurun -- bash tools/r4_lines.sh
set -uo pipefail
out=gpurun_out/r4g
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
python bench.py > "$out/bench_line.json" 2> "$out/bench.err"; echo "default rc=$?"
python bench.py --bagon --no-cpu-baseline > "$out/bench_line_bagon.json" 2> "$out/bench_bagon.err"; echo "bagon rc=$?"
python bench.py --seq-len 12 --batch 128 --no-cpu-baseline --steps 50 > "$out/bench_line_s12_b128.json" 2> "$out/s12.err"; echo "s12 rc=$?"
KVQ_OWN_GEMM=0 python bench.py --seq-len 12 --batch 128 --no-cpu-baseline --steps 50 > "$out/bench_line_s12_b128_library_gemm.json" 2> "$out/s12lib.err"; echo "s12 lib rc=$?"
python bench.py --seq-len 12 --batch 64 --no-cpu-baseline --steps 50 > "$out/bench_line_s12_b64.json" 2> "$out/s12b64.err"; echo "s12 b64 rc=$?"
KVQ_OWN_GEMM=0 python bench.py --seq-len 12 --batch 64 --no-cpu-baseline --steps 50 > "$out/bench_line_s12_b64_library_gemm.json" 2> "$out/s12b64lib.err"; echo "s12 b64 lib rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/step_bagon" -- python bench.py --bagon --no-cpu-baseline --steps 10 --warmup 3 --family-steps 0 > "$out/step_bagon.log" 2>&1
python tools/step_breakdown.py "$out/step_bagon" 60 > "$out/breakdown_bagon.txt"; head -3 "$out/breakdown_bagon.txt"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/step" -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 --family-steps 0 > "$out/step.log" 2>&1
python tools/step_breakdown.py "$out/step" 60 > "$out/breakdown.txt"; head -3 "$out/breakdown.txt"
