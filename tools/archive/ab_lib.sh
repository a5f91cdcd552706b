#!/usr/bin/env bash
# same-box A/B of two builds of the library (lib/prev/libkvq.so = the previous commit's sources, built by hand):  gpurun -- bash tools/ab_lib.sh
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['final_loss'], '%.3f ms @ %.0f MHz' % (d['ms_per_step'], d['clock_mhz']))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "current "
  KVQ_LIB_PATH=$PWD/kindergarten-vq-vae_amd/lib/prev/libkvq.so python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "previous"
done
