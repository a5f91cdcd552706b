#!/usr/bin/env bash
# run-to-run determinism of the default bench (final loss of 35 steps), current library vs lib/prev:  gpurun -- bash tools/det_check.sh
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['final_loss'], '%.3f ms @ %.0f MHz' % (d['ms_per_step'], d['clock_mhz']))"; }
python tools/attn_repeat.py 2>/dev/null
KVQ_LIB_PATH=$PWD/kindergarten-vq-vae_amd/lib/prev/libkvq.so python tools/attn_repeat.py 2>/dev/null | sed 's/^/HEAD library: /'
for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "current        "
  KVQ_ATTN_STC=0 python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "current STC=0  "
  KVQ_ATTN_STC=0 KVQ_ATTN_COAL=0 python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "current both=0 "
  KVQ_LIB_PATH=$PWD/kindergarten-vq-vae_amd/lib/prev/libkvq.so python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "HEAD library   "
done
