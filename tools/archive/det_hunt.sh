#!/usr/bin/env bash
# one quick look at a box: its clock under the bench and whether the final loss is the canonical one (7.664987683296204 after 35 steps);
# if not, the library without the whole-line attention rows (lib/prev, built by hand) on the same box, twice each
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['final_loss'], '%.3f ms @ %.0f MHz' % (d['ms_per_step'], d['clock_mhz']))"; }
python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "current " | tee /tmp/first.txt
if ! grep -q "7.664987683296204" /tmp/first.txt; then
  for i in 1 2; do
    KVQ_LIB_PATH=$PWD/kindergarten-vq-vae_amd/lib/prev/libkvq.so python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "previous"
    python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "current "
    KVQ_GRAPH=0 python bench.py --no-cpu-baseline --steps 30 --family-steps 0 2>/dev/null | p "current eager"
  done
  python tools/attn_repeat.py 2>/dev/null
fi
