#!/usr/bin/env bash
# same-box A/B of lib/ref/libkvq.so (tools/build_ref.sh <commit>) against lib/libkvq.so: interleaved bench.py runs
n="${1:-3}"
for i in $(seq "$n"); do
  for v in ref new; do
    if [ "$v" = ref ]; then export KVQ_LIB_PATH="$PWD/kindergarten-vq-vae_amd/lib/ref/libkvq.so"; else unset KVQ_LIB_PATH; fi
    python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), d['final_loss'])"
  done
done
