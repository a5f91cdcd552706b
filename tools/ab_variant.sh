#!/usr/bin/env bash
# same-box A/B of the product library against a variant build (tools/build_variant.sh <name> ...):  gpurun -- bash tools/ab_variant.sh <name> [rounds] [bench flags...]
name="$1"; rounds="${2:-3}"; shift; shift || true
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['final_loss'], '%.3f ms @ %.0f MHz' % (d['ms_per_step'], d['clock_mhz']))"; }
for i in $(seq 1 "$rounds"); do
  python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | p "product       "
  KVQ_LIB_PATH=$PWD/kindergarten-vq-vae_amd/lib/$name/libkvq.so python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | p "variant $name"
done
