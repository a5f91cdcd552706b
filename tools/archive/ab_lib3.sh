#!/usr/bin/env bash
# interleaved bench.py runs of several builds of the library on one box:  tools/ab_lib3.sh <rounds> name=path ...
n="$1"; shift
for i in $(seq "$n"); do
  for kv in "$@"; do
    name="${kv%%=*}"; path="${kv#*=}"
    if [ -n "$path" ]; then export KVQ_LIB_PATH="$PWD/$path"; else unset KVQ_LIB_PATH; fi
    python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'])"
  done
done
