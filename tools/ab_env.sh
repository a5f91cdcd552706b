#!/usr/bin/env bash
# interleaved bench.py runs under different environments on one box:  tools/ab_env.sh <rounds> "name|VAR=val VAR2=val" ...
n="$1"; shift
for i in $(seq "$n"); do
  for spec in "$@"; do
    name="${spec%%|*}"; envs="${spec#*|}"
    env $envs python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'])"
  done
done
