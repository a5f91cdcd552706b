set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'])"; }
run fused_lmce A=1
run separate KVQ_OWN_LMCE=0
run fused_lmce2 A=1
run separate2 KVQ_OWN_LMCE=0
