set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'], d['graph'])"; }
run late_adam KVQ_EARLY_ADAM=0
run early_128 KVQ_ADAM_BLOCKS_EXPERIMENT=128
run early_256 KVQ_ADAM_BLOCKS_EXPERIMENT=256
run early_512 KVQ_ADAM_BLOCKS_EXPERIMENT=512
run early_1024 KVQ_ADAM_BLOCKS_EXPERIMENT=1024
run late_adam2 KVQ_EARLY_ADAM=0
