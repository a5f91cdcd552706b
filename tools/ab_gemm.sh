set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'], d['graph'])"; }
run base A=1
run red_per_layer KVQ_RED_PAIR=0
run base2 A=1
run red_per_layer2 KVQ_RED_PAIR=0
