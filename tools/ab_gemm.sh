set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))"; }
run base A=1
run b2048 KVQ_ADAM_BLOCKS_EXPERIMENT=2048
run b4096 KVQ_ADAM_BLOCKS_EXPERIMENT=4096
run b8192 KVQ_ADAM_BLOCKS_EXPERIMENT=8192
run b16384 KVQ_ADAM_BLOCKS_EXPERIMENT=16384
run base2 A=1
