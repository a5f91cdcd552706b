set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))"; }
run base A=1
run wg_per_layer KVQ_WG_PAIR=0
run fwd_no_ffn2 KVQ_OWN_FWD="768x768:128x192"
run base2 A=1
