set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))"; }
run base A=1
run fused_gelu KVQ_OWN_GELU="3072x768:256x192"
run fused_dgelu KVQ_OWN_DGELU="3072x768:256x192"
run fused_both KVQ_OWN_GELU="3072x768:256x192" KVQ_OWN_DGELU="3072x768:256x192"
run fused_both_128 KVQ_OWN_GELU="3072x768:128x256" KVQ_OWN_DGELU="3072x768:128x256"
run base2 A=1
