set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'])"; }
D="768x768:128x192;768x2304:128x192;768x3072:128x192;3072x768:256x192"
run both_own A=1
run neither KVQ_OWN_DGRAD="$D"
run crosskv_only KVQ_OWN_DGRAD="$D;768x18432:128x192"
run both_own2 A=1
run neither2 KVQ_OWN_DGRAD="$D"
run crosskv_only2 KVQ_OWN_DGRAD="$D;768x18432:128x192"
