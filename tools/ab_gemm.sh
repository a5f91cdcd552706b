set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'], d['graph'])"; }
F="768x768:128x192;768x3072:128x192;2304x768:128x192"
run base A=1
run crosskv_own KVQ_OWN_FWD="$F;18432x768:256x256"
run lmhead_own KVQ_OWN_FWD="$F;30528x768:256x256"
run base2 A=1
run crosskv_own2 KVQ_OWN_FWD="$F;18432x768:256x256"
run lmhead_own2 KVQ_OWN_FWD="$F;30528x768:256x256"
