set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))"; }
run base A=1
run fwd_none KVQ_OWN_FWD=""
run fwd_ffn1 KVQ_OWN_FWD="768x768:128x256;3072x768:256x192"
run fwd_qkv KVQ_OWN_FWD="768x768:128x256;2304x768:256x192"
run fwd_ffn2 KVQ_OWN_FWD="768x768:128x256;768x3072:128x256"
run fwd_big KVQ_OWN_FWD="768x768:128x256;18432x768:256x256;30528x768:256x256"
run fwd_all KVQ_OWN_FWD="768x768:128x256;2304x768:256x192;3072x768:256x192;768x3072:128x256;18432x768:256x256;30528x768:256x256"
run dgrad_lm KVQ_OWN_DGRAD="768x768:128x256;768x2304:128x256;768x3072:128x256;3072x768:256x192;768x18432:128x256;768x30528:128x256"
run base2 A=1
