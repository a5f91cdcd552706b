set -e
run() { name=$1; shift; env KVQ_DP_SINGLE_RANK=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['exposed_comm_ms_per_step'], d['dist_backend'])"; }
run b32 --bucket-mib 32
run b64 --bucket-mib 64
run b128 --bucket-mib 128
run b256 --bucket-mib 256
run b1024 --bucket-mib 1024
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', round(d['ms_per_step'],3))"
