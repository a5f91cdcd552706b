set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'vq us', round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],3))"; }
run tt2 A=1
run tt4 KVQ_VQ_TT4_EXPERIMENT=1
run tt2b A=1
run tt4b KVQ_VQ_TT4_EXPERIMENT=1
