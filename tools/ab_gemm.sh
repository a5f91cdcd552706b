set -e
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['final_loss'], d['graph'])"; }
run nt_all A=1
run nt_ge_20MB KVQ_GEMM_NT_MIN_MB=20
run nt_ge_100MB KVQ_GEMM_NT_MIN_MB=100
run nt_none KVQ_GEMM_NT_MIN_MB=100000
run nt_all2 A=1
run nt_ge_20MB2 KVQ_GEMM_NT_MIN_MB=20
