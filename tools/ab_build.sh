# A/B of two builds of csrc/kvq_nn.hip on ONE box: bench with the shipped library, rebuild with the given -D flags, bench, rebuild plain, bench
set -e
bench() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3))"; }
bench shipped
touch kindergarten-vq-vae_amd/csrc/kvq_nn.hip
bash kindergarten-vq-vae_amd/build.sh "$@" > /dev/null
bench "variant($*)"
touch kindergarten-vq-vae_amd/csrc/kvq_nn.hip
bash kindergarten-vq-vae_amd/build.sh > /dev/null
bench plain
touch kindergarten-vq-vae_amd/csrc/kvq_nn.hip
bash kindergarten-vq-vae_amd/build.sh "$@" > /dev/null
bench "variant($*)"
