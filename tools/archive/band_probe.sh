for b in 1 2 3 4 6 8; do echo "band=$b"; KVQ_GEMM_BAND=$b KVQ_PROBE=bigfwd timeout -k 10 120 python tools/gemm2_probe_cold.py 3 2>&1 | grep -v amdgpu; done
