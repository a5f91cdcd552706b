mkdir -p gpurun_out/r3c
python -m pytest tests/test_gemm2_gpu.py -x -q -m gpu > gpurun_out/r3c/t_gemm.log 2>&1; tail -3 gpurun_out/r3c/t_gemm.log
python -m pytest tests/test_engine_base_shapes_gpu.py -x -q -m gpu > gpurun_out/r3c/t_base.log 2>&1; tail -3 gpurun_out/r3c/t_base.log
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r3c/b_new_$i.log 2>&1; python - <<PY
import json;d=json.loads(open("gpurun_out/r3c/b_new_$i.log").read().strip().splitlines()[-1]);print("new",d["ms_per_step"],d["final_loss"])
PY
KVQ_OWN_FWD="768x768:128x192;768x3072:128x192;2304x768:128x192" KVQ_OWN_GELU="3072x768:256x192" python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r3c/b_old_$i.log 2>&1; python - <<PY
import json;d=json.loads(open("gpurun_out/r3c/b_old_$i.log").read().strip().splitlines()[-1]);print("old",d["ms_per_step"],d["final_loss"])
PY
done
