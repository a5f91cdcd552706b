#!/usr/bin/env bash
# memory-system counters of three GEMM kernels (separate --pmc passes), run ON the GPU box through gpurun
set -euo pipefail
out=gpurun_out/gemm_diag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python tools/gemm2_diag.py > "$out/$name.log" 2>&1; echo "$name done"; }
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
# TCC_EA0_* / TCC_TAG_STALL in ONE pass ask for more TCC counters than a pass holds ("Could not construct profile cfg failed with
# error code 38: Request exceeds the capabilities of the hardware to collect", gpurun_out/gemm_diag/ea.log of round 2 -- the
# abort was that, not a tool crash): one or two TCC counters per pass, as for FETCH_SIZE / WRITE_SIZE
run ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run ea_stall TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
python - <<'PY'
import csv, glob, collections
for p in ("lat", "tlb", "ea_rd", "ea_wr", "ea_stall"):
    f = glob.glob(f"gpurun_out/gemm_diag/{p}/*/*counter_collection.csv")
    if not f:
        print(p, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "gemm2" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("Cfg<")[1].split(">")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(p, k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
