#!/usr/bin/env bash
# memory-system counters of three GEMM kernels (separate --pmc passes), run ON the GPU box through gpurun
set -euo pipefail
out=gpurun_out/gemm_diag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python tools/gemm2_diag.py > "$out/$name.log" 2>&1; echo "$name done"; }
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
# (a third pass with TCC_EA0_* / TCC_TAG_STALL aborted rocprofv3 on this image -- signal 6 inside the tool -- and is not run)
python - <<'PY'
import csv, glob, collections
for p in ("lat", "tlb"):
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
