#!/usr/bin/env bash
# L1 -> L2 traffic of the attention kernels, row chunks per lane (KVQ_ATTN_COAL=0 KVQ_ATTN_STC=0) against whole lines (default):
# separate --pmc passes (one counter family each), run ON the GPU box:  gpurun -- bash tools/run_attn_shape_pmc.sh
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_attn_pmc; rm -rf "$out"; mkdir -p "$out"
for shape in chunks lines; do
  if [ $shape = chunks ]; then export KVQ_ATTN_COAL=0 KVQ_ATTN_STC=0; else export KVQ_ATTN_COAL=1 KVQ_ATTN_STC=1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${shape}_trace" -- python tools/attn_shape_pmc.py > "$out/${shape}_trace.log" 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$out/${shape}_l1" -- python tools/attn_shape_pmc.py > "$out/${shape}_l1.log" 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d "$out/${shape}_l1w" -- python tools/attn_shape_pmc.py > "$out/${shape}_l1w.log" 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/${shape}_fetch" -- python tools/attn_shape_pmc.py > "$out/${shape}_fetch.log" 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/${shape}_write" -- python tools/attn_shape_pmc.py > "$out/${shape}_write.log" 2>&1
  echo "$shape done"
done
python - <<'PY'
import csv, glob, collections
out = "gpurun_out/r04_attn_pmc"
rows = []
for shape in ("chunks", "lines"):
    dur = {}
    for f in glob.glob(f"{out}/{shape}_trace/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "attn_" in r["Name"]:
                dur[r["Name"].split("<")[0].replace("void ", "")] = float(r["AverageNs"]) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ("l1", "l1w", "fetch", "write"):
        for f in glob.glob(f"{out}/{shape}_{p}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "attn_" in r["Kernel_Name"]:
                    agg[r["Kernel_Name"].split("<")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        m = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else float("nan")
        print(f"{shape:6s} {k:28s} {dur.get(k, float('nan')):6.1f} us | L1->L2 read req {m('TCP_TCC_READ_REQ_sum')/1e6:6.2f} M, write req {m('TCP_TCC_WRITE_REQ_sum')/1e6:6.2f} M, "
              f"L1 accesses {m('TCP_TOTAL_CACHE_ACCESSES_sum')/1e6:6.2f} M, pending-stall {m('TCP_PENDING_STALL_CYCLES_sum')/1e6:6.1f} Mcyc, "
              f"read latency {m('TCP_TCC_READ_REQ_LATENCY_sum')/max(m('TCP_TCC_READ_REQ_sum'),1):5.0f} cyc | fabric {(2*m('FETCH_SIZE')+m('WRITE_SIZE'))*1024/1e6:6.1f} MB")
PY
