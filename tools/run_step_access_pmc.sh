#!/usr/bin/env bash
# vector-cache accesses and L1 -> L2 requests of EVERY kernel of one training step (eager launches), to find access shapes that
# cost more cache accesses per byte than the rest:  gpurun -- bash tools/run_step_access_pmc.sh
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_step_access; rm -rf "$out"; mkdir -p "$out"
export KVQ_GRAPH=0
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d "$out/l1" -- python bench.py --no-cpu-baseline --steps 3 --warmup 2 --family-steps 0 > "$out/l1.log" 2>&1; echo "l1 rc=$?"
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r04_step_access/l1/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["_dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in agg.items():
    n = len(c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    acc = sum(c["TCP_TOTAL_CACHE_ACCESSES_sum"]) / n
    rd = sum(c["TCP_TCC_READ_REQ_sum"]) / n
    wr = sum(c["TCP_TCC_WRITE_REQ_sum"]) / n
    rows.append((sum(c["_dur"]) / 3 / 3, k, n // 5 // 3 if n >= 15 else n, acc, rd, wr, sum(c["_dur"]) / len(c["_dur"])))
print("kernel | launches | avg us | cache accesses M | L1->L2 read req M | write req M | accesses per read+write request")
for tot, k, n, acc, rd, wr, d in sorted(rows, reverse=True)[:28]:
    print(f"{k:70s} | {d:7.1f} us | acc {acc/1e6:7.2f} M | rd {rd/1e6:6.2f} M | wr {wr/1e6:6.2f} M | {acc/max(rd+wr,1):5.2f}")
PY
