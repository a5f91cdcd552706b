#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py: the last two steps (between Adam launches).
usage: step_breakdown.py <dir with *_kernel_trace.csv> [rows]"""
import collections
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)   # (a re-run merges next to older files)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ad = [i for i, n in enumerate(names) if "adam_kernel<1>" in n]
a, b = ad[-3], ad[-1]
tot = collections.defaultdict(lambda: [0, 0])
for r in rows[a + 1:b + 1]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"][:78]
    tot[k][0] += d
    tot[k][1] += 1
T = sum(v[0] for v in tot.values()) / 2
walls = sorted((int(rows[ad[j + 1]]["End_Timestamp"]) - int(rows[ad[j]]["End_Timestamp"])) / 1e6 for j in range(max(0, len(ad) - 8), len(ad) - 1))
wall = walls[len(walls) // 2]               # median of the last steps (the profiler's buffer flushes land in some of them)
gemm = sum(v[0] for k, v in tot.items() if "gemm2" in k or "gemm3" in k or "Cijk" in k) / 2e6
own = sum(v[0] for k, v in tot.items() if "gemm2" in k or "gemm3" in k) / 2e6
print(f"kernel time {T / 1e6:.3f} ms/step, wall {wall:.3f} ms/step, launches {sum(v[1] for v in tot.values()) // 2}, GEMM {gemm:.3f} ms (own {own:.3f})")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{v[0] / 2e6:7.3f} ms {v[1] // 2:4d} x {v[0] / v[1] / 1000:8.1f} us  {k}")
