#!/usr/bin/env python3
"""gpurun_out/<run>/ (tools/run_block_pmc_r3.sh) -> profiles/<tag>_block_pmc.md: the HBM-streaming kernels on cold operands."""
import collections
import csv
import glob
import os
import sys

run, tag = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(max(glob.glob(f"{run}/trace/*/*_kernel_stats.csv"), key=os.path.getmtime))):
    dur[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for p, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for r in csv.DictReader(open(max(glob.glob(f"{run}/{p}/*/*counter_collection.csv"), key=os.path.getmtime))):
        if r["Counter_Name"] == c:
            cnt[r["Kernel_Name"].split("(")[0].replace("void ", "")][c].append(float(r["Counter_Value"]))
alg = {"kvq::drln_fwd_kernel<1, 3>": 50.3, "kvq::drln_bwd16_kernel<3>": 55.0, "kvq::attn_fwd_mfma_kernel": 50.3, "kvq::attn_bwd_mfma_kernel": 88.1,
       "__amd_rocclr_copyBuffer": 100.7}
lines = [f"# {tag}: the HBM-streaming kernels of the step on COLD operands (1x MI355X, rocprofv3, ROCm 7.2)", "",
         "`tools/run_block_pmc_r3.sh` over `tools/cold_stream_probe.py`: every kernel on a rotation of 24 buffer sets (0.3 - 1.2 GB per operand "
         "kind, larger than the 256 MiB Infinity Cache), as in the training step where a kernel's inputs were written milliseconds earlier. "
         "[8192, 768] bf16 rows (LayerNorm), 256 sentences x 12 heads x 32 tokens (attention), dropout 0.1; the device copy moves 50 MB in + "
         "50 MB out with the runtime's own kernel and is the yardstick for what a short streaming kernel reaches here.", "",
         "fabric MB = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (calibration: r03_gemm_pmc.md); TB/s = algorithmic MB / duration.", "",
         "| kernel | calls | avg us | algorithmic MB | fabric MB | algorithmic TB/s |", "|---|---|---|---|---|---|"]
for k, a in alg.items():
    if k not in dur:
        continue
    f = cnt[k]
    fab = (2 * sum(f["FETCH_SIZE"]) / max(len(f["FETCH_SIZE"]), 1) + sum(f["WRITE_SIZE"]) / max(len(f["WRITE_SIZE"]), 1)) * 1024 / 1e6
    lines.append(f"| `{k}` | {dur[k][0]} | {dur[k][1]:.1f} | {a:.1f} | {fab:.1f} | {a / dur[k][1]:.2f} |")
open(f"profiles/{tag}_block_pmc.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
