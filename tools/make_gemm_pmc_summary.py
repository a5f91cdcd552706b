#!/usr/bin/env python3
"""gpurun_out/<run>/ (tools/run_gemm_pmc.sh) -> profiles/<tag>_gemm_pmc.md: per GEMM shape of the step, own kernel and vendor
library: duration, MFMA-busy, HBM/fabric bytes (FETCH_SIZE doubled: calibrated in the same run by tools/ubench/fetch_calib.hip)."""
import collections
import csv
import glob
import json
import os
import sys

run, tag = sys.argv[1], sys.argv[2]
plan = json.load(open(f"{run}/gemm_pmc_plan.json"))


def groups(pass_name):
    """dispatches of one pass grouped by the marker kernels (FillFunctor<int>) that precede each plan entry"""
    f = max(glob.glob(f"{run}/{pass_name}/*/*counter_collection.csv"), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    by_disp = collections.OrderedDict()
    for r in rows:
        d = int(r["Dispatch_Id"])
        e = by_disp.setdefault(d, dict(name=r["Kernel_Name"], t=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, c={}))
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    out, cur, seen = [], None, 0
    for d, e in by_disp.items():
        if "FillFunctor<int>" in e["name"]:
            seen += 1
            cur = []
            out.append(cur)
            continue
        if cur is not None and ("gemm2_kernel" in e["name"] or "gemm2s_kernel" in e["name"] or "gemm3_kernel" in e["name"] or "Cijk" in e["name"]):
            cur.append(e)
    out = out[len(out) - len(plan):]          # the marker tensor's own creation is a fill kernel too
    assert len(out) == len(plan), (len(out), len(plan))
    return out


G = {p: groups(p) for p in ("sq", "fetch", "write", "grbm")}
calib = {}
for kind in ("fetch", "write"):
    f = sorted(glob.glob(f"{run}/calib_{kind}/*/*counter_collection.csv"), key=os.path.getmtime, reverse=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            calib.setdefault((kind, r["Kernel_Name"].split("(")[0][-40:]), []).append(float(r["Counter_Value"]) / 1024)
lines = [f"# {tag}: GEMM counters per shape of the training step (1x MI355X, rocprofv3 --pmc, ROCm 7.2)", "",
         "`tools/run_gemm_pmc.sh` (separate passes: SQ_*, FETCH_SIZE, WRITE_SIZE, GRBM+TCC) over `tools/gemm2_pmc.py`: every GEMM shape of the "
         "benchmarked step (8192 tokens, bert-base widths), the hand-written kernel (`csrc/kvq_gemm2.hip`, tile as the engine uses it) and the "
         "vendor library (torch.mm -> hipBLASLt) on the same operands, 4 launches each, averages below.", "",
         "* MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); TF = 2MNK / duration under the profiler.",
         "* fabric MB = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE counts half of the bytes read for 4-, 8- and 16-byte loads per lane "
         "alike, WRITE_SIZE is exact -- calibrated in the same run on 512 MiB buffers (`tools/ubench/fetch_calib.hip`): "
         + "; ".join(f"{k[1].strip()} {k[0]} {sum(v)/len(v):.0f} MiB" for k, v in sorted(calib.items()) if sum(v) > 0) + ".",
         "* alg MB = 2 (MK + NK + MN): every operand once.  Reads served by the 256 MiB Infinity Cache are counted in FETCH_SIZE, so a "
         "ratio above 1 is L2-level re-fetching (operand panels pulled into several of the 8 XCD L2s), not necessarily HBM traffic.", "",
         "| GEMM | layout | M | N | K | kernel | us | TF | MFMA util | wave WAIT_INST / WAIT_ANY / ACTIVE | LDS conflict / active | fabric MB | alg MB | ratio | L2 hit |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for i, p in enumerate(plan):
    if not p["who"]:
        continue
    fam = ("gemm2_kernel", "gemm2s_kernel", "gemm3_kernel") if p["who"].startswith("own") else ("Cijk",)       # (the other flavour's warm-up launch sits in this group too)
    sq, fe, wr, gr = ([e for e in G[k][i] if any(f in e["name"] for f in fam)] for k in ("sq", "fetch", "write", "grbm"))
    if not sq:
        continue
    # a library call may be several kernels per launch: sum per launch
    per = max(1, len(sq) // 4)
    # 4 timed launches per entry; a same-family warm-up launch of the NEXT entry may trail the group: drop it
    sq, fe, wr, gr = sq[:4 * per], fe[:4 * per], wr[:4 * per], gr[:4 * per]
    n = len(sq) / per
    t = sum(e["t"] for e in gr) / n
    c = lambda grp, k: sum(e["c"].get(k, 0.0) for e in grp) / n  # noqa: E731
    mf = c(sq, "SQ_VALU_MFMA_BUSY_CYCLES")
    gui = c(gr, "GRBM_GUI_ACTIVE")
    util = mf / (gui / 8 * 1024) if gui else float("nan")
    wc = c(sq, "SQ_WAVE_CYCLES")
    fab = (2 * c(fe, "FETCH_SIZE") + c(wr, "WRITE_SIZE")) * 1024 / 1e6
    alg = p["alg_bytes"] / 1e6
    hit = c(gr, "TCC_HIT_sum") / max(c(gr, "TCC_HIT_sum") + c(gr, "TCC_MISS_sum"), 1)
    kern = p["who"] if p["who"].startswith("own") else ("lib: " + ", ".join(sorted({e["name"].split("_BBS")[0].replace("Custom_", "") + " " +
                                                            (e["name"].split("MT")[1].split("_")[0] if "MT" in e["name"] else "") for e in sq})))
    lines.append(f"| {p['label']} | {p['layout']} | {p['M']} | {p['N']} | {p['K']} | {kern} | {t:.1f} | {p['flops']/t/1e6:.0f} | {util:.2f} | "
                 f"{c(sq,'SQ_WAIT_INST_ANY')/wc:.2f} / {c(sq,'SQ_WAIT_ANY')/wc:.2f} / {c(sq,'SQ_ACTIVE_INST_ANY')/wc:.2f} | "
                 f"{c(sq,'SQ_LDS_BANK_CONFLICT'):.3g} / {c(sq,'SQ_LDS_IDX_ACTIVE'):.3g} | {fab:.1f} | {alg:.1f} | {fab/alg:.2f} | {hit:.2f} |")
open(f"profiles/{tag}_gemm_pmc.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-45:]))
