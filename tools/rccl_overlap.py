#!/usr/bin/env python3
"""What the gradient exchange costs the kernels it runs beside, from ONE rocprofv3 --kernel-trace of a
KVQ_DP_SINGLE_RANK=1 (or multi-rank) bench.py run:

    python tools/rccl_overlap.py <dir with *_kernel_trace.csv> [<dir of a trace WITHOUT the exchange>]

For the last traced steps: every libkvq.so kernel launch is classed "beside" when a collective kernel (names containing
Reduce / ncclDevKernel / rccl) was running during more than half of it, "alone" otherwise; per kernel name: launches and mean
duration in each class, and the step's stretch = sum over beside-launches of (duration - that name's alone mean).  Also: the
collective kernels' own time, how much of it the compute queue covered, and the wall step."""
import collections
import csv
import glob
import os
import sys


def load(d):
    f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
            for r in csv.DictReader(open(f))]
    rows.sort()
    return rows


def is_coll(n):
    return any(t in n for t in ("Reduce", "ncclDevKernel", "rccl", "AllGather", "Broadcast")) and "reduce_batch" not in n


def steps_of(rows, last=6):
    # a step ends with the LAST adam_kernel<1> launch of the step (the exchange splits Adam in two: head and tail)
    ends = [i for i, r in enumerate(rows) if "step_state_advance" in r[2]]
    return [(rows[ends[j]][0], rows[ends[j + 1]][0]) for j in range(max(0, len(ends) - 1 - last), len(ends) - 1)]


def short(n):
    n = n.replace("void ", "").replace("kvq::", "").replace("(anonymous namespace)::", "")
    return n[:86]


def main():
    rows = load(sys.argv[1])
    spans = steps_of(rows)
    if not spans:
        raise SystemExit("no steps found (step_state_advance_kernel marks the step boundary)")
    coll_t = comp = 0
    by = collections.defaultdict(lambda: {"alone": [], "beside": []})
    walls = []
    coll_names = collections.Counter()
    coll_wg = collections.Counter()
    for t0, t1 in spans:
        walls.append((t1 - t0) / 1e6)
        ks = [r for r in rows if t0 <= r[0] < t1]
        cs = [r for r in ks if is_coll(r[2])]
        for c in cs:
            coll_t += c[1] - c[0]
            coll_names[short(c[2])] += 1
            coll_wg[c[4]] += 1
        for r in ks:
            if is_coll(r[2]):
                continue
            d = r[1] - r[0]
            ov = sum(max(0, min(r[1], c[1]) - max(r[0], c[0])) for c in cs)
            by[short(r[2])]["beside" if ov > d / 2 else "alone"].append(d)
            comp += ov
    n = len(spans)
    print(f"{n} steps, wall {sum(walls) / n:.3f} ms/step (median {sorted(walls)[n // 2]:.3f}); collective kernels {coll_t / n / 1e6:.3f} ms/step "
          f"({dict(coll_names)}; workgroups per launch {dict(coll_wg)}), covered by compute-queue kernels {comp / n / 1e6:.3f} ms/step")
    out = []
    stretch = 0.0
    for k, v in by.items():
        if not v["beside"]:
            continue
        al = sum(v["alone"]) / len(v["alone"]) if v["alone"] else None
        be = sum(v["beside"]) / len(v["beside"])
        st = (be - al) * len(v["beside"]) / n if al else 0.0
        stretch += st
        out.append((st, k, len(v["alone"]) / n, al, len(v["beside"]) / n, be))
    print(f"{'stretch ms/step':>15s} {'alone x':>8s} {'us':>8s} {'beside x':>9s} {'us':>8s}  kernel")
    for st, k, na, al, nb, be in sorted(out, reverse=True):
        print(f"{st / 1e6:15.3f} {na:8.1f} {(al or 0) / 1e3:8.1f} {nb:9.1f} {be / 1e3:8.1f}  {k}")
    print(f"sum of stretches {stretch / 1e6:.3f} ms/step (kernels with no launch outside a collective are not priced)")
    if len(sys.argv) > 2:
        base = load(sys.argv[2])
        bs = steps_of(base)
        bw = [(b - a) / 1e6 for a, b in bs]
        print(f"trace without the exchange: wall {sum(bw) / len(bw):.3f} ms/step")
        ref = collections.defaultdict(list)
        for t0, t1 in bs:
            for r in base:
                if t0 <= r[0] < t1:
                    ref[short(r[2])].append(r[1] - r[0])
        tot = 0.0
        print("against the same kernel's mean in the trace without the exchange (all launches):")
        res = []
        for k, v in by.items():
            if k in ref:
                mine = (sum(v["alone"]) + sum(v["beside"])) / n
                theirs = sum(ref[k]) / len(bs)
                res.append((mine - theirs, k, mine, theirs))
                tot += mine - theirs
        for dlt, k, mine, theirs in sorted(res, reverse=True)[:14]:
            print(f"{dlt / 1e6:+9.3f} ms/step  {mine / 1e6:7.3f} vs {theirs / 1e6:7.3f}  {k}")
        print(f"total {tot / 1e6:+.3f} ms/step")


if __name__ == "__main__":
    main()
