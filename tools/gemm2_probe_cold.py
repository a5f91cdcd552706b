#!/usr/bin/env python3
"""Own GEMM tiles vs the library with COLD operands: every call works on a different (A, B, C) set out of a pool larger
than the 256 MB Infinity Cache, as in the training step, where each GEMM's weights come from HBM and its output goes to
memory nobody has touched for 19 ms.  (tools/gemm2_probe.py re-uses one set: everything it touches is cache-resident.)
usage: gemm2_probe_cold.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

T = 8192
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda"


def rnd(*s):
    return torch.randn(s, device=dev).to(torch.bfloat16)


def run(name, layout, M, N, K, tiles, pool_bytes=1.5e9):
    per = 2 * (M * K + N * K + M * N)
    R = max(2, int(pool_bytes // per))
    sets = []
    for _ in range(R):
        if layout == "nt":
            a, b = rnd(M, K), rnd(N, K)
        elif layout == "nn":
            a, b = rnd(M, K), rnd(K, N)
        else:
            a, b = rnd(K, M), rnd(K, N)
        sets.append((a, b, torch.empty((M, N), device=dev, dtype=torch.bfloat16)))

    def lib(a, b, c):
        if layout == "nt":
            torch.mm(a, b.t(), out=c)
        elif layout == "nn":
            torch.mm(a, b, out=c)
        else:
            torch.mm(a.t(), b, out=c)

    fns = {"lib": lib}
    for t in tiles:
        fns[t] = (lambda a, b, c, t=t: nnops.gemm(a, b, layout, out=c, tile=t))
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            f(*sets[0])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for s in sets:
                f(*s)
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / R * 1e3)
    fl = 2.0 * M * N * K
    line = f"{name:6s} {layout} M={M:6d} N={N:6d} K={K:6d} sets={R:3d}: "
    for k, v in res.items():
        m = sorted(v)[len(v) // 2]
        line += f"{k} {m:7.1f} us {fl / m / 1e6:6.0f} TF | "
    print(line, flush=True)


SHAPES = os.environ.get("KVQ_PROBE", "layer")
for n, k, tl in [] if SHAPES != "layer" else [(768, 768, ["128x192", "128x256"]), (2304, 768, ["128x192", "256x192", "256x256"]),
                 (3072, 768, ["256x192", "256x256", "128x256"]), (768, 3072, ["128x192", "128x256"])]:
    run("fwd", "nt", T, n, k, tl)
for n, k, tl in [] if SHAPES != "layer" else [(768, 768, ["128x192"]), (768, 2304, ["128x192"]), (768, 3072, ["128x192"]), (3072, 768, ["256x192", "256x256"])]:
    run("dgrad", "nn", T, n, k, tl)
if SHAPES == "big":
    for n in (18432, 30528):
        run("fwd", "nt", T, n, 768, ["256x256", "256x192"], pool_bytes=2.5e9)
    for k in (18432, 30528):
        run("dgrad", "nn", T, 768, k, ["128x192", "128x256"], pool_bytes=2.5e9)
    for m in (18432, 30528):
        run("wgrad", "tn", m, 768, T, ["256x256", "256x192"], pool_bytes=2.5e9)
if SHAPES == "bigfwd":
    for n in (18432, 30528):
        run("fwd", "nt", T, n, 768, ["256x256", "256x192"], pool_bytes=2.5e9)
