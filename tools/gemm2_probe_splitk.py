#!/usr/bin/env python3
"""256 x 256 tiles with split-K on the N = 768 products (96 tiles -> 192 / 288 workgroups).  Timing probe with the
existing kernels: the K slices as a grouped launch writing bf16 partial outputs, plus the cheapest possible combine (one elementwise
add over bf16 partials -- an exact combine needs f32 partials: twice the bytes).  Against the engine's choice (128 x 192, no split)."""
import os
import sys

sys.argv = [sys.argv[0], "none"] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm2_probe as g  # noqa: E402
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

T = g.T
for layout, K in (("nt", 3072), ("nn", 3072), ("nn", 2304)):
    N = 768
    if layout == "nt":
        a, b = g.rnd(T, K), g.rnd(N, K)
    else:
        a, b = g.rnd(T, K), g.rnd(K, N)
    out = torch.empty((T, N), device="cuda", dtype=torch.bfloat16)
    base = lambda: nnops.gemm(a, b, layout, out=out, tile="128x192")
    res = {"128x192 no split": base}
    for split in (2, 3):
        ks = K // split
        parts = [torch.empty((T, N), device="cuda", dtype=torch.bfloat16) for _ in range(split)]
        probs = []
        for i in range(split):
            ai = a[:, i * ks:(i + 1) * ks]
            bi = b[:, i * ks:(i + 1) * ks] if layout == "nt" else b[i * ks:(i + 1) * ks]
            probs.append(nnops.gemm_problem(ai, bi, parts[i], layout))
        for tile in ("256x256", "256x192"):
            def run(probs=probs, parts=parts, tile=tile):
                nnops.gemm_grouped(probs, layout, tile)
                acc = parts[0]
                for p in parts[1:]:
                    acc = torch.add(acc, p, out=out)
            def run_gemm_only(probs=probs, tile=tile):
                nnops.gemm_grouped(probs, layout, tile)
            res[f"split {split} {tile} + add"] = run
            res[f"split {split} {tile} GEMM only"] = run_gemm_only
    line = f"{layout} [8192, 768] x {K}: "
    tm = {k: [] for k in res}
    for _ in range(g.rounds):
        for k, f in res.items():
            tm[k].append(g.bench(f))
    for k, v in tm.items():
        line += f"{k} {sorted(v)[len(v) // 2]:.1f} us | "
    print(line, flush=True)
