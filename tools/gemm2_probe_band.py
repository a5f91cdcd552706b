#!/usr/bin/env python3
"""Tile order of wide outputs: row bands (KVQ_GEMM_BAND = r > 0) vs column bands (= -c), one-tile and persistent kernels,
interleaved rounds in one process.  usage: gemm2_probe_band.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda"


def bench(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rnd(*s):
    return torch.randn(s, device=dev).to(torch.bfloat16)


def run(M, N, K, tile, bands):
    a, b, bias = rnd(M, K), rnd(N, K), rnd(N)
    ref = torch.addmm(bias, a, b.t())
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    fns = {"lib": lambda: torch.addmm(bias, a, b.t())}
    for d, nm in (("0", "tile"), ("2", "persist")):
        for bd in bands:
            def f(d=d, bd=bd):
                os.environ["KVQ_GEMM_BAND"] = str(bd)
                return nnops.gemm(a, b, "nt", bias=bias, out=out, tile=tile + ("p" if d == "2" else ""))
            out.fill_(float("nan"))
            o = f()
            torch.cuda.synchronize()
            rel = (o.float() - ref.float()).norm().item() / ref.float().norm().item()
            assert rel < 5e-3, (M, N, K, tile, nm, bd, rel)
            fns[f"{nm}/band{bd}"] = f
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            res[k].append(bench(f))
    fl = 2.0 * M * N * K
    print(f"nt M={M} N={N} K={K} tile {tile}:")
    for k, v in res.items():
        m = sorted(v)[len(v) // 2]
        print(f"    {k:22s} {m:7.1f} us {fl / m / 1e6:5.0f} TF", flush=True)
    os.environ["KVQ_GEMM_BAND"] = "0"


run(8192, 30528, 768, "256x256", [2, 4, -4, -6, -8, -15])
run(8192, 18432, 768, "256x256", [2, 4, -4, -6, -9])
run(8192, 3072, 768, "256x192", [1, 2, -4, -8, -16])
run(8192, 2304, 768, "128x192", [1, 2, -4, -6, -12])
