#!/usr/bin/env python3
"""One-tile-per-workgroup kernel vs the persistent tile loop (tile name + "p" = KVQ_GEMM_PERSISTENT) of csrc/kvq_gemm2.hip, NT
layout, interleaved rounds in one process; results checked against torch.  usage: gemm2_probe_persist.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda"
T = 8192


def bench(fn, iters=20):
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


def run(N, K, tiles, M=T):
    a, b, bias = rnd(M, K), rnd(N, K), rnd(N)
    ref = torch.addmm(bias, a, b.t())
    fns, res = {"lib": lambda: torch.addmm(bias, a, b.t())}, {"lib": []}
    for t in tiles:
        for d, nm in (("0", "tile"), ("2", "persist")):
            out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)

            def f(t=t, d=d, out=out):
                return nnops.gemm(a, b, "nt", bias=bias, out=out, tile=t + ("p" if d == "2" else ""))
            out.fill_(float("nan"))
            o = f()
            torch.cuda.synchronize()
            rel = (o.float() - ref.float()).norm().item() / ref.float().norm().item()
            if d == "0":
                base = o.clone()
            same = bool(torch.equal(o, base))
            assert rel < 5e-3, (N, K, t, nm, rel)
            fns[f"{t}/{nm}"] = f
            res[f"{t}/{nm}"] = []
            if d != "0":
                print(f"   {t} persistent == tile kernel bit for bit: {same}", flush=True)
    for _ in range(rounds):
        for k, f in fns.items():
            res[k].append(bench(f))
    fl = 2.0 * M * N * K
    line = f"nt M={M} N={N:6d} K={K:5d}: "
    for k, v in res.items():
        m = sorted(v)[len(v) // 2]
        line += f"{k} {m:7.1f} us {fl / m / 1e6:5.0f} TF | "
    print(line, flush=True)


run(30528, 768, ["256x256", "256x192", "128x256"])
run(18432, 768, ["256x256", "256x192", "128x256"])
run(3072, 768, ["256x192", "128x256"])
run(2304, 768, ["128x192", "256x192"])
run(768, 768, ["128x192"])
run(768, 3072, ["128x192"])
run(30528, 768, ["256x256"], M=2048)
run(30522 // 8 * 8, 768, ["256x256"], M=264)
