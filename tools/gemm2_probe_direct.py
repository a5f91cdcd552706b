#!/usr/bin/env python3
"""LDS-transposed epilogue vs the register-direct one (v_permlane16_swap, 64-byte row segments) of csrc/kvq_gemm2.hip, NT layout,
interleaved rounds in one process.  usage: gemm2_probe_direct.py [rounds]"""
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


def run(N, K, tiles):
    a, b, bias = rnd(T, K), rnd(N, K), rnd(N)
    ref = torch.addmm(bias, a, b.t()).float()
    fns, res = {"lib": lambda: torch.addmm(bias, a, b.t())}, {"lib": []}
    for t in tiles:
        for d in ("0", "1"):
            out = torch.empty((T, N), device=dev, dtype=torch.bfloat16)

            def f(t=t, d=d, out=out):
                os.environ["KVQ_GEMM_DIRECT"] = d
                return nnops.gemm(a, b, "nt", bias=bias, out=out, tile=t)
            out.zero_()
            o = f()
            rel = (o.float() - ref).norm().item() / ref.norm().item()
            assert rel < 5e-3, (N, K, t, d, rel)
            fns[f"{t}/{'direct' if d == '1' else 'lds'}"] = f
            res[f"{t}/{'direct' if d == '1' else 'lds'}"] = []
    for _ in range(rounds):
        for k, f in fns.items():
            res[k].append(bench(f))
    fl = 2.0 * T * N * K
    line = f"nt M={T} N={N:6d} K={K:5d}: "
    for k, v in res.items():
        m = sorted(v)[len(v) // 2]
        line += f"{k} {m:7.1f} us {fl / m / 1e6:5.0f} TF | "
    print(line, flush=True)


run(768, 768, ["128x192"])
run(2304, 768, ["128x192", "256x192"])
run(3072, 768, ["256x192", "128x256"])
run(768, 3072, ["128x192"])
run(18432, 768, ["256x256", "256x192"])
run(30528, 768, ["256x256", "256x192"])
