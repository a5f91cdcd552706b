#!/usr/bin/env python3
"""Own GEMM family (csrc/kvq_gemm2.hip) vs torch (hipBLASLt) on the GEMM shapes of the benchmarked step, interleaved rounds
in one process (MI355X).  usage: gemm2_probe.py [fwd|dgrad|wgrad|all] [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

T = 8192
which = sys.argv[1] if len(sys.argv) > 1 else "all"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda"


def bench(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3       # us


def rnd(*s):
    return torch.randn(s, device=dev).to(torch.bfloat16)


def run(name, layout, M, N, K, tiles):
    if layout == "nt":
        a, b = rnd(M, K), rnd(N, K)
        lib = lambda: torch.mm(a, b.t())
    elif layout == "nn":
        a, b = rnd(M, K), rnd(K, N)
        lib = lambda: torch.mm(a, b)
    else:
        a, b = rnd(K, M), rnd(K, N)
        lib = lambda: torch.mm(a.t(), b)
    ref = lib().float()
    fl = 2.0 * M * N * K
    res = {"lib": []}
    fns = {"lib": lib}
    for t in tiles:
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        fns[t] = (lambda t=t, out=out: nnops.gemm(a, b, layout, out=out, tile=t))
        o = fns[t]()
        rel = (o.float() - ref).norm().item() / ref.norm().item()
        assert rel < 5e-3, (name, t, rel)
        res[t] = []
    for _ in range(rounds):
        for k, f in fns.items():
            res[k].append(bench(f))
    line = f"{name:10s} {layout} M={M:6d} N={N:6d} K={K:6d}: "
    for k, v in res.items():
        m = sorted(v)[len(v) // 2]
        line += f"{k} {m:7.1f} us {fl / m / 1e6:6.0f} TF | "
    print(line, flush=True)


TL = ["128x192", "128x256", "256x192", "256x256"]
if which in ("fwd", "all"):
    for n, k in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (18432, 768), (30528, 768)]:
        run("fwd", "nt", T, n, k, TL)
if which in ("dgrad", "all"):
    for n, k in [(768, 768), (768, 2304), (768, 3072), (3072, 768), (768, 18432), (768, 30528)]:
        run("dgrad", "nn", T, n, k, TL)
if which in ("wgrad", "all"):
    for m, n in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (18432, 768), (30528, 768)]:
        run("wgrad", "tn", m, n, T, TL)
    # one encoder layer's four weight gradients as one grouped launch vs four library calls
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    gys = [rnd(T, m) for m, _ in shapes]
    xs = [rnd(T, n) for _, n in shapes]
    outs = [torch.empty((m, n), device=dev, dtype=torch.bfloat16) for m, n in shapes]
    probs = [nnops.gemm_problem(g, x, o, "tn") for g, x, o in zip(gys, xs, outs)]
    fl = sum(2.0 * T * m * n for m, n in shapes)
    def lib4():
        for g, x, o in zip(gys, xs, outs):
            torch.mm(g.t(), x, out=o)
    for tile in ("128x256", "128x192", "256x192", "256x256"):
        ts = sorted(bench(lambda: nnops.gemm_grouped(probs, "tn", tile)) for _ in range(rounds))
        print(f"enc-layer wgrad grouped {tile}: {ts[len(ts)//2]:7.1f} us {fl / ts[len(ts)//2] / 1e6:6.0f} TF", flush=True)
    ts = sorted(bench(lib4) for _ in range(rounds))
    print(f"enc-layer wgrad 4 x torch.mm: {ts[len(ts)//2]:7.1f} us {fl / ts[len(ts)//2] / 1e6:6.0f} TF", flush=True)
if which in ("epi", "all"):
    # epilogue variants at the FFN shapes: accumulate, GELU, dGELU + bias partials
    x, w1, b1 = rnd(T, 768), rnd(3072, 768), rnd(3072)
    h = torch.addmm(b1, x, w1.t())
    gf, w2 = rnd(T, 768), rnd(768, 3072)
    gx = rnd(T, 768)
    gy3 = rnd(T, 3072)
    w1n = rnd(3072, 768)
    cases = {
        "FFN1 fwd lib addmm+gelu": lambda: torch.nn.functional.gelu(torch.addmm(b1, x, w1.t())),
        "FFN1 fwd own plain 256x192": lambda: nnops.gemm(x, w1, "nt", bias=b1, tile="256x192"),
        "FFN1 fwd own fused gelu 256x192": lambda: nnops.gemm_gelu(x, w1, b1, tile="256x192"),
        "FFN1 fwd own fused gelu 128x256": lambda: nnops.gemm_gelu(x, w1, b1, tile="128x256"),
        "FFN2 dgrad own plain 256x192": lambda: nnops.gemm(gf, w2, "nn", tile="256x192"),
        "FFN2 dgrad own fused dgelu 256x192": lambda: nnops.gemm_dgelu(gf, w2, h, tile="256x192"),
        "FFN2 dgrad own fused dgelu 128x256": lambda: nnops.gemm_dgelu(gf, w2, h, tile="128x256"),
        "FFN1 dgrad own 128x256": lambda: nnops.gemm(gy3, w1n, "nn", tile="128x256"),
        "FFN1 dgrad own 128x256 accumulate": lambda: nnops.gemm(gy3, w1n, "nn", out=gx, accumulate=True, tile="128x256"),
        "FFN1 dgrad lib addmm_": lambda: gx.addmm_(gy3, w1n),
    }
    res = {k: [] for k in cases}
    for _ in range(rounds):
        for k, f in cases.items():
            res[k].append(bench(f))
    for k, v in res.items():
        print(f"{k:40s} {sorted(v)[len(v)//2]:7.1f} us", flush=True)
