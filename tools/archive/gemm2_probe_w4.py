#!/usr/bin/env python3
"""256 x 256 tile with FOUR waves (one per SIMD, 128 x 128 per wave: 0.5 LDS fragment reads per MFMA instead of 0.75) against the
eight-wave 256 x 256 and 128 x 256 tiles, on the products of the step that run 256-wide tiles: weight gradients (TN, contraction
over 8192 tokens; single and grouped launches of two layers), LM head and all-layer cross-K/V forward (NT), LM-head input gradient
(NN).  Interleaved rounds, one process.  (The "256x256w4" tile was removed after this measurement -- profiles/r04_rejected.md; re-add `Cfg<256, 256, 2, 2, AK, BKM, 2>` as
a tile of csrc/kvq_gemm2.hip to run the probe again.)
usage: gemm2_probe_w4.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev, T = "cuda", 8192
TL = ["256x256", "256x256w4", "128x256", "256x192"]


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


def run(name, layout, M, N, K):
    if layout == "nt":
        a, b = rnd(M, K), rnd(N, K)
        ref = a.float() @ b.float().t() if M * N < 1 << 27 else None
    elif layout == "nn":
        a, b = rnd(M, K), rnd(K, N)
        ref = a.float() @ b.float()
    else:
        a, b = rnd(K, M), rnd(K, N)
        ref = a.float().t() @ b.float()
    fns, res = {}, {}
    for t in TL:
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        fns[t] = (lambda t=t, out=out: nnops.gemm(a, b, layout, out=out, tile=t))
        o = fns[t]()
        if ref is not None:
            rel = (o.float() - ref).norm().item() / ref.norm().item()
            assert rel < 5e-3, (name, t, rel)
        res[t] = []
    for _ in range(rounds):
        for k, f in fns.items():
            res[k].append(bench(f))
    fl = 2.0 * M * N * K
    print(f"{name:22s} {layout} M={M:6d} N={N:6d} K={K:6d}: " + " | ".join(
        f"{t} {sorted(v)[len(v) // 2]:7.1f} us {fl / sorted(v)[len(v) // 2] / 1e6:5.0f} TF" for t, v in res.items()), flush=True)


for m, n in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (18432, 768), (30528, 768)]:
    run("weight gradient", "tn", m, n, T)
run("LM head forward", "nt", T, 30528, 768)
run("cross-K/V forward", "nt", T, 18432, 768)
run("FFN1 forward", "nt", T, 3072, 768)
run("LM head input grad", "nn", T, 768, 30528)
# the engine's grouped launch: the weight gradients of two decoder layers (252 tiles of 256 x 256)
shapes = [(2304, 768), (768, 768), (768, 768), (768, 768), (3072, 768), (768, 3072)] * 2
gys = [rnd(T, m) for m, _ in shapes]
xs = [rnd(T, n) for _, n in shapes]
outs = [torch.empty((m, n), device=dev, dtype=torch.bfloat16) for m, n in shapes]
probs = [nnops.gemm_problem(g, x, o, "tn") for g, x, o in zip(gys, xs, outs)]
fl = sum(2.0 * T * m * n for m, n in shapes)
res = {t: [] for t in ("256x256", "256x256w4", "128x256")}
for _ in range(rounds):
    for t in res:
        res[t].append(bench(lambda: nnops.gemm_grouped(probs, "tn", t)))
print("two decoder layers grouped (252 tiles): " + " | ".join(f"{t} {sorted(v)[len(v) // 2]:7.1f} us {fl / sorted(v)[len(v) // 2] / 1e6:5.0f} TF" for t, v in res.items()))
nnops.gemm_grouped(probs, "tn", "256x256w4")
for o, g, x in zip(outs, gys, xs):
    ref = g.float().t() @ x.float()
    assert (o.float() - ref).norm().item() / ref.norm().item() < 5e-3
print("grouped w4 results match f32 products")
