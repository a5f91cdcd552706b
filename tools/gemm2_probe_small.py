#!/usr/bin/env python3
"""Own GEMM family vs torch (hipBLASLt) at the row counts of the REFERENCE's own batches: 12 tokens x 64 / 128 / 512 sentences =
768 / 1536 / 6144 rows (models/shelgon3/Trainer.py:82; analysis scripts use 512 / 2048 sentences), plus 3072, for every product
shape of the bert-base step and every tile of csrc/kvq_gemm2.hip incl. the 64x128 small tile; interleaved rounds in one process,
back-to-back launches on warm operands (what a launch-latency-sized product sees inside a replayed graph).
usage: gemm2_probe_small.py [rounds]   -> table for profiles/ and the cut-over rule of kvq/nnops.py::pick_tile"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda"
TL = ["64x128", "128x192", "128x256", "256x192", "256x256"]


def bench(fn, iters=30):
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


def run(name, layout, M, N, K):
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
    fns, res = {"lib": lib}, {"lib": []}
    for t in TL:
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        fns[t] = (lambda t=t, out=out: nnops.gemm(a, b, layout, out=out, tile=t))
        rel = (fns[t]().float() - ref).norm().item() / ref.norm().item()
        assert rel < 5e-3, (name, t, rel)
        res[t] = []
    auto = nnops.TILE_NAMES[nnops.pick_tile(M, N, K)]
    for _ in range(rounds):
        for k, f in fns.items():
            res[k].append(bench(f))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    best = min(TL, key=lambda t: med[t])
    print(f"| {name} | {layout} | {M} | {N} | {K} | {med['lib']:.1f} | " + " | ".join(f"{med[t]:.1f}" for t in TL) +
          f" | {best} | {auto} ({med[auto]:.1f}) |", flush=True)


print("| product | layout | M | N | K | library us | " + " | ".join(t + " us" for t in TL) + " | best own | pick_tile |")
print("|---|---|---|---|---|---|" + "---|" * len(TL) + "---|---|")
for T in (768, 1536, 3072, 6144):
    for n, k in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (18432, 768), (30528, 768)]:
        run("forward", "nt", T, n, k)
    for n, k in [(768, 768), (768, 2304), (768, 3072), (3072, 768), (768, 18432), (768, 30528)]:
        run("input grad", "nn", T, n, k)
    for m, n in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (18432, 768), (30528, 768)]:
        run("weight grad", "tn", m, n, T)
