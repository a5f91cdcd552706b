#!/usr/bin/env python3
"""Grouped weight-gradient launches (TN, contraction over 8192 tokens) of 1, 2 and 4 encoder layers per launch, per tile,
on rotating (cold) operand sets; us per LAYER."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

T, dev = 8192, "cuda"
rnd = lambda *s: torch.randn(s, device=dev).to(torch.bfloat16)  # noqa: E731
LAYER = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
DEC = LAYER + [(768, 768), (768, 768)]


def make(shapes):
    gys = [rnd(T, m) for m, _ in shapes]
    xs = [rnd(T, n) for _, n in shapes]
    outs = [torch.empty((m, n), device=dev, dtype=torch.bfloat16) for m, n in shapes]
    return [nnops.gemm_problem(g, x, o, "tn") for g, x, o in zip(gys, xs, outs)], (gys, xs, outs)


def check(shapes, tile):
    probs, (gys, xs, outs) = make(shapes)
    nnops.gemm_grouped(probs, "tn", tile)
    for g, x, o in zip(gys, xs, outs):
        ref = g.float().t() @ x.float()
        rel = (o.float() - ref).norm().item() / ref.norm().item()
        assert rel < 5e-3, (tile, rel)


def run(name, shapes, layers, tile, sets=6):
    pool = [make(shapes) for _ in range(sets)]
    fl = sum(2.0 * T * m * n for m, n in shapes)
    ts = []
    for _ in range(5):
        nnops.gemm_grouped(pool[0][0], "tn", tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for probs, _ in pool:
            nnops.gemm_grouped(probs, "tn", tile)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / sets * 1e3)
    us = sorted(ts)[2]
    print(f"{name:22s} {tile}: {us:7.1f} us = {us / layers:6.1f} us/layer  {fl / us / 1e6:6.0f} TF", flush=True)


for t in ("128x256", "256x256", "256x192", "128x192"):   # (round 2 also had "192x192" here)
    check(LAYER, t)
run("1 enc layer", LAYER, 1, "128x256")
run("2 enc layers", LAYER * 2, 2, "256x256")
run("4 enc layers", LAYER * 4, 4, "256x256")
run("2 dec layers", DEC * 2, 2, "256x256")
run("cross-KV (18432x768)", [(18432, 768)], 1, "256x256", sets=4)
run("LM head (30528x768)", [(30528, 768)], 1, "256x256", sets=3)
run("LM head (30528x768)", [(30528, 768)], 1, "256x192", sets=3)
run("LM head (30528x768)", [(30528, 768)], 1, "128x256", sets=3)
