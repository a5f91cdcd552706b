#!/usr/bin/env python3
"""The step's memory-bound kernels on rotating (cold) operand sets at the step's shapes: LayerNorm fwd/bwd with and without
dropout, GELU fwd/bwd, attention fwd/bwd.  Prints us per launch and the GB/s of the bytes each launch has to move."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

dev = "cuda"
B, S, H, NH, F = 256, 32, 768, 12, 3072
N = B * S
R = 24


def rnd(*s):
    return torch.randn(s, device=dev).to(torch.bfloat16)


def timeit(name, fn, sets, nbytes):
    fn(*sets[0])
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in sets:
            fn(*s)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / len(sets) * 1e3)
    us = sorted(best)[len(best) // 2]
    print(f"{name:34s} {us:7.1f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)


gamma = torch.ones(H, device=dev)
beta = torch.zeros(H, device=dev)
sets = [(rnd(N, H), rnd(N, H)) for _ in range(R)]
for p in (0.0, 0.1):
    timeit(f"ln_fwd p={p}", lambda y, x: nnops.ln_fwd(y, x, gamma, beta, 1e-12, p, 1234, 3), sets, 4 * N * H * 2)
saved = [nnops.ln_fwd(y, x, gamma, beta, 1e-12, 0.1, 1234, 3) for y, x in sets]
bsets = [(rnd(N, H), s[1], s[2], s[3]) for s in saved]
for p in (0.0, 0.1):
    timeit(f"ln_bwd_partial p={p}", lambda g, pre, mean, rstd: nnops.ln_bwd_partial(g, pre, mean, rstd, gamma, p, 1234, 3, want_dbias=True),
           bsets, 4 * N * H * 2)
del saved, bsets, sets
hs = [(rnd(N, F),) for _ in range(R)]
timeit("gelu_fwd", lambda h: nnops.gelu_fwd(h), hs, 2 * N * F * 2)
gs = [(h[0], rnd(N, F)) for h in hs]
timeit("gelu_bwd_bias (in place)", lambda h, g: nnops.gelu_bwd_bias(h, g, out=g), gs, 3 * N * F * 2)
del hs, gs
mask = torch.ones((B, S), device=dev, dtype=torch.int64)
qs = [(rnd(N, 3 * H), torch.empty((N, H), device=dev, dtype=torch.bfloat16)) for _ in range(R)]
for p in (0.0, 0.1):
    timeit(f"attn_fwd p={p}", lambda qkv, out: nnops.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask, B, NH, S, S, False, p, 99, 5, out=out),
           qs, 4 * N * H * 2)
bs = [(q[0], rnd(N, H), torch.empty((N, 3 * H), device=dev, dtype=torch.bfloat16), torch.empty((B, 3 * H), device=dev)) for q in qs]
for p in (0.0, 0.1):
    timeit(f"attn_bwd p={p}",
           lambda qkv, g, gq, pb: nnops.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask, g, B, NH, S, S, False, p, 99, 5,
                                                 gq[:, :H], gq[:, H:2 * H], gq[:, 2 * H:], pb[:, :H], pb[:, H:2 * H], pb[:, 2 * H:]),
           bs, 7 * N * H * 2)
