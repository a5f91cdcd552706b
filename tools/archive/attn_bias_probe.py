#!/usr/bin/env python3
"""Where do the 8 us go that attn_bwd_mfma_kernel takes in the step (29 us) over the cold-operand probe (21 us)?  The step asks the
kernel for the q/k/v bias-gradient partial rows; the probe did not.  Times, on a rotation of cold buffer sets (larger than the 256 MiB
Infinity Cache), 256 sentences x 12 heads x 32 tokens, bf16, dropout 0.1:
  (a) attn_bwd without partials            (b) with the three partial outputs (self-attention form)
  (c) colsum_partial over the stored [N, 3H] gradient -- the alternative way to the same bias gradients
usage: attn_bias_probe.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev, nh, H, B, S = "cuda", 12, 768, 256, 32
NSET = 24
g = torch.Generator(device=dev).manual_seed(0)
qkv = [torch.randn(B * S, 3 * H, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(NSET)]
go = [torch.randn(B * S, H, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(NSET)]
gq = [torch.empty_like(qkv[0]) for _ in range(NSET)]
pb = [torch.empty((B, 3 * H), dtype=torch.float32, device=dev) for _ in range(NSET)]
mask = torch.ones(B, S, dtype=torch.int64, device=dev)


def bwd(i, partials, causal=False):
    t, o = qkv[i], gq[i]
    extra = (pb[i][:, :H], pb[i][:, H:2 * H], pb[i][:, 2 * H:]) if partials else ()
    nnops.attn_bwd(t[:, :H], t[:, H:2 * H], t[:, 2 * H:], mask, go[i], B, nh, S, S, causal, 0.1, 9, 3, o[:, :H], o[:, H:2 * H], o[:, 2 * H:], *extra)


def timed(fn):
    for i in range(NSET):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(NSET):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / NSET * 1e3


def fwd(i, causal=False):
    t = qkv[i]
    nnops.attn_fwd(t[:, :H], t[:, H:2 * H], t[:, 2 * H:], mask, B, nh, S, S, causal, 0.1, 9, 3, out=go[(i + 1) % NSET])


cases = {"attn_fwd": lambda i: fwd(i), "attn_bwd, no partials": lambda i: bwd(i, False), "attn_bwd + q/k/v bias partials": lambda i: bwd(i, True),
         "attn_bwd causal + partials": lambda i: bwd(i, True, True),
         "colsum_partial of [N, 3H]": lambda i: nnops.colsum_partial(gq[i]),
         "device copy 50 MB (yardstick)": lambda i: gq[i].copy_(qkv[i])}
res = {k: [] for k in cases}
for _ in range(rounds):
    for k, f in cases.items():
        res[k].append(timed(f))
for k, v in res.items():
    print(f"{k:36s} {sorted(v)[len(v) // 2]:7.1f} us   (all: {' '.join(f'{x:.1f}' for x in v)})", flush=True)
