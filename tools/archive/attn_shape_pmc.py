#!/usr/bin/env python3
"""The two 32-token attention kernels on a rotation of cold buffers, a few launches each: the workload of tools/run_attn_shape_pmc.sh
(L1 -> L2 request counters of the row-chunk access shape against the whole-line one; KVQ_ATTN_COAL / KVQ_ATTN_STC select)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

dev, nh, H, B, S, NSET = "cuda", 12, 768, 256, 32, 24
g = torch.Generator(device=dev).manual_seed(0)
qkv = [torch.randn(B * S, 3 * H, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(NSET)]
go = [torch.randn(B * S, H, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(NSET)]
gq = [torch.empty_like(qkv[0]) for _ in range(NSET)]
ctx = [torch.empty_like(go[0]) for _ in range(NSET)]
pb = torch.empty((B, 3 * H), dtype=torch.float32, device=dev)
mask = torch.ones(B, S, dtype=torch.int64, device=dev)
for rep in range(2):
    for i in range(NSET):
        t, o = qkv[i], gq[i]
        nnops.attn_fwd(t[:, :H], t[:, H:2 * H], t[:, 2 * H:], mask, B, nh, S, S, False, 0.1, 9, 3, out=ctx[i])
    for i in range(NSET):
        t, o = qkv[i], gq[i]
        nnops.attn_bwd(t[:, :H], t[:, H:2 * H], t[:, 2 * H:], mask, go[i], B, nh, S, S, False, 0.1, 9, 3, o[:, :H], o[:, H:2 * H], o[:, 2 * H:],
                       pb[:, :H], pb[:, H:2 * H], pb[:, 2 * H:])
torch.cuda.synchronize()
print("ok")
