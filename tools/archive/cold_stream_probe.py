#!/usr/bin/env python3
"""What a short HBM-streaming kernel can reach with COLD operands (as in the step): a plain 16-byte copy kernel (torch clone) next to
the LayerNorm / attention kernels, each on a rotation of buffers larger than the Infinity Cache.  Run under
rocprofv3 --kernel-trace --stats; read the per-kernel averages."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

N, H, B, nh, S = 8192, 768, 256, 12, 32
PD = float(os.environ.get("PROBE_PDROP", "0.1"))
R = 24                                                       # rotation: 24 x (12.6 .. 38) MB per operand
dev = "cuda"
ys = [torch.randn(N, H, device=dev).bfloat16() for _ in range(R)]
rs = [torch.randn(N, H, device=dev).bfloat16() for _ in range(R)]
big = [torch.randn(N, 4 * H, device=dev).bfloat16() for _ in range(R)]        # 50 MB copies
qkv = [torch.randn(N, 3 * H, device=dev).bfloat16() for _ in range(R)]
gq = [torch.empty(N, 3 * H, device=dev, dtype=torch.bfloat16) for _ in range(R)]
gamma = torch.randn(H, device=dev); beta = torch.randn(H, device=dev)
mask = (torch.arange(S, device=dev)[None] < torch.randint(4, 13, (B, 1), device=dev)).long()
outs = [torch.empty(N, 4 * H, device=dev, dtype=torch.bfloat16) for _ in range(R)]
for rep in range(3):
    for i in range(R):
        outs[i].copy_(big[i])                                                  # 50 MB in + 50 MB out
        o, pre, mean, rstd = nnops.ln_fwd(ys[i], rs[i], gamma, beta, 1e-12, PD, 3, 4)
        nnops.ln_bwd_partial(rs[(i + 7) % R], ys[(i + 3) % R], mean, rstd, gamma, PD, 3, 4, want_dbias=True)
        q, k, v = qkv[i][:, :H], qkv[i][:, H:2 * H], qkv[i][:, 2 * H:]
        ctx, _ = nnops.attn_fwd(q, k, v, mask, B, nh, S, S, False, PD, 1, 2)
        nnops.attn_bwd(q, k, v, mask, ys[(i + 5) % R], B, nh, S, S, False, PD, 1, 2, gq[i][:, :H], gq[i][:, H:2 * H], gq[i][:, 2 * H:])
torch.cuda.synchronize()
print("ok")
