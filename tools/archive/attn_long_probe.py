"""The blocked attention kernels (33 .. 128 tokens) at 8192 tokens x 12 heads, bf16, dropout 0.1: us per launch against the 32-token
kernels at the same token count (cold rotating buffers)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
import torch
from kvq import nnops

dev, nh, H, NB = torch.device("cuda:0"), 12, 768, 8
g = torch.Generator(device=dev).manual_seed(0)
qkv = [torch.randn(8192, 3 * H, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(NB)]
gq = [torch.empty_like(t) for t in qkv]
go = [torch.randn(8192, H, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(NB)]


def timed(fn, reps=32):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % NB)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for S in (32, 64, 128):
    B = 8192 // S
    mask = torch.ones(B, S, dtype=torch.int64, device=dev)
    for causal in (False, True):
        saved = [nnops.attn_fwd(t[:, :H], t[:, H:2 * H], t[:, 2 * H:], mask, B, nh, S, S, causal, 0.1, 9, 3) for t in qkv]
        tf = timed(lambda i: nnops.attn_fwd(qkv[i][:, :H], qkv[i][:, H:2 * H], qkv[i][:, 2 * H:], mask, B, nh, S, S, causal, 0.1, 9, 3, out=saved[i][0]))
        tb = timed(lambda i: nnops.attn_bwd(qkv[i][:, :H], qkv[i][:, H:2 * H], qkv[i][:, 2 * H:], mask, go[i], B, nh, S, S, causal, 0.1, 9, 3,
                                            gq[i][:, :H], gq[i][:, H:2 * H], gq[i][:, 2 * H:], ctx=saved[i][0], lse=saved[i][1]))
        print(f"S = {S:3d} (B = {B:3d}) causal {int(causal)}: forward {tf:6.1f} us, backward {tb:6.1f} us", flush=True)
