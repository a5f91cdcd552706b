"""How does the MFMA attention backward scale with the number of workgroups (one wave per (sentence, head))?  Tail-round check."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops
nh, S, H = 12, 32, 768
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (64, 128, 170, 171, 213, 256, 341, 512):
    qkv = torch.randn(B * S, 3 * H, device="cuda").bfloat16(); g = torch.randn(B * S, H, device="cuda").bfloat16()
    mask = (torch.arange(S, device="cuda")[None] < torch.randint(4, 13, (B, 1), device="cuda")).long()
    gq = torch.empty_like(qkv)
    q, k, v = qkv[:, :H], qkv[:, H:2*H], qkv[:, 2*H:]
    f = t(lambda: nnops.attn_fwd(q, k, v, mask, B, nh, S, S, True, 0.1, 1, 2))
    b = t(lambda: nnops.attn_bwd(q, k, v, mask, g, B, nh, S, S, True, 0.1, 1, 2, gq[:, :H], gq[:, H:2*H], gq[:, 2*H:]))
    print(f"B={B:4d} ({B*nh:5d} waves, {B*nh/256:5.2f}/CU): fwd {f:6.1f} us  bwd {b:6.1f} us   per 1k waves: fwd {f/(B*nh)*1e3:5.2f} bwd {b/(B*nh)*1e3:5.2f}")
