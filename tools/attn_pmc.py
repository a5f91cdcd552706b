import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops
B, nh, S, H = 256, 12, 32, 768
qkv = torch.randn(B * S, 3 * H, device="cuda").bfloat16(); g = torch.randn(B * S, H, device="cuda").bfloat16()
mask = (torch.arange(S, device="cuda")[None] < torch.randint(4, 13, (B, 1), device="cuda")).long()
gq = torch.empty_like(qkv)
for _ in range(5):
    ctx, lse = nnops.attn_fwd(qkv[:, :H], qkv[:, H:2*H], qkv[:, 2*H:], mask, B, nh, S, S, True, 0.1, 1, 2)
    nnops.attn_bwd(qkv[:, :H], qkv[:, H:2*H], qkv[:, 2*H:], mask, g, B, nh, S, S, True, 0.1, 1, 2, gq[:, :H], gq[:, H:2*H], gq[:, 2*H:])
torch.cuda.synchronize()
