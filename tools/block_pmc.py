"""PMC target: the MFMA attention kernels and the own NT GEMM at the bench shapes (one process, a few launches each)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops
B, nh, S, H = 256, 12, 32, 768
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * H, device="cuda").bfloat16(); g = torch.randn(B * S, H, device="cuda").bfloat16()
mask = (torch.arange(S, device="cuda")[None] < torch.randint(4, 13, (B, 1), device="cuda")).long()
gq = torch.empty_like(qkv); pb = torch.empty(B, 3 * H, device="cuda")
a = torch.randn(B * S, H, device="cuda").bfloat16(); w = (torch.randn(H, H, device="cuda") * 0.05).bfloat16(); bias = torch.randn(H, device="cuda").bfloat16()
out = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    nnops.attn_fwd(qkv[:, :H], qkv[:, H:2*H], qkv[:, 2*H:], mask, B, nh, S, S, True, 0.1, 1, 2)
    nnops.attn_bwd(qkv[:, :H], qkv[:, H:2*H], qkv[:, 2*H:], mask, g, B, nh, S, S, True, 0.1, 1, 2, gq[:, :H], gq[:, H:2*H], gq[:, 2*H:],
                   pb[:, :H], pb[:, H:2*H], pb[:, 2*H:])
    nnops.gemm(a, w, "nt", bias=bias, out=out)
torch.cuda.synchronize()
print("ok")
