import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops
M, N, K = 8192, 768, 3072
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    nnops.gemm_nt(a, w, None, out=out)
torch.cuda.synchronize()
