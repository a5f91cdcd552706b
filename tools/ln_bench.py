"""Time the dropout+residual+LayerNorm kernels and the elementwise kernels at the bench shape (N=8192, H=768 / 3072, bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops
N, H = 8192, 768
torch.manual_seed(0)
y = torch.randn(N, H, device="cuda").bfloat16(); r = torch.randn(N, H, device="cuda").bfloat16(); g = torch.randn(N, H, device="cuda").bfloat16()
gamma = torch.randn(H, device="cuda"); beta = torch.randn(H, device="cuda")
def t(fn, n=100):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out, pre, mean, rstd = nnops.ln_fwd(y, r, gamma, beta, 1e-12, 0.1, 3, 4)
print(f"ln_fwd  {t(lambda: nnops.ln_fwd(y, r, gamma, beta, 1e-12, 0.1, 3, 4)):.1f} us   (floor: 50 MB / 8 TB/s = 6.3 us)")
print(f"ln_bwd_partial {t(lambda: nnops.ln_bwd_partial(g, pre, mean, rstd, gamma, 0.1, 3, 4, want_dbias=True)):.1f} us   (floor: 60 MB -> 7.5 us)")
h = torch.randn(N, 4 * H, device="cuda").bfloat16(); ga = torch.randn(N, 4 * H, device="cuda").bfloat16()
print(f"gelu_fwd {t(lambda: nnops.gelu_fwd(h)):.1f} us (floor 12.6)   gelu_bwd {t(lambda: nnops.gelu_bwd(h, ga)):.1f} us   gelu_bwd_bias {t(lambda: nnops.gelu_bwd_bias(h, ga)):.1f} us (floor 18.9)")
