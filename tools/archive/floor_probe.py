"""Fixed-latency floor of the block kernels: time at a tiny row count next to the bench size (scratch tool)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops
H = 768
def t(fn, n=100):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gamma = torch.randn(H, device="cuda"); beta = torch.randn(H, device="cuda")
for N in (64, 1024, 4096, 8192):
    y = torch.randn(N, H, device="cuda").bfloat16(); r = torch.randn(N, H, device="cuda").bfloat16(); g = torch.randn(N, H, device="cuda").bfloat16()
    out, pre, mean, rstd = nnops.ln_fwd(y, r, gamma, beta, 1e-12, 0.1, 3, 4)
    h = torch.randn(N, 4 * H, device="cuda").bfloat16(); ga = torch.randn(N, 4 * H, device="cuda").bfloat16()
    e = torch.empty(N, H, device="cuda", dtype=torch.bfloat16)
    print(f"N={N:5d}: empty-ish dropout(p=0) {t(lambda: nnops.dropout(y, 0.0, 1, 1, out=e)):5.1f}  ln_fwd {t(lambda: nnops.ln_fwd(y, r, gamma, beta, 1e-12, 0.1, 3, 4)):5.1f}  "
          f"ln_bwd {t(lambda: nnops.ln_bwd_partial(g, pre, mean, rstd, gamma, 0.1, 3, 4, want_dbias=True)):5.1f}  gelu_fwd {t(lambda: nnops.gelu_fwd(h)):5.1f}  "
          f"gelu_bwd_bias {t(lambda: nnops.gelu_bwd_bias(h, ga)):5.1f} us")
