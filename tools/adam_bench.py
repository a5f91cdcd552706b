"""Adam kernel at the bench size (248 M parameters, bf16 gradients + bf16 shadow): time and bytes/s (scratch tool)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops
n = 247_800_000 // 8 * 8
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda").bfloat16(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
def t(fn, k=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
ms = t(lambda: nnops.adam_step(p, g, m, v, 3, 1e-4, shadow=sh))
print(f"adam {ms*1e3:.0f} us  {n*28/ms/1e9:.2f} TB/s")
