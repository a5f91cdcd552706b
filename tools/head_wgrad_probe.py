"""LM-head weight gradient: [Vp,H] = g_logits^T hN directly vs. as its transpose + flip (scratch tool)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import engine
engine._load_gemm_tuning()
N, Vp, H = 8192, 30528, 768
g = torch.randn(N, Vp, device="cuda").bfloat16(); hN = torch.randn(N, H, device="cuda").bfloat16()
out = torch.empty(Vp, H, device="cuda", dtype=torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def flip():
    gWt = torch.mm(hN.t(), g); out.copy_(gWt.t())
print(f"transpose+flip {t(flip):.1f} us;  mm only {t(lambda: torch.mm(hN.t(), g)):.1f} us")
print(f"direct mm(g.t(), hN, out) {t(lambda: torch.mm(g.t(), hN, out=out)):.1f} us")
for S in (2, 4, 8):
    def split():
        part = torch.bmm(g.view(S, N // S, Vp).transpose(1, 2), hN.view(S, N // S, H)); return part
    print(f"bmm split {S}: {t(split):.1f} us (+ slab sum)")
