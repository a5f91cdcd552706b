"""Time the bf16 attention kernel flavours (0 fma, 1 dot2, 2 mfma) at the bench shape: B=256, 12 heads, S=32, dropout 0.1."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops, _ffi
B, nh, S, H = 256, 12, 32, 768
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * H, device="cuda").bfloat16(); g = torch.randn(B * S, H, device="cuda").bfloat16()
mask = (torch.arange(S, device="cuda")[None] < torch.randint(4, 13, (B, 1), device="cuda")).long()
gq = torch.empty_like(qkv)
q, k, v = qkv[:, :H], qkv[:, H:2*H], qkv[:, 2*H:]
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for var in (0, 1, 2):
    _ffi.lib().kvq_attn_set_variant(var)
    f = t(lambda: nnops.attn_fwd(q, k, v, mask, B, nh, S, S, True, 0.1, 1, 2))
    b = t(lambda: nnops.attn_bwd(q, k, v, mask, g, B, nh, S, S, True, 0.1, 1, 2, gq[:, :H], gq[:, H:2*H], gq[:, 2*H:]))
    pb = torch.empty(B, 3 * H, device="cuda")
    bp = t(lambda: nnops.attn_bwd(q, k, v, mask, g, B, nh, S, S, True, 0.1, 1, 2, gq[:, :H], gq[:, H:2*H], gq[:, 2*H:],
                                  pb[:, :H], pb[:, H:2*H], pb[:, 2*H:]))
    print(f"variant {var}: fwd {f:.1f} us  bwd {b:.1f} us  bwd + bias partials {bp:.1f} us")
