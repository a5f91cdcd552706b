"""What the Philox dropout bits cost inside the LayerNorm and attention kernels: each kernel at step shapes (8192 tokens, 768 wide,
256 x 12 heads x 32), p = 0 against p = 0.1, HIP events around 50 launches over rotating buffers (cold operands, as in the step)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
import torch
from kvq import nnops

dev = torch.device("cuda:0")
N, H, B, nh, S = 8192, 768, 256, 12, 32
NB = 12                                          # rotating buffer sets (12 x ~50 MB: well past the L2s)
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
ys = [torch.randn(N, H, device=dev, dtype=bf, generator=g) for _ in range(NB)]
rs = [torch.randn(N, H, device=dev, dtype=bf, generator=g) for _ in range(NB)]
gamma, beta = torch.ones(H, device=dev, dtype=bf), torch.zeros(H, device=dev, dtype=bf)
qkv = [torch.randn(N, 3 * H, device=dev, dtype=bf, generator=g) for _ in range(NB)]
gq = [torch.empty(N, 3 * H, device=dev, dtype=bf) for _ in range(NB)]
mask = torch.ones(B, S, dtype=torch.int64, device=dev)


def timed(fn, reps=48):
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


for p in (0.0, 0.1):
    saved = [nnops.ln_fwd(ys[i], rs[i], gamma, beta, 1e-12, p, 1234, 7) for i in range(NB)]
    t_f = timed(lambda i: nnops.ln_fwd(ys[i], rs[i], gamma, beta, 1e-12, p, 1234, 7))
    t_b = timed(lambda i: nnops.ln_bwd_partial(ys[i], saved[i][1], saved[i][2], saved[i][3], gamma, p, 1234, 7, want_dbias=True))
    ctx = [nnops.attn_fwd(qkv[i][:, :H], qkv[i][:, H:2 * H], qkv[i][:, 2 * H:], mask, B, nh, S, S, False, p, 99, 3)[0] for i in range(NB)]
    t_af = timed(lambda i: nnops.attn_fwd(qkv[i][:, :H], qkv[i][:, H:2 * H], qkv[i][:, 2 * H:], mask, B, nh, S, S, False, p, 99, 3, out=ctx[i]))
    t_ab = timed(lambda i: nnops.attn_bwd(qkv[i][:, :H], qkv[i][:, H:2 * H], qkv[i][:, 2 * H:], mask, ctx[i], B, nh, S, S, False, p, 99, 3,
                                          gq[i][:, :H], gq[i][:, H:2 * H], gq[i][:, 2 * H:]))
    print(f"p_drop {p}: LN fwd {t_f:.1f} us, LN bwd (partial) {t_b:.1f} us, attention fwd {t_af:.1f} us, attention bwd {t_ab:.1f} us", flush=True)
