"""Own NT GEMM kernel vs torch (hipBLASLt) on the step's forward shapes: correctness + time (scratch tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops  # noqa: E402


def t(fn, it=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


torch.manual_seed(0)
M = 8192
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (1536, 768), (30528, 768), (768, 2304)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    ref = torch.addmm(bias.float(), a.float(), w.float().t())
    out = nnops.gemm_nt(a, w, bias)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    base = torch.addmm(bias, a, w.t())
    err_b = (base.float() - ref).abs().max().item() / ref.abs().max().item()
    c0 = torch.randn(M, N, device="cuda").bfloat16()
    c1 = c0.clone()
    nnops.gemm_nt(a, w, None, out=c1, accumulate=True)
    err_acc = (c1.float() - (c0.float() + a.float() @ w.float().t())).abs().max().item() / ref.abs().max().item()
    from kvq import _ffi
    _ffi.lib().kvq_gemm_set_stages(2)
    us_own2 = t(lambda: nnops.gemm_nt(a, w, bias, out=out))
    _ffi.lib().kvq_gemm_set_stages(3)
    out3 = nnops.gemm_nt(a, w, bias)
    assert torch.equal(out3, out), "3-stage result differs from 2-stage"
    us_own = t(lambda: nnops.gemm_nt(a, w, bias, out=out))
    us_ref = t(lambda: torch.addmm(bias, a, w.t(), out=base))
    fl = 2.0 * M * N * K
    print(f"N={N:6d} K={K:5d}: own2 {us_own2:7.1f} us | own3 {us_own:7.1f} us {fl/us_own/1e6:6.0f} TF (rel err {err:.1e}, acc {err_acc:.1e}) | torch {us_ref:7.1f} us {fl/us_ref/1e6:6.0f} TF (rel err {err_b:.1e})", flush=True)

import torch.nn.functional as F
a = torch.randn(M, 768, device="cuda").bfloat16(); w = (torch.randn(3072, 768, device="cuda") * 0.05).bfloat16(); bias = torch.randn(3072, device="cuda").bfloat16()
h, g = nnops.gemm_nt_gelu(a, w, bias)
href = torch.addmm(bias.float(), a.float(), w.float().t())
print("gelu fused: h err", ((h.float() - href).abs().max() / href.abs().max()).item(), "a err vs gelu(h_bf16)", (g.float() - F.gelu(h.float())).abs().max().item())
us = t(lambda: nnops.gemm_nt_gelu(a, w, bias))
us2 = t(lambda: F.gelu(torch.addmm(bias, a, w.t())))
print(f"FFN1+GELU: own fused {us:.1f} us | torch addmm+gelu {us2:.1f} us")
gf = torch.randn(M, 768, device="cuda").bfloat16(); w2t = (torch.randn(3072, 768, device="cuda") * 0.05).bfloat16()
out = nnops.gemm_nt_dgelu(gf, w2t, h)
hr = h.float().requires_grad_(True); F.gelu(hr).backward(gf.float() @ w2t.float().t())
print("dgelu fused err", ((out.float() - hr.grad).abs().max() / hr.grad.abs().max()).item())
us = t(lambda: nnops.gemm_nt_dgelu(gf, w2t, h))
print(f"FFN2-dgrad+dGELU: own fused {us:.1f} us")
