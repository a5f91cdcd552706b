#!/usr/bin/env python3
"""Per forward GEMM of the step: own bf16 kernel (rule-picked tile; FFN1 with its GELU epilogue) against the fp8 kernel + what fp8
needs around it (the activation's quantisation pass; for FFN1 the separate GELU kernel) -- which GEMMs does fp8 pay for?
usage: gemm2_probe_fp8_own.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402
from kvq._ffi import check, lib, stream_ptr  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev, T = "cuda", 8192


def bench(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rnd(*s):
    return torch.randn(s, device=dev).to(torch.bfloat16)


print("| forward GEMM | own bf16 us | fp8 GEMM us | + quantise x us | (+ GELU kernel us) | fp8 total us |")
print("|---|---|---|---|---|---|")
for name, n, k, gelu in [("attention output / cross-q 768x768", 768, 768, False), ("QKV 2304x768", 2304, 768, False),
                         ("FFN1 3072x768 + GELU", 3072, 768, True), ("FFN2 768x3072", 768, 3072, False),
                         ("all-layer cross-K/V 18432x768", 18432, 768, False), ("LM head 30528x768", 30528, 768, False)]:
    x, w, b = rnd(T, k), rnd(n, k), rnd(n)
    x8, sx = nnops.fp8_quantize(x)
    w8, sw = nnops.fp8_quantize(w)
    st = torch.zeros(lib().kvq_fp8_state_floats(), dtype=torch.float32, device=dev)
    st[0] = 1.0
    out = torch.empty((T, n), device=dev, dtype=torch.bfloat16)
    h = torch.empty_like(out)
    fns = {"own": (lambda: nnops.gemm_gelu(x, w, b)) if gelu else (lambda: nnops.gemm(x, w, "nt", bias=b, out=out)),
           "fp8": lambda: nnops.gemm_fp8_nt(x8, w8, sx, sw, bias=b, out=out),
           "quant": lambda: check(lib().kvq_fp8_quantize_delayed(x.data_ptr(), T, k, k, x8.data_ptr(), st.data_ptr(), stream_ptr()), "q"),
           "gelu": (lambda: nnops.gelu_fwd(out)) if gelu else None}
    res = {kk: [] for kk, f in fns.items() if f is not None}
    for _ in range(rounds):
        for kk in res:
            res[kk].append(bench(fns[kk]))
    med = {kk: sorted(v)[len(v) // 2] for kk, v in res.items()}
    tot = med["fp8"] + med["quant"] + med.get("gelu", 0.0)
    print(f"| {name} | {med['own']:.1f} | {med['fp8']:.1f} | {med['quant']:.1f} | {med.get('gelu', 0.0):.1f} | {tot:.1f} |", flush=True)
