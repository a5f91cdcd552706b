#!/usr/bin/env python3
"""Three kernels for a memory-system counter comparison (tools/run_gemm_diag.sh): the two-layer grouped weight gradient (TN, 256x256),
an input gradient with a long contraction (NN, 128x192, K = 18432), the LM-head forward (NT, 256x256)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

T, dev = 8192, "cuda"
rnd = lambda *s: torch.randn(s, device=dev).to(torch.bfloat16)  # noqa: E731
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * 2
gys = [rnd(T, m) for m, _ in shapes]
xs = [rnd(T, n) for _, n in shapes]
outs = [torch.empty((m, n), device=dev, dtype=torch.bfloat16) for m, n in shapes]
probs = [nnops.gemm_problem(g, x, o, "tn") for g, x, o in zip(gys, xs, outs)]
a, b = rnd(T, 18432), rnd(18432, 768)
c = torch.empty((T, 768), device=dev, dtype=torch.bfloat16)
x, w = rnd(T, 768), rnd(30528, 768)
lo = torch.empty((T, 30528), device=dev, dtype=torch.bfloat16)
torch.cuda.synchronize()
for _ in range(3):
    nnops.gemm_grouped(probs, "tn", "256x256")
    nnops.gemm(a, b, "nn", out=c, tile="128x192")
    nnops.gemm(x, w, "nt", out=lo, tile="256x256")
torch.cuda.synchronize()
