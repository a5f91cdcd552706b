#!/usr/bin/env python3
"""Every GEMM shape of the benchmarked step, own kernel (csrc/kvq_gemm2.hip) and vendor library, a few launches each with a
marker kernel in between -- the workload of the rocprofv3 --pmc passes behind profiles/r02_gemm_pmc.md
(tools/run_gemm_pmc.sh; summary: tools/make_gemm_pmc_summary.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import nnops  # noqa: E402

T, REP = 8192, 4
dev = "cuda"
rnd = lambda *s: torch.randn(s, device=dev).to(torch.bfloat16)  # noqa: E731
marker = torch.zeros(64, device=dev, dtype=torch.int32)
plan = []


def mark(label):
    plan.append(label)
    marker.fill_(len(plan))          # one elementwise fill kernel = separator in the dispatch trace


def both(label, layout, M, N, K, tile):
    if layout == "nt":
        a, b = rnd(M, K), rnd(N, K); lib = lambda: torch.mm(a, b.t())  # noqa: E702,E731
    elif layout == "nn":
        a, b = rnd(M, K), rnd(K, N); lib = lambda: torch.mm(a, b)  # noqa: E702,E731
    else:
        a, b = rnd(K, M), rnd(K, N); lib = lambda: torch.mm(a.t(), b)  # noqa: E702,E731
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    for who, fn in (("own " + tile, lambda: nnops.gemm(a, b, layout, out=out, tile=tile)), ("lib", lib)):
        fn(); torch.cuda.synchronize()
        mark(dict(label=label, who=who, layout=layout, M=M, N=N, K=K, flops=2.0 * M * N * K,
                  alg_bytes=2 * (M * K + N * K + M * N)))
        for _ in range(REP):
            fn()
    torch.cuda.synchronize()


# tiles as the engine's tables select them at round 3 ("p" = the persistent tile loop)
for n, k, t in [(768, 768, "128x192"), (2304, 768, "128x192p"), (3072, 768, "256x192"), (768, 3072, "128x192"), (18432, 768, "256x256p"),
                (30528, 768, "256x256")]:
    both("fwd", "nt", T, n, k, t)
for n, k, t in [(768, 768, "128x192"), (768, 2304, "128x192"), (768, 3072, "128x192"), (3072, 768, "256x192"), (768, 18432, "128x192"),
                (768, 30528, "128x192")]:
    both("dgrad", "nn", T, n, k, t)
for m, n, t in [(18432, 768, "256x256"), (30528, 768, "128x256")]:
    both("wgrad", "tn", m, n, T, t)


def own_only(label, who, fn, flops, alg_bytes, layout, M, N, K):
    fn(); torch.cuda.synchronize()  # noqa: E702
    mark(dict(label=label, who=who, layout=layout, M=M, N=N, K=K, flops=flops, alg_bytes=alg_bytes))
    for _ in range(REP):
        fn()
    torch.cuda.synchronize()


# FFN1 forward with the activation in the epilogue (two outputs), FFN2 input gradient with GELU' and the bias partials
x, w1, b1 = rnd(T, 768), rnd(3072, 768), rnd(3072)
own_only("fwd+gelu", "own fused 256x192", lambda: nnops.gemm_gelu(x, w1, b1, tile="256x192"), 2.0 * T * 3072 * 768,
         2 * (T * 768 + 3072 * 768 + 2 * T * 3072), "nt", T, 3072, 768)
gf, w2, h = rnd(T, 768), rnd(768, 3072), rnd(T, 3072)
own_only("dgrad*gelu'", "own fused 256x192", lambda: nnops.gemm_dgelu(gf, w2, h, tile="256x192"), 2.0 * T * 3072 * 768,
         2 * (T * 768 + 3072 * 768 + 2 * T * 3072), "nn", T, 3072, 768)
# the engine's grouped weight-gradient launch: TWO encoder layers (QKV, O, FFN1, FFN2 each) as 216 tiles of 256 x 256
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * 2
gys = [rnd(T, m) for m, _ in shapes]; xs = [rnd(T, n) for _, n in shapes]  # noqa: E702
outs = [torch.empty((m, n), device=dev, dtype=torch.bfloat16) for m, n in shapes]
probs = [nnops.gemm_problem(g, x_, o, "tn") for g, x_, o in zip(gys, xs, outs)]
own_only("wgrad-2-layers", "own grouped 256x256", lambda: nnops.gemm_grouped(probs, "tn", "256x256"),
         sum(2.0 * T * m * n for m, n in shapes), sum(2 * (T * m + T * n + m * n) for m, n in shapes), "tn", 2 * 6912, 0, T)
mark(dict(label="end", who="", layout="", M=0, N=0, K=0, flops=0, alg_bytes=0))
out_dir = os.environ.get("KVQ_PMC_PLAN_DIR", "gpurun_out")
json.dump(plan, open(os.path.join(out_dir, "gemm_pmc_plan.json"), "w"))
