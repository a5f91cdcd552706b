#!/usr/bin/env python3
"""Per-tile stamps of the persistent GEMM (diagnostic library): tile start, after k-tile 0 / 1, main loop done, stores issued.
usage: gemm3_stamps.py M N K tile"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import _ffi  # noqa: E402

_ffi.LIB_PATH = os.path.join(ROOT, "kindergarten-vq-vae_amd", "lib", "diag", "libkvq.so")
from kvq import nnops  # noqa: E402

M, N, K, tile = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
lib = _ffi.lib()
lib.kvq_diag_set_buffer.argtypes = [ctypes.c_void_p]
dev = "cuda"
a = torch.randn((M, K), device=dev).to(torch.bfloat16)
b = torch.randn((N, K), device=dev).to(torch.bfloat16)
bias = torch.randn((N,), device=dev).to(torch.bfloat16)
out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
bm, bn = (int(v) for v in tile.split("x"))
ntiles = -(-M // bm) * -(-N // bn)
buf = torch.zeros((ntiles, 16), dtype=torch.int64, device=dev)
for _ in range(3):
    nnops.gemm(a, b, "nt", bias=bias, out=out, tile=tile + "p")
torch.cuda.synchronize()
assert lib.kvq_diag_set_buffer(buf.data_ptr()) == 0
nnops.gemm(a, b, "nt", bias=bias, out=out, tile=tile + "p")
torch.cuda.synchronize()
lib.kvq_diag_set_buffer(None)
s = buf.cpu().numpy().astype(np.int64)
t0, rt0, hw, t1, t2, t5, t7, rt9 = s[:, 0], s[:, 1], s[:, 2], s[:, 3], s[:, 4], s[:, 5], s[:, 7], s[:, 9]
life = (rt9 - rt0) * 0.01
clk = np.median((t7 - t0) / np.maximum(life, 1e-9) / 1e3)
print(f"persistent nt M={M} N={N} K={K} tile {tile}: {ntiles} tiles, span {(rt9.max() - rt0.min()) * 0.01:.1f} us, clock {clk:.2f} GHz")
print(f"  tile start -> stores issued: median {np.median(life):.2f} us p10 {np.percentile(life, 10):.2f} p90 {np.percentile(life, 90):.2f}")
for nm, d in (("k-tile 0", t1 - t0), ("k-tile 1", t2 - t1), ("k-tiles 2..", t5 - t2), ("epilogue (issue)", t7 - t5)):
    u = d / (clk * 1e3)
    print(f"  {nm:22s} median {np.median(u):6.2f} us  p10 {np.percentile(u, 10):6.2f}  p90 {np.percentile(u, 90):6.2f}")
# first tile of a workgroup vs later tiles
G = 256
first = np.arange(ntiles) < G
for nm, sel in (("first tile of a workgroup", first), ("later tiles", ~first)):
    if sel.any():
        print(f"  {nm}: lifetime median {np.median(life[sel]):.2f} us; k-tile 0 {np.median((t1 - t0)[sel] / (clk * 1e3)):.2f}, k-tile 1 {np.median((t2 - t1)[sel] / (clk * 1e3)):.2f}, "
              f"rest {np.median((t5 - t2)[sel] / (clk * 1e3)):.2f}, epilogue {np.median((t7 - t5)[sel] / (clk * 1e3)):.2f}")
