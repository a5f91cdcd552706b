#!/usr/bin/env python3
"""Where a tile of the GEMM family spends its time: in-kernel phase stamps of the DIAGNOSTIC library (tools/build_diag.sh).
usage: gemm2_stamps.py layout M N K tile [cold]      e.g.  gemm2_stamps.py nt 8192 30528 768 256x256
Stamps (wave 0 of every workgroup): entry, ring issued, first k-tile landed, main loop done, C tile in LDS, stores issued, stores done."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import _ffi  # noqa: E402

_ffi.LIB_PATH = os.path.join(ROOT, "kindergarten-vq-vae_amd", "lib", "diag", "libkvq.so")
from kvq import nnops  # noqa: E402

layout, M, N, K, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
extra = sys.argv[6] if len(sys.argv) > 6 else ""
dev = "cuda"
lib = _ffi.lib()
import ctypes
lib.kvq_diag_set_buffer.argtypes = [ctypes.c_void_p]


def rnd(*s):
    return torch.randn(s, device=dev).to(torch.bfloat16)


if layout == "nt":
    a, b = rnd(M, K), rnd(N, K)
elif layout == "nn":
    a, b = rnd(M, K), rnd(K, N)
else:
    a, b = rnd(K, M), rnd(K, N)
out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
bias = rnd(N)
bm, bn = (int(v) for v in tile.rstrip("hp").split("x"))      # ("128x192h": the four-wave tile, two workgroups per CU)
ntiles = -(-M // bm) * -(-N // bn)
buf = torch.zeros((ntiles, 16), dtype=torch.int64, device=dev)
trash = torch.empty(1 << 28, dtype=torch.uint8, device=dev)


def call():
    if extra == "gelu":
        return nnops.gemm_gelu(a, b, bias, tile=tile)
    return nnops.gemm(a, b, layout, bias=bias if layout == "nt" else None, out=out, tile=tile)


for _ in range(3):
    call()
torch.cuda.synchronize()
if extra == "cold":
    trash.fill_(1)
assert lib.kvq_diag_set_buffer(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
call()
e1.record()
torch.cuda.synchronize()
assert lib.kvq_diag_set_buffer(None) == 0
s = buf.cpu().numpy().astype(np.int64)
t_entry, rt0, hw, t_issued, t_first, t_loop, t_lds, t_st, t_done, rt1 = (s[:, i] for i in range(10))
tot_rt = (rt1 - rt0) * 0.01                                  # us (100 MHz)
cyc = (t_done - t_entry).astype(np.float64)
clk = cyc / np.maximum(tot_rt, 1e-9) / 1e3                   # GHz
print(f"{layout} M={M} N={N} K={K} tile {tile} {extra}: {ntiles} tiles, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (event pair, with launch), "
      f"shader clock {np.median(clk):.2f} GHz")
span = (rt1.max() - rt0.min()) * 0.01
print(f"  first entry -> last done: {span:.1f} us; workgroup lifetime median {np.median(tot_rt):.2f} us (p10 {np.percentile(tot_rt, 10):.2f}, p90 {np.percentile(tot_rt, 90):.2f})")


def us(d):
    return d / (np.median(clk) * 1e3)


t_setup, t_k0 = s[:, 10], s[:, 11]
d_vm, d_bar = s[:, 12].astype(np.float64), s[:, 13].astype(np.float64)
nkt = K // 64
loop = (t_loop - t_first).astype(np.float64)
print(f"  k loop of wave 0: {np.median(loop) / nkt:7.0f} cycles per k-tile; of the loop {100 * np.median(d_vm / loop):4.1f} % in the vmcnt wait before the barrier "
      f"(own DMA pieces not landed), {100 * np.median(d_bar / loop):4.1f} % in the barrier (other waves)   [the two timer reads per k-tile cost ~100 cycles themselves]")
for name, d in (("  entry -> setup done (tile lookup, bias, offsets)", t_setup - t_entry), ("  setup done -> k-tile 0 issued", t_k0 - t_setup),
                ("  k-tile 0 issued -> ring issued", t_issued - t_k0), ("entry -> ring issued", t_issued - t_entry), ("ring issued -> first k-tile landed", t_first - t_issued),
                ("main loop", t_loop - t_first), ("accumulators -> LDS (+barrier)", t_lds - t_loop),
                ("LDS -> store instructions issued", t_st - t_lds), ("stores issued -> stores done", t_done - t_st)):
    print(f"  {name:38s} median {np.median(us(d)):6.2f} us   p10 {np.percentile(us(d), 10):6.2f}   p90 {np.percentile(us(d), 90):6.2f}")
# gaps between consecutive workgroups on one CU: key = (xcc, hw_id without wave/simd bits)
key = ((hw >> 32) << 32) | ((hw & 0xffffffff) & ~0xff)
gaps, per_cu = [], []
for k in np.unique(key):
    sel = np.nonzero(key == k)[0]
    o = sel[np.argsort(rt0[sel])]
    per_cu.append(len(o))
    for i in range(1, len(o)):
        gaps.append((rt0[o[i]] - rt1[o[i - 1]]) * 0.01)
print(f"  CUs seen {len(per_cu)}, tiles per CU min {min(per_cu)} max {max(per_cu)}")
if gaps:
    g = np.array(gaps)
    print(f"  previous workgroup's stores done -> next workgroup's entry on the same CU: median {np.median(g):.2f} us  p10 {np.percentile(g, 10):.2f}  p90 {np.percentile(g, 90):.2f}")
print(f"  start skew: first entries spread over {(np.sort(rt0)[min(255, len(rt0) - 1)] - rt0.min()) * 0.01:.2f} us (first 256 workgroups)")
