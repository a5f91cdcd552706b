#!/usr/bin/env python3
"""Phases of the 32-token attention kernels from inside (DIAGNOSTIC library, -DKVQ_NN_DIAG): every wave stores s_memtime at entry, when
its operand rows have landed, when the softmax part is done, and at its end.  8192 tokens x 12 heads, bf16, dropout 0.1, cold buffers."""
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

lib = _ffi.lib()
lib.kvq_nn_diag_set_buffer.argtypes = [ctypes.c_void_p]
dev, nh, H, B, S = "cuda", 12, 768, 256, 32
g = torch.Generator(device=dev).manual_seed(0)
qkv = [torch.randn(B * S, 3 * H, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(6)]
go = torch.randn(B * S, H, device=dev, dtype=torch.bfloat16, generator=g)
gq = torch.empty_like(qkv[0])
mask = torch.ones(B, S, dtype=torch.int64, device=dev)
buf = torch.zeros((B * nh, 8), dtype=torch.int64, device=dev)


def fwd(i):
    t = qkv[i]
    return nnops.attn_fwd(t[:, :H], t[:, H:2 * H], t[:, 2 * H:], mask, B, nh, S, S, False, 0.1, 9, 3)


def bwd(i):
    t = qkv[i]
    nnops.attn_bwd(t[:, :H], t[:, H:2 * H], t[:, 2 * H:], mask, go, B, nh, S, S, False, 0.1, 9, 3, gq[:, :H], gq[:, H:2 * H], gq[:, 2 * H:])


for name, fn in (("forward", fwd), ("backward", bwd)):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    buf.zero_()
    assert lib.kvq_nn_diag_set_buffer(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(5); e1.record()
    torch.cuda.synchronize()
    lib.kvq_nn_diag_set_buffer(None)
    t = buf.cpu().numpy().astype(np.float64)
    t0, t1, t2, t3, rt = (t[:, i] for i in range(5))
    span_us = (rt.max() - rt.min()) * 0.01
    clk = np.median(t3 - t0) / 1e3                                       # cycles per us needs a time base: the 100 MHz counter's span
    life = t3 - t0
    print(f"{name}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us (event pair), wave ends spread over {span_us:.1f} us; wave lifetime median {np.median(life):.0f} cycles "
          f"(p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
    for label, d in (("entry -> operand rows landed (loads, mask, dropout bits drawn meanwhile)", t1 - t0), ("rows landed -> softmax / dS done", t2 - t1),
                     ("remaining products + stores issued", t3 - t2)):
        print(f"   {label:78s} median {np.median(d):7.0f} cycles = {100 * np.median(d / life):4.1f} % of the wave")
    # s_memtime is per XCD (the counters are not aligned across the eight dies); the 100 MHz s_memrealtime taken at the wave's END is:
    # a wave's entry on that clock = its end minus its lifetime at the kernel's shader clock (~2.07 GHz)
    start_us = rt * 0.01 - life / 2070.0
    st = np.sort(start_us - start_us.min())
    print(f"   wave ENTRIES (100 MHz clock): median wave entered {np.median(st):.1f} us after the first, 90 % by {np.percentile(st, 90):.1f} us, last at {st[-1]:.1f} us "
          f"-> {len(st) / max(st[-1], 1e-9):.0f} workgroups dispatched per us")
