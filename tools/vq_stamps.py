#!/usr/bin/env python3
"""Where a workgroup of the VQ distance kernel spends its time: in-kernel stamps of the DIAGNOSTIC library (tools/build_diag.sh,
-DKVQ_VQ_DIAG).  Wave 0 of every workgroup stamps s_memtime at entry (0), after the prologue (1), before / after every stage
barrier (2 + 2 st / 3 + 2 st) and at the end (60).  usage: vq_stamps.py [K]   (N = 8192, D = 768, bf16 tokens)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from kvq import _ffi  # noqa: E402

_ffi.LIB_PATH = os.path.join(ROOT, "kindergarten-vq-vae_amd", "lib", "diag", "libkvq.so")
import kvq  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, D = 8192, 768
lib = _ffi.lib()
lib.kvq_vq_diag_set_buffer.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
zs = [torch.randn(N, D, device="cuda").bfloat16() for _ in range(4)]
E = torch.randn(K, D, device="cuda")
nwg = (N // 64) * ((K + 127) // 128)
buf = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
with torch.no_grad():
    for i in range(3):
        kvq.vector_quantize(zs[i], E, 0.25)
    torch.cuda.synchronize()
    lib.kvq_prof_enable(1)
    assert lib.kvq_vq_diag_set_buffer(buf.data_ptr()) == 0
    kvq.vector_quantize(zs[3], E, 0.25)
    torch.cuda.synchronize()
    lib.kvq_vq_diag_set_buffer(None)
ev = (ctypes.c_float * 1)()
lib.kvq_prof_read(ev, 1)
lib.kvq_prof_enable(0)
t = buf.cpu().numpy().astype(np.float64)
entry, pro, acc_stage, acc_bar, last_bar, end, rt = (t[:, i] for i in range(7))
span_rt = (rt.max() - rt.min()) * 0.01                     # us between the first and the last workgroup's end (100 MHz counter)
print(f"N={N} K={K}: {nwg} workgroups; kernel {ev[0] * 1e3:.1f} us (event pair around the launch, diagnostic build)")
cyc = lambda a: f"median {np.median(a):8.0f} cycles (p10 {np.percentile(a, 10):8.0f}, p90 {np.percentile(a, 90):8.0f})"
print("  workgroup lifetime           ", cyc(end - entry))
print("  entry -> prologue done       ", cyc(pro - entry), " first z tile in LDS, first codebook fragments")
print("  24 stages, between barriers  ", cyc(acc_stage), f" = {np.median(acc_stage) / 24:.0f} per stage (32 MFMAs of one wave alone: 2048)")
print("  24 stages, inside the barrier", cyc(acc_bar), f" = {np.median(acc_bar) / 24:.0f} per stage (thread 0's wave waiting for the other three; one barrier per PAIR of stages since round 3)")
print("  last barrier -> end          ", cyc(end - last_bar), " arg-min, LDS merge, atomicMin")
print(f"  if the matrix pipe never idled a stage would take 4096 cycles with two waves per SIMD; measured {np.median(acc_stage + acc_bar) / 24:.0f}")
