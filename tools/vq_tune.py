"""A/B the tuning knobs of the tiled VQ distance kernel in ONE process, interleaved rounds (kernel-only HIP-event times)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
import kvq  # noqa: E402
from kvq import _ffi  # noqa: E402

lib = _ffi.lib()


def kernel_us(fn, iters=20):
    fn(); torch.cuda.synchronize()
    lib.kvq_prof_enable(iters)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_float * iters)()
    n = lib.kvq_prof_read(buf, iters)
    lib.kvq_prof_enable(0)
    v = sorted(buf[i] for i in range(n))
    return v[len(v) // 2] * 1e3, v[0] * 1e3


torch.manual_seed(0)
D = 768
for (N, K) in [(8192, 512), (65536, 512), (8192, 8192)]:
    z = torch.randn(N, D, device="cuda").bfloat16()
    E = torch.randn(K, D, device="cuda")
    ref = None
    for rnd in range(3):
        for (kc, packed) in ((32, 0), (32, 1), (32, 2)):
            for prio in (1,):
                lib.kvq_vq_set_tuning(kc, prio, packed)
                med, mn = kernel_us(lambda: kvq.vector_quantize(z, E, 0.25))
                idx = kvq.vector_quantize(z, E, 0.25)[3]
                if ref is None:
                    ref = idx.clone()
                assert torch.equal(idx, ref)
                print(f"N={N} K={K} round {rnd} kc={kc} packed={packed} prio={prio}: median {med:7.1f} us  min {mn:7.1f} us  {2.0*N*K*D/med/1e6:6.1f} TF", flush=True)
lib.kvq_vq_set_tuning(32, 1, 2)
