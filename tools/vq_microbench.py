"""Time the fused VQ kernels (scratch tool; bench.py holds the judged measurement).
Forward: kernel-only time from the library's event hook (kvq_prof_*) next to the end-to-end call time."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
import kvq  # noqa: E402
from kvq import _ffi  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def kernel_us(fn, iters=30):
    lib = _ffi.lib()
    fn(); torch.cuda.synchronize()
    lib.kvq_prof_enable(iters)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_float * iters)()
    n = lib.kvq_prof_read(buf, iters)
    lib.kvq_prof_enable(0)
    v = sorted(buf[i] for i in range(n))
    return v[len(v) // 2] * 1e3 if v else float("nan")


def run(tag, N, K, D, dt, z, E):
    run1(tag, N, K, D, dt, z, E)


def run1(tag, N, K, D, dt, z, E):
    us = timeit(lambda: kvq.vector_quantize(z, E, 0.25))
    kus = kernel_us(lambda: kvq.vector_quantize(z, E, 0.25))
    fl = 2.0 * N * K * D
    by = N * (2 * D * z.element_size() + 8) + K * D * 4
    zr = z.clone().requires_grad_(True)
    Er = E.clone().requires_grad_(True)
    loss, z_q, *_ = kvq.vector_quantize(zr, Er, 0.25)
    g = torch.randn_like(z_q)
    bus = timeit(lambda: torch.autograd.grad([loss, z_q], [zr, Er], [torch.ones_like(loss), g], retain_graph=True), 20, 3)
    print(f"{tag:12s} N={N:6d} K={K:5d} {str(dt)[6:]:8s} fwd call {us:8.1f} us | kernel {kus:8.1f} us = {fl/kus/1e6:6.1f} TFLOP/s "
          f"{by/kus/1e3:7.1f} GB/s | bwd call {bus:8.1f} us", flush=True)


def main():
    torch.manual_seed(0)
    D = 768
    for (N, K) in [(8192, 512), (65536, 512), (8192, 8192), (8192, 10)]:
        for dt in (torch.float32, torch.bfloat16):
            E = torch.randn(K, D, device="cuda")
            run("spread", N, K, D, dt, torch.randn(N, D, device="cuda").to(dt), E)
            run("collapsed", N, K, D, dt, (E[3] + 0.05 * torch.randn(N, D, device="cuda")).to(dt), E)


if __name__ == "__main__":
    main()
