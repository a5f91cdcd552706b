"""Time the fused VQ kernels with HIP events (scratch tool; bench.py holds the judged measurement)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
import kvq  # noqa: E402


def timeit(fn, iters=50, warm=10):
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


def main():
    torch.manual_seed(0)
    for (N, K, D) in [(8192, 512, 768), (65536, 512, 768), (8192, 8192, 768), (256, 512, 768)]:
        for dt in (torch.float32, torch.bfloat16):
            z = torch.randn(N, D, device="cuda").to(dt)
            E = torch.randn(K, D, device="cuda")
            us = timeit(lambda: kvq.vector_quantize(z, E, 0.25))
            fl = 2.0 * N * K * D
            by = N * (2 * D * z.element_size() + 8) + K * D * 4
            print(f"fwd N={N:6d} K={K:5d} D={D} {str(dt)[6:]:9s}: {us:9.1f} us  {fl/us/1e6:7.1f} TFLOP/s  {by/us/1e3:8.1f} GB/s alg")
            zr = z.clone().requires_grad_(True)
            Er = E.clone().requires_grad_(True)
            loss, z_q, *_ = kvq.vector_quantize(zr, Er, 0.25)
            g = torch.randn_like(z_q)

            def bwd():
                torch.autograd.grad([loss, z_q], [zr, Er], [torch.ones_like(loss), g], retain_graph=True)
            print(f"bwd {'':38s}: {timeit(bwd, 20, 5):9.1f} us")


if __name__ == "__main__":
    main()
