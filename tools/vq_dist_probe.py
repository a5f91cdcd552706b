"""The distance kernel alone (HIP-event pairs the library records around its launch): us per launch at N = 8192, D = 768."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
import torch
import kvq
from kvq import _ffi

lib = _ffi.lib()
N, D = 8192, 768
for K in (512, 8192):
    torch.manual_seed(0)
    zs = [torch.randn(N, D, device="cuda").bfloat16() for _ in range(8)]
    E = torch.randn(K, D, device="cuda")
    reps = 48 if K == 512 else 16
    with torch.no_grad():
        for i in range(4):
            kvq.vector_quantize(zs[i], E, 0.25)
        torch.cuda.synchronize()
        lib.kvq_prof_enable(reps)
        for i in range(reps):
            kvq.vector_quantize(zs[i % 8], E, 0.25)
        torch.cuda.synchronize()
    buf = (ctypes.c_float * reps)()
    n = lib.kvq_prof_read(buf, reps)
    lib.kvq_prof_enable(0)
    v = sorted(buf[i] * 1e3 for i in range(n))
    fl = 2.0 * N * K * D
    print(f"K = {K}: {n} launches, median {v[n // 2]:.1f} us, mean {sum(v) / n:.1f} us, min {v[0]:.1f} -> {fl / (sum(v) / n) / 1e6:.1f} TF = {fl / (sum(v) / n) / 1e6 / 157.3:.3f} of peak", flush=True)
