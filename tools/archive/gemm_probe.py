"""Probe hipBLASLt (through torch) on the training step's GEMM shapes: plain mm vs manual split-K via bmm (scratch tool)."""
import torch


def t(fn, it=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


K = 8192
for (M, N) in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (1536, 768), (30528, 768)]:
    gy = torch.randn(K, M, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    us = t(lambda: torch.mm(gy.t(), x, out=out))
    line = f"wgrad M={M:5d} N={N:5d}: mm {us:7.1f} us {fl/us/1e6:6.0f} TF |"
    for S in (2, 4, 8, 16):
        part = torch.empty(S, M, N, device="cuda", dtype=torch.bfloat16)
        def f():
            torch.bmm(gy.view(S, K // S, M).transpose(1, 2), x.view(S, K // S, N), out=part)
            torch.sum(part, 0, out=out)
        us2 = t(f)
        line += f" S={S}: {us2:6.1f} ({fl/us2/1e6:5.0f})"
    # f32 accumulate across splits via baddbmm chain is not available; report also x.t() @ gy (transposed output)
    out2 = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    us3 = t(lambda: torch.mm(x.t(), gy, out=out2))
    line += f" | swapped {us3:6.1f} ({fl/us3/1e6:5.0f})"
    print(line, flush=True)
