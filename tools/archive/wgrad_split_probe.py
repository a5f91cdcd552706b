"""Split-K factor sweep for the weight-gradient GEMMs gW[M,N] = gy[Ntok,M]^T x[Ntok,N] (tuned online with TunableOp) plus the
cost of summing the slabs inside kvq_reduce_batch.  Prints the best S per shape (feeds TrainEngine._SPLITS)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(30)
if hasattr(tn, "write_file_on_exit"): tn.write_file_on_exit(False)
Ntok = 8192
def t(fn, n=30):
    for _ in range(4): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    gy = torch.randn(Ntok, M, device="cuda").bfloat16(); x = torch.randn(Ntok, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = []
    for S in (1, 2, 4, 8, 16, 32):
        if S == 1:
            g = t(lambda: torch.mm(gy.t(), x, out=out)); r = 0.0
        else:
            f = lambda: torch.bmm(gy.view(S, Ntok // S, M).transpose(1, 2), x.view(S, Ntok // S, N))
            g = t(f)
            part = f()
            item = [nnops.reduce_item(part, out, S, M * N, M * N)]
            r = t(lambda: nnops.reduce_batch(item))
        # inside the layer's batched launch the slab sum overlaps with the other items: count its HBM time, not its latency
        r_stream = S * M * N * 2 / 5.5e12 * 1e6 if S > 1 else 0.0
        res.append((S, g, r, g + r_stream))
    best = min(res, key=lambda a: a[3])
    print(f"({M},{N}): " + "  ".join(f"S={S}: {g:.1f}+{r:.1f}us" for S, g, r, _ in res) + f"   -> best S={best[0]} ({best[3]:.1f} us)", flush=True)
