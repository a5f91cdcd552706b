"""Where does a decoder layer's kvq_reduce_batch launch spend its time?  (scratch tool)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
from kvq import nnops as ops
dev = "cuda"
H = 768
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def slab(S, M, N):
    part = torch.randn(S, M, N, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    return ops.reduce_item(part, out, S, M * N, M * N), (part, out)
def tree(P, C, ld=None):
    part = torch.randn(P, ld or C, device=dev); out = torch.empty(C, device=dev, dtype=torch.bfloat16)
    return ops.reduce_item(part, out, P, C, ld or C), (part, out)
keep = []
def mk(lst):
    items = []
    for it, k in lst:
        items.append(it); keep.append(k)
    return items
slabs = mk([slab(8, 2304, 768), slab(16, 768, 768), slab(16, 768, 768), slab(16, 768, 768), slab(4, 3072, 768), slab(4, 768, 3072)])
lns = mk([tree(1024, 2304), tree(1024, 2304), tree(1024, 2304)])
gelu = mk([tree(1024, 3072)])
attn = mk([tree(256, 2304), tree(256, 768)])
for name, items in (("slabs (6 wgrads, 123 MB)", slabs), ("3 LN partials (28 MB)", lns), ("gelu partial (12.6 MB)", gelu), ("attn partials (3 MB)", attn),
                    ("all", slabs + lns + gelu + attn)):
    print(f"{name:28s} {t(lambda: ops.reduce_batch(items)):7.1f} us")
