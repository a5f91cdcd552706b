"""Feasibility probe: capture one whole train step (values baked in) in a HIP graph and time replays against eager steps."""
import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd")); sys.path.insert(0, ROOT)
from dsentences.synthetic import random_token_batch
from models.shelgon3.Shelgon import Shelgon
from models.shelgon3.VectorQuantizer import VectorQuantizer
from kvq.engine import TrainEngine
from kvq import nnops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
vq = VectorQuantizer(n_e=512, e_dim=768, beta=0.25); vq.materialize_min_encodings = False
model = Shelgon("bert-base-uncased", vq, "bert-base-uncased", None, compute_dtype=torch.bfloat16).to(dev)
model.set_mode("full"); model.train()
eng = TrainEngine(model, lr=1e-4, weight_decay=0.0, amsgrad=False, milestones=[10000, 20000], gamma=0.1)
import torch.nn.functional as F
def emb_fwd(prefix, cfg, ids, training, word_rows=None):
    fl = eng.flat
    B, S = ids.shape
    word = fl.w(prefix + "word", rows=word_rows) if word_rows else fl.w(prefix + "word")
    y = F.embedding(ids.reshape(-1), word)
    pt = (fl.w(prefix + "pos")[:S] + fl.w(prefix + "type")[0]).repeat(B, 1)
    out, pre, mean, rstd = nnops.ln_fwd(y, pt, fl.w32(prefix + "ln.w"), fl.w32(prefix + "ln.b"), cfg.layer_norm_eps)
    keep = torch.full(out.shape, 1.0 / 0.9, dtype=out.dtype, device=out.device)
    return out * keep, (ids, pre, mean, rstd, keep)
eng._emb_fwd = emb_fwd
gen = torch.Generator().manual_seed(69)
ids, mask = (t.to(dev) for t in random_token_batch(256, 32, gen))
for i in range(5): eng.train_step(ids, mask)
torch.cuda.synchronize()
def timeit(fn, K=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
print(f"eager  {timeit(lambda: eng.train_step(ids, mask)):.2f} ms/step", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = eng.train_step(ids, mask)
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(3): g.replay()
print(f"graph  {timeit(g.replay):.2f} ms/step  loss {float(out['loss_recon']):.4f}", flush=True)
print(f"eager  {timeit(lambda: eng.train_step(ids, mask)):.2f} ms/step", flush=True)
