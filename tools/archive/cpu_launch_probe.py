"""Is the train step CPU-launch-bound?  Enqueue K steps without synchronising and compare host enqueue time with the GPU time."""
import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd")); sys.path.insert(0, ROOT)
from dsentences.synthetic import random_token_batch
from models.shelgon3.Shelgon import Shelgon
from models.shelgon3.VectorQuantizer import VectorQuantizer
from kvq.engine import TrainEngine
dev = torch.device("cuda", 0)
torch.manual_seed(0)
vq = VectorQuantizer(n_e=512, e_dim=768, beta=0.25); vq.materialize_min_encodings = False
model = Shelgon("bert-base-uncased", vq, "bert-base-uncased", None, compute_dtype=torch.bfloat16).to(dev)
model.set_mode("full"); model.train()
eng = TrainEngine(model, lr=1e-4, weight_decay=0.0, amsgrad=False, milestones=[10000, 20000], gamma=0.1)
gen = torch.Generator().manual_seed(69)
pool = [tuple(t.to(dev) for t in random_token_batch(256, 32, gen)) for _ in range(4)]
for i in range(5): eng.train_step(*pool[i % 4])
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for i in range(K): eng.train_step(*pool[i % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/K:.2f} ms/step; total {1e3*(t2-t0)/K:.2f} ms/step; final sync wait {1e3*(t2-t1):.2f} ms")
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(3): eng.train_step(*pool[i % 4])
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
