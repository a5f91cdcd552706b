"""Which aten ops (outside the GEMMs and the kvq kernels) does one engine step launch?  torch.profiler table, scratch tool."""
import os, sys, torch
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
ids, mask = (t.to(dev) for t in random_token_batch(256, 32, gen))
for i in range(4): eng.train_step(ids, mask)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(3): eng.train_step(ids, mask)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=60))
