"""Peak device memory of the bench workload in graph and eager mode (scratch tool)."""
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
eng = TrainEngine(model, lr=1e-4)
gen = torch.Generator().manual_seed(69)
ids, mask = (t.to(dev) for t in random_token_batch(256, 32, gen))
for i in range(6):
    eng.train_step(ids, mask)
    torch.cuda.synchronize()
    print(f"step {i+1}: allocated {torch.cuda.memory_allocated()/2**30:.1f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB, peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, graphs {len(eng._graphs)}")
