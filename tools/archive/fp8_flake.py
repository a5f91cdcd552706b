#!/usr/bin/env python3
"""Find what makes the fp8 "all" step differ from run to run (bench.py --fp8-all ended at different losses in different processes):
train two identically seeded engines for a few steps in ONE process and print the per-step losses of both; run the script a few
times.  usage: fp8_flake.py [all|wide|off] [factors] [steps] [graph 0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
import torch  # noqa: E402
from dsentences.synthetic import random_token_batch  # noqa: E402
from kvq.engine import TrainEngine  # noqa: E402
from models.shelgon3.MultiVectorQuantizer import MultiVectorQuantizer  # noqa: E402
from models.shelgon3.Shelgon import Shelgon  # noqa: E402
from models.shelgon3.VectorQuantizer import VectorQuantizer  # noqa: E402

scope = sys.argv[1] if len(sys.argv) > 1 else "all"
factors = int(sys.argv[2]) if len(sys.argv) > 2 else 9
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
graph = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
gen = torch.Generator().manual_seed(69)
NP = int(os.environ.get('POOL', '8'))
pool = [tuple(t.cuda() for t in random_token_batch(256, 32, gen)) for _ in range(NP)]
pool_ref = [(a.clone(), b.clone()) for a, b in pool]


def run():
    torch.manual_seed(0)
    vq = MultiVectorQuantizer(n_factors=factors, n_e=512, e_dim=768, beta=0.25) if factors > 1 else VectorQuantizer(512, 768, 0.25)
    model = Shelgon("bert-base-uncased", vq, "bert-base-uncased", None, compute_dtype=torch.bfloat16).cuda()
    model.set_mode("full")
    model.train()
    eng = TrainEngine(model, lr=1e-4, milestones=[10000, 20000], fp8_forward={"all": "all", "wide": True, "off": False}[scope])
    eng.use_graph = graph
    out = []
    packs = [eng.pack_batch(*b) for b in pool] if os.environ.get("PREP") == "1" else [None] * NP
    if os.environ.get("PROF") == "1":
        from kvq._ffi import lib
        lib().kvq_prof_enable(steps + 4)
    for i in range(steps):
        pk = packs[i % NP]
        if pk is not None and os.environ.get("CLONEPACK") == "1":
            pk = pk.clone()
        o = eng.train_step(*pool[i % NP], prepared=pk)
        out.append((float(o["loss_recon"]), float(o["loss_vq"])))
    if packs[0] is not None:
        torch.cuda.synchronize()
        for i, pk in enumerate(packs):
            fresh = eng.pack_batch(*pool[i])
            if not torch.equal(pk, fresh):
                d = (pk != fresh).nonzero()
                print(f"   pack {i} CHANGED under the run: {d.shape[0]} entries, rows {sorted(set(d[:, 0].tolist()))}, first {d[:3].tolist()}", flush=True)
        for i, (ids, mask) in enumerate(pool):
            if not torch.equal(ids, pool_ref[i][0]) or not torch.equal(mask, pool_ref[i][1]):
                print(f"   batch {i} ids / mask CHANGED under the run", flush=True)
    del eng, model
    torch.cuda.empty_cache()
    return out


a, b = run(), run()
first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
print(f"scope={scope} factors={factors} graph={graph}: first differing step between two in-process runs: {first}; "
      f"final {a[-1][0] + a[-1][1]:.6f} vs {b[-1][0] + b[-1][1]:.6f}; run A losses {[round(x + y, 4) for x, y in a]}", flush=True)
