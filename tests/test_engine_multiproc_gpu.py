"""Two data-parallel ranks of the TrainEngine (both on cuda:0, gloo process group so one GPU suffices) must end up with
the weights of a single process that trained on the concatenated batch: checks the tail-chunk gradient exchange,
its stream ordering and the codebook-gradient all-reduce.  (On a multi-GPU node the same code runs over RCCL.)"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS = 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build():
    sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
    from models.shelgon3.Shelgon import Shelgon
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    torch.manual_seed(0)
    vq = VectorQuantizer(32, 128, 0.25, vq_codebook_init_values=torch.randn(32, 128))
    vq.materialize_min_encodings = False
    return Shelgon("kvq-bert-tiny", vq, "kvq-bert-tiny", None, compute_dtype=torch.float32).cuda().eval()


def _data():
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1000, 2000, (8, 16), generator=g)
    lens = torch.randint(3, 17, (8,), generator=g)
    ids = ids * (torch.arange(16)[None] < lens[:, None])
    return ids.cuda(), (ids != 0).long().cuda()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
    import torch.distributed as dist
    from kvq import ddp
    from kvq.engine import TrainEngine
    torch.cuda.set_device(0)
    ddp.init_distributed("gloo")
    ids, mask = _data()
    half = slice(rank * 4, rank * 4 + 4)
    saved = {}
    for tag, use_graph in (("eager", False), ("graph", True)):
        model = _build()
        ddp.broadcast_parameters(model)
        eng = TrainEngine(model, lr=1e-3, bucket_mib=0)           # bucket_mib=0 -> clamped inside: many small all-reduce chunks
        eng.use_graph = use_graph
        assert eng.world == 2
        for step in range(STEPS):               # graph run: steps 1-2 launch eagerly, from step 3 on the hipGraph chain is replayed
            eng.train_step(ids[half], mask[half])
            if step == 1 and not use_graph:
                saved["eager2"] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        if use_graph:
            assert len(eng._graphs) == 1 and len(next(iter(eng._graphs.values())).inter) >= 3      # quantiser + all-reduces + wait
        torch.cuda.synchronize()
        saved[tag] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if rank == 0:
        torch.save(saved, out)
    dist.barrier()
    dist.destroy_process_group()


def _differences(got, ref, steps):
    bad = []
    for k, v in ref.items():
        if "pooler" in k or "key.bias" in k or "position_ids" in k:
            continue
        # Adam divides by sqrt(v): an entry whose gradient is pure rounding noise (different GEMM shapes per rank, atomic
        # scatter-add order) moves by +-lr whatever the noise's size, so a handful of such entries may differ by up to
        # steps * lr; everything else must agree tightly.
        close = torch.isclose(got[k], v.cpu(), rtol=2e-3, atol=3e-5)
        worst = (got[k] - v.cpu()).abs().max().item()
        if (~close).float().mean().item() > 2e-3 or worst > 1.25e-3 * steps:
            bad.append((k, worst, (~close).float().mean().item()))
    return bad


def test_two_ranks_equal_single_process_big_batch(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # (1) the hipGraph chain with eager all-reduce interludes == the same two-rank run launched kernel by kernel
    bad = _differences(got["graph"], got["eager"], STEPS)
    assert not bad, ("graph vs eager", bad[:5])
    # (2) two ranks == one process on the concatenated batch (two steps: further on, a flipped code assignment makes the
    #     two trajectories diverge for reasons that have nothing to do with the gradient exchange)
    sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
    from kvq.engine import TrainEngine
    model = _build()
    eng = TrainEngine(model, lr=1e-3)
    eng.use_graph = False
    ids, mask = _data()
    # the two half-batches have equal token counts, so the mean of the rank means is the global mean
    for _ in range(2):
        eng.train_step(ids, mask)
    bad = _differences(got["eager2"], model.state_dict(), 2)
    assert not bad, ("two ranks vs one process", bad[:5])


def _worker_rccl_one_rank(rank, port, out):
    """ONE rank, backend nccl (= RCCL): every all-reduce, side-stream hand-over and graph interlude of the data-parallel engine
    executes on the GPU box's real collective library; with one rank the averages leave the gradients as they are."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      KVQ_DP_SINGLE_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
    import torch.distributed as dist
    from kvq import ddp
    from kvq.engine import TrainEngine
    torch.cuda.set_device(0)
    ddp.init_distributed("nccl")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    ids, mask = _data()
    saved = {}
    for tag, use_graph in (("eager", False), ("graph", True)):
        model = _build()
        eng = TrainEngine(model, lr=1e-3, bucket_mib=0)
        eng.use_graph = use_graph
        assert eng._dp and eng.world == 1 and eng._avg_in_comm
        eng.reset_comm_timing()
        for _ in range(STEPS):
            eng.train_step(ids, mask)
        if use_graph:
            assert len(eng._graphs) == 1 and len(next(iter(eng._graphs.values())).inter) >= 3
        torch.cuda.synchronize()
        saved[tag] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        saved[tag + "_exposed_ms"] = eng.exposed_comm_ms()
    torch.save(saved, out)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_over_rccl_equals_the_plain_engine(tmp_path):
    out = str(tmp_path / "rccl1.pt")
    mp.spawn(_worker_rccl_one_rank, args=(_free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))
    from kvq.engine import TrainEngine
    model = _build()
    eng = TrainEngine(model, lr=1e-3)
    eng.use_graph = False
    assert not eng._dp
    ids, mask = _data()
    for _ in range(STEPS):
        eng.train_step(ids, mask)
    for tag in ("eager", "graph"):
        bad = _differences(got[tag], model.state_dict(), STEPS)
        assert not bad, (tag, bad[:5])
        assert got[tag + "_exposed_ms"] >= 0.0
