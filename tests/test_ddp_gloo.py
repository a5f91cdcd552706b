"""Multi-process data parallelism on CPU (gloo, world_size 2): kvq.ddp.GradSync must reproduce the single-process
gradient of the concatenated batch, with bucketing and the zero_grad/backward/finish protocol of the trainers."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                               torch.nn.Linear(64, 8))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from kvq import ddp
    r, _, w = ddp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    if rank == 1:                                    # diverge on purpose: broadcast must repair it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    ddp.broadcast_parameters(model)
    model[2].bias.requires_grad_(False)              # a frozen parameter must not break the buckets
    sync = ddp.GradSync(model.parameters(), bucket_mib=0)      # bucket_mib=0 -> one bucket per parameter (max bucket count)
    assert len(sync.buckets) == 5
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 16, generator=g); y = torch.randn(8, 8, generator=g)
    per = 8 // world
    xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1)
    for _ in range(2):
        sync.zero_grad()
        torch.nn.functional.mse_loss(model(xs), ys).backward()
        sync.finish()
        opt.step()
    t = torch.tensor([1.0 + rank])
    ddp.all_reduce_mean_(t)
    assert t.item() == (world + 1) / 2
    # the helpers of the entry points' test stage (models/*/main.py): rank 0's decision and run id everywhere, records to rank 0
    assert ddp.agree(rank == 0) is True and ddp.agree(rank != 0) is False
    assert ddp.same_everywhere(f"run-of-rank-{rank}") == "run-of-rank-0"
    rows = ddp.gather_lists([{"rank": rank, "i": i} for i in range(rank + 1)])
    assert rows == ([{"rank": r, "i": i} for r in range(world) for i in range(r + 1)] if rank == 0 else [])
    if rank == 0:
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradsync_matches_single_process_big_batch(tmp_path, world):
    """world 8 = the rank count of BASELINE.json configs[2] (8 x MI355X), exercised in logic on the CPU: 8 shards of one sample."""
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    model = _model()
    model[2].bias.requires_grad_(False)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 16, generator=g); y = torch.randn(8, 8, generator=g)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1)
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.mse_loss(model(x), y).backward()      # mean over the global batch == mean of the two rank means
        opt.step()
    for k, v in model.state_dict().items():
        torch.testing.assert_close(got[k], v, rtol=1e-5, atol=1e-6)


def test_gradsync_single_process_is_a_noop_wrapper():
    from kvq import ddp
    model = _model()
    sync = ddp.GradSync(model.parameters(), bucket_mib=1)
    assert sync.world == 1 and len(sync.buckets) == 1 and sync.grad_bytes() == sum(p.numel() * 4 for p in model.parameters())
    sync.zero_grad()
    model(torch.randn(2, 16)).sum().backward()
    sync.finish()
    for p in model.parameters():
        assert p.grad is not None and p.grad.data_ptr() >= sync.buckets[0].flat.data_ptr()


def _stats_worker(rank, world, port, out_dir):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kindergarten-vq-vae_amd"))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from models.shelgon3.Trainer import end_of_epoch_stats_update, init_stats_best, init_stats_run
    run = init_stats_run()
    n = 10 + 5 * rank                                         # ranks saw different numbers of sentences
    run["loss_recon_run"] = torch.tensor(2.0 * (rank + 1)) * n
    run["loss_vq_run"] = 0.5 * n
    run["metric_acc_run"] = torch.tensor(50.0 + 10 * rank) * n
    run["metric_perp_run"] = 7.0 * n
    run["loss_full_run"] = run["loss_recon_run"] + run["loss_vq_run"]
    stats, best = end_of_epoch_stats_update(run, init_stats_best(), n, 3)
    torch.save({k: float(v) for k, v in stats.items()}, os.path.join(out_dir, f"stats{rank}.pt"))
    dist.destroy_process_group()


def test_epoch_statistics_are_global_means_on_every_rank(tmp_path):
    """Trainer.end_of_epoch_stats_update under data parallelism: one all-reduce of the running sums, every rank reports the mean
    over the whole split (what a single process would print and base its best-checkpoint decision on)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_stats_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "stats0.pt"), torch.load(tmp_path / "stats1.pt")
    assert a == b
    want_recon = (2.0 * 10 + 4.0 * 15) / 25
    assert abs(a["loss_recon_run"] - want_recon) < 1e-9 and abs(a["loss_vq_run"] - 0.5) < 1e-9
    assert abs(a["metric_acc_run"] - (50.0 * 10 + 60.0 * 15) / 25) < 1e-9 and abs(a["metric_perp_run"] - 7.0) < 1e-9
