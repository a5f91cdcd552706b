"""GumbelQuantizer on libkvq.so against the CPU oracle (oracle/gumbel_oracle.py, itself pinned to the reference module's outputs)
and directly against those golden outputs, with the Gumbel noise passed in explicitly; plus the statistics of the library's own
Philox noise and the model-level dispatch (Shelgon.py:60-65)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "gumbel_*.npz")))


def _module(c, dtype=torch.float32):
    from models.shelgon3.GumbelQuantizer import GumbelQuantizer
    K, H = c["W"].shape
    m = GumbelQuantizer(enc_out_size=H, n_embed=K, embedding_dim=c["E"].shape[1], temperature=float(c["tau"]),
                        kl_div_scale=float(c["kld_scale"]), straight_through=bool(c["straight_through"])).cuda()
    with torch.no_grad():
        m.proj.weight.copy_(torch.from_numpy(c["W"])[:, :, None]); m.proj.bias.copy_(torch.from_numpy(c["b"]))
        m.embed.weight.copy_(torch.from_numpy(c["E"]))
    return m


@pytest.mark.parametrize("name", CASES)
def test_module_matches_reference_outputs_and_gradients(name):
    c = dict(np.load(os.path.join(GOLD, name + ".npz")))
    m = _module(c)
    assert set(m.state_dict()) == {"proj.weight", "proj.bias", "embed.weight"} and m.proj.weight.shape[2] == 1
    m.train(bool(c["is_training"]))
    z = torch.from_numpy(c["z"]).cuda().requires_grad_(True)
    B, S, K = c["noise"].shape
    z_q, diff, ind = m(z, bool(c["is_training"]), noise=torch.from_numpy(c["noise"]).cuda().reshape(B * S, K))
    assert ind.dtype == torch.int64 and ind.shape == (B, S) and z_q.shape == c["z_q"].shape
    assert np.array_equal(ind.cpu().numpy(), c["ind"])                         # same codes as the reference
    np.testing.assert_allclose(z_q.detach().cpu().numpy(), c["z_q"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(float(diff), float(c["diff"]), rtol=1e-4, atol=1e-9)
    ((z_q * torch.from_numpy(c["G"]).cuda()).sum() + diff * float(c["c"])).backward()
    got = dict(grad_z=z.grad, grad_W=m.proj.weight.grad[:, :, 0], grad_b=m.proj.bias.grad, grad_E=m.embed.weight.grad)
    for k, v in got.items():
        np.testing.assert_allclose(v.cpu().numpy(), c[k], rtol=2e-3, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("N,K,hard", [(300, 512, True), (257, 70, False), (64, 1024, True), (5, 3, True)])
def test_row_kernel_equals_oracle(N, K, hard):
    from kvq import functional as KF
    from oracle import gumbel_oracle as GO
    rng = np.random.default_rng(N + K)
    logits = rng.normal(size=(N, K)).astype(np.float32) * 2
    noise = -np.log(rng.exponential(size=(N, K))).astype(np.float32)
    lg = torch.from_numpy(logits).cuda().requires_grad_(True)
    y, diff, ind = KF.gumbel_quantize(lg, 0.8, hard, 0.37, noise=torch.from_numpy(noise).cuda())
    # oracle with W = identity: logits in, E = identity: y out
    o = GO.forward(logits[None], np.eye(K, dtype=np.float32), np.zeros(K, np.float32), np.eye(K, dtype=np.float32), noise[None], 0.8, hard, 0.37)
    assert np.array_equal(ind.cpu().numpy(), o["ind"][0])
    np.testing.assert_allclose(y.detach().cpu().numpy(), o["y"][0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(diff), float(o["diff"]), rtol=1e-4, atol=1e-8)
    if hard:
        assert torch.all((y.detach() > 0.5).sum(1) == 1)                     # exactly one hot entry per row
    G = torch.from_numpy(rng.normal(size=(N, K)).astype(np.float32)).cuda()
    ((y * G).sum() + 1.7 * diff).backward()
    g = GO.forward_backward_torch(logits[None], np.eye(K, dtype=np.float32), np.zeros(K, np.float32), np.eye(K, dtype=np.float32), noise[None],
                                  0.8, hard, 0.37, G.cpu().numpy()[None], 1.7)
    # d/dlogits == d/db of the oracle's affine map summed per column?  no: take grad_z with W = I -> exactly d/dlogits
    np.testing.assert_allclose(lg.grad.cpu().numpy(), g["grad_z"][0], rtol=2e-3, atol=2e-5)


def test_philox_noise_is_gumbel_and_reproducible():
    from kvq import functional as KF
    N, K = 4096, 64
    zeros = torch.zeros(N, K, device="cuda")
    _, _, ind = KF.gumbel_quantize(zeros, 1.0, True, 0.0, seed=5, site=1)
    _, _, ind2 = KF.gumbel_quantize(zeros, 1.0, True, 0.0, seed=5, site=1)
    _, _, ind3 = KF.gumbel_quantize(zeros, 1.0, True, 0.0, seed=5, site=2)
    assert torch.equal(ind, ind2) and not torch.equal(ind, ind3)
    counts = torch.bincount(ind, minlength=K).float()                        # argmax of iid Gumbels over equal logits is uniform
    assert (counts - N / K).abs().max() < 6 * (N / K) ** 0.5
    # with logits l the argmax follows softmax(l) (the Gumbel-max trick)
    l = torch.log(torch.tensor([0.5, 0.25, 0.125, 0.125], device="cuda")).repeat(20000, 1)
    _, _, i4 = KF.gumbel_quantize(l, 1.0, True, 0.0, seed=9, site=3)
    freq = torch.bincount(i4, minlength=4).float() / 20000
    torch.testing.assert_close(freq.cpu(), torch.tensor([0.5, 0.25, 0.125, 0.125]), rtol=0, atol=0.015)
    y_soft, _, _ = KF.gumbel_quantize(l[:100].bfloat16(), 0.5, False, 0.0, seed=1, site=1)
    assert y_soft.dtype == torch.bfloat16 and torch.allclose(y_soft.float().sum(1), torch.ones(100, device="cuda"), atol=2e-2)


def test_shelgon_dispatches_on_the_class_name_and_trains():
    from models.shelgon3.GumbelQuantizer import GumbelQuantizer
    from models.shelgon3.Shelgon import Shelgon
    torch.manual_seed(0)
    gq = GumbelQuantizer(enc_out_size=128, n_embed=16, embedding_dim=128, temperature=1.0, kl_div_scale=5e-4, straight_through=True)
    model = Shelgon("kvq-bert-tiny", gq, "kvq-bert-tiny", None, compute_dtype=torch.bfloat16).cuda().train()
    ids = torch.randint(1000, 2000, (8, 12), device="cuda"); mask = torch.ones_like(ids)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(25):
        vq_loss, perp, idx, l_rec, acc, recon = model.forward_loss(ids, mask)
        opt.zero_grad(); (l_rec + vq_loss).backward(); opt.step()
        losses.append(float(l_rec))
    assert idx.shape == (8, 12) and 1 <= float(perp) <= 16 and gq.proj.weight.grad is not None and gq.embed.weight.grad is not None
    assert losses[-1] < 0.8 * losses[0], losses[::6]
    model.eval()
    with torch.no_grad():
        vq_loss, perp, idx, logits = model(ids, mask)
    assert logits.shape[:2] == (8, 12)
    from kvq.engine import TrainEngine, engine_of
    assert TrainEngine.supports(model, 12)                    # round 2: the engine schedules this quantiser too
    assert engine_of(model, create=False) is not None        # ... and the no-grad forward above already ran on it
