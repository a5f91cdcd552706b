"""Round 5: dropout + residual in the epilogue of the dense layer in front of every residual LayerNorm (BertSelfOutput / BertOutput,
modeling_bert.py:282-296, 339-352; csrc/kvq_gemm2.hip EPI_DROPRES) against the two-kernel form of rounds 1 - 4 (plain GEMM, then
kvq_dropout_residual_ln_fwd reading two tensors): the TRAINING step -- dropout on, the masks of step t drawn from the device step
counter -- must not change by a bit.  (The engine takes the fused form by default only for steps WITHOUT dropout -- the validation and
test stages, model.forward: with dropout the Philox rounds cost a GEMM epilogue more than the LayerNorm kernel saves,
profiles/r05_gemm_ceiling.md; KVQ_FUSE_DROPRES=1 forces it, as here.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ids(B, S, seed, hi=2000):
    from dsentences.synthetic import random_token_batch
    ids, mask = random_token_batch(B, S, torch.Generator().manual_seed(seed), vocab_hi=hi, min_len=3, max_len=S)
    return ids.cuda(), mask.cuda()


def _model(kind, name):
    from models.bagon.Bagon import Bagon, LOCAL_BERT_CONFIGS
    from models.shelgon3.Shelgon import Shelgon
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    H = LOCAL_BERT_CONFIGS[name].get("hidden_size", 768)
    torch.manual_seed(0)
    if kind == "bagon":
        return Bagon(name, name, True, compute_dtype=torch.bfloat16).cuda().train()
    vq = VectorQuantizer(32, H, 0.25, vq_codebook_init_values=torch.randn(32, H))
    vq.materialize_min_encodings = False
    return Shelgon(name, vq, name, None, compute_dtype=torch.bfloat16).cuda().train()


@pytest.mark.parametrize("kind,name,B,S", [("vq", "kvq-bert-tiny", 16, 12), ("bagon", "kvq-bert-tiny", 16, 12), ("vq", "kvq-bert-base-2l", 64, 32)])
def test_training_steps_with_the_fused_epilogue_equal_the_two_kernel_form(kind, name, B, S, monkeypatch):
    from kvq import nnops
    from kvq.engine import TrainEngine
    ids, mask = _ids(B, S, seed=4)
    runs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("KVQ_FUSE_DROPRES", fused)
        model = _model(kind, name)
        eng = TrainEngine(model, lr=1e-3)
        assert eng._fuse_dropres == fused
        calls = {"n": 0}
        real = nnops.gemm_dropres
        monkeypatch.setattr(nnops, "gemm_dropres", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), real(*a, **k))[1])
        kw = {}
        if kind == "bagon":
            dec = torch.where(torch.rand(ids.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) < 0.2,
                              torch.full_like(ids, 1500), ids) * mask
            kw = dict(dec_ids=dec, dec_mask=mask)
        losses = []
        for _ in range(5):                               # eager warm-up, capture, replays: all forms of the step
            out = eng.train_step(ids, mask, **kw)
            losses.append(float(out["loss_recon"]))
        torch.cuda.synchronize()
        monkeypatch.setattr(nnops, "gemm_dropres", real)
        assert (calls["n"] > 0) == (fused == "1"), calls
        runs[fused] = (losses, eng.flat.master.clone(), out["recon_ids"].clone())
    (l1, w1, r1), (l0, w0, r0) = runs["1"], runs["0"]
    assert np.isfinite(l1).all() and l1[-1] < l1[0], l1
    assert l1 == l0, (l1, l0)
    assert torch.equal(w1.view(torch.int32), w0.view(torch.int32)) and torch.equal(r1, r0)


def test_default_policy_fuses_the_steps_without_dropout_only(monkeypatch):
    from kvq import nnops
    from kvq.engine import TrainEngine
    monkeypatch.delenv("KVQ_FUSE_DROPRES", raising=False)
    ids, mask = _ids(16, 12, seed=9)
    eng = TrainEngine(_model("vq", "kvq-bert-tiny"), lr=1e-3)
    calls = {"n": 0}
    real = nnops.gemm_dropres
    monkeypatch.setattr(nnops, "gemm_dropres", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), real(*a, **k))[1])
    eng.forward_backward(ids, mask, training=True, compute_grads=True)
    assert calls["n"] == 0                                   # dropout on: the two-kernel form
    out = eng.forward_backward(ids, mask, training=False, compute_grads=False)
    assert calls["n"] == 10 and np.isfinite(float(out["loss_recon"]))     # 2 + 2 layers: ten residual LayerNorms
