"""The composed step (encoder -> VectorQuantizer -> decoder -> one-hot KL) against tests/golden/step_tiny.npz, the fixture
tests/golden/make_step_golden.py produced by running the reference's own VectorQuantizer between HuggingFace's BertModel and
BertLMHeadModel, wired as models/shelgon3/Shelgon.py:50-73 and models/shelgon3/Trainer.py:94-105.

CPU part : pins oracle/step_oracle.py (the CPU restatement bench.py times) to the fixture.
GPU part : the HIP path -- Shelgon.forward and the TrainEngine -- against the same fixture.
"""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_tiny.npz")


def _load():
    f = dict(np.load(GOLDEN, allow_pickle=False))
    cfg = {str(k): int(v) for k, v in zip(f["cfg_keys"], f["cfg_vals"])}
    params = {k[2:]: torch.from_numpy(v) for k, v in f.items() if k.startswith("p:")}
    grads = {k[2:]: v for k, v in f.items() if k.startswith("g:")}
    return f, cfg, params, grads


def _load_into(model, params, codebook):
    enc_sd = {k[len("encoder."):]: v for k, v in params.items() if k.startswith("encoder.")}
    dec_sd = {k[len("decoder."):]: v for k, v in params.items() if k.startswith("decoder.")}
    missing = model.encoder.load_state_dict(enc_sd, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing
    missing = model.decoder.load_state_dict(dec_sd, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing
    model.vector_quantizer.embedding.weight.data.copy_(torch.from_numpy(codebook))


def test_fixture_config_matches_the_local_model_table():
    from models.bagon.Bagon import LOCAL_BERT_CONFIGS
    _, cfg, _, _ = _load()
    assert cfg == LOCAL_BERT_CONFIGS["kvq-bert-fixture"]


def test_step_oracle_reproduces_the_reference_fixture():
    """oracle/step_oracle.py (CPU, f32) == the fixture: same indices and recon ids, logits / losses to f32 rounding."""
    from oracle import step_oracle as SO
    torch.set_num_threads(1)
    f, cfg, params, grads = _load()
    m = SO.OracleShelgon(cfg, n_e=int(f["K"]), e_dim=cfg["hidden_size"], beta=float(f["beta"]),
                         codebook_init=torch.from_numpy(f["codebook"])).eval()
    _load_into(m, params, f["codebook"])
    ids, mask = torch.from_numpy(f["ids"]), torch.from_numpy(f["mask"])
    out = SO.step(m, None, ids, mask, cfg["vocab_size"])
    assert np.array_equal(out["idx"].reshape(-1).numpy(), f["idx"])
    assert np.array_equal(out["recon_ids"].numpy(), f["recon_ids"])
    np.testing.assert_allclose(out["loss_recon"].item(), f["loss_recon"], rtol=1e-6)
    np.testing.assert_allclose(out["loss_vq"].item(), f["loss_vq"], rtol=1e-6)
    np.testing.assert_allclose(out["perplexity"].item(), f["perplexity"], rtol=1e-6)
    np.testing.assert_allclose(float(out["acc"]), f["acc"], atol=1e-7)
    _, _, _, logits = m(ids, mask)
    np.testing.assert_allclose(logits.detach().numpy(), f["logits"], rtol=1e-5, atol=1e-5)


def _gpu_model(dtype=torch.float32):
    from models.shelgon3.Shelgon import Shelgon
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    f, cfg, params, grads = _load()
    vq = VectorQuantizer(int(f["K"]), cfg["hidden_size"], float(f["beta"]))
    vq.materialize_min_encodings = False
    model = Shelgon("kvq-bert-fixture", vq, "kvq-bert-fixture", None, compute_dtype=dtype)
    _load_into(model, params, f["codebook"])
    return model.cuda().eval(), f, grads


@pytest.mark.gpu
def test_shelgon_forward_on_gpu_matches_the_reference_fixture():
    """model(ids, mask, device, False) -> (vq_loss, perplexity, indices, logits) (Shelgon.py:50-73) on the HIP path, f32."""
    model, f, _ = _gpu_model()
    ids, mask = torch.from_numpy(f["ids"]).cuda(), torch.from_numpy(f["mask"]).cuda()
    with torch.no_grad():
        vq_loss, perp, idx, logits = model(ids, mask, ids.device, False)
    assert idx.shape == (ids.shape[0], ids.shape[1], 1) and idx.dtype == torch.int64
    assert np.array_equal(idx.reshape(-1).cpu().numpy(), f["idx"])            # min top-2 gap of the fixture is 0.11: no near ties
    np.testing.assert_allclose(logits.float().cpu().numpy(), f["logits"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(vq_loss.item(), f["loss_vq"], rtol=2e-5)
    np.testing.assert_allclose(perp.item(), f["perplexity"], rtol=2e-5)


@pytest.mark.gpu
def test_train_engine_on_gpu_matches_the_reference_fixture():
    """The TrainEngine step (f32, dropout off): losses, indices, recon ids, accuracy and gradients against the fixture."""
    from kvq.engine import TrainEngine
    model, f, grads = _gpu_model()
    ids, mask = torch.from_numpy(f["ids"]).cuda(), torch.from_numpy(f["mask"]).cuda()
    eng = TrainEngine(model, lr=1e-3)
    eng.sync_from_model()
    out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
    assert np.array_equal(out["indices"].reshape(-1).cpu().numpy(), f["idx"])
    assert np.array_equal(out["recon_ids"].cpu().numpy(), f["recon_ids"])
    np.testing.assert_allclose(out["loss_recon"].item(), f["loss_recon"], rtol=2e-5)
    np.testing.assert_allclose(out["loss_vq"].item(), f["loss_vq"], rtol=2e-5)
    np.testing.assert_allclose(out["perplexity"].item(), f["perplexity"], rtol=2e-5)
    np.testing.assert_allclose(out["acc"].item(), f["acc"], atol=1e-7)
    name_of = {id(p): n for n, p in model.named_parameters()}
    checked = 0
    for ename, p in eng.param_of.items():
        n = name_of[id(p)]
        if n in grads:
            got = eng.flat.g(ename).float().cpu().numpy()
            ref = grads[n]
            np.testing.assert_allclose(got[: ref.shape[0]] if got.ndim == 2 else got[: ref.shape[0]], ref, rtol=5e-3, atol=5e-6,
                                       err_msg=n)
            checked += 1
    assert checked == len(grads), (checked, sorted(grads))
    np.testing.assert_allclose(eng.gE.cpu().numpy(), f["grad_codebook"], rtol=2e-3, atol=1e-7)
