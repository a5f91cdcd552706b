"""The plain Bagon step with DIFFERENT encoder / decoder ids against tests/golden/step_bagon_tiny.npz, the fixture
tests/golden/make_step_bagon_golden.py produced from HuggingFace's BertModel -> BertLMHeadModel wired as
models/bagon/Bagon.py:40-55 with the loss block of models/bagon/Trainer.py:100-111 (one-hot KL against the decoder's perturbed
ids, argmax(softmax), the reference's own seq_acc).

CPU part : pins oracle/step_oracle.py::bagon_step to the fixture.
GPU part : the HIP path -- Bagon.forward, TrainEngine.forward_backward(dec_ids=...) and the trainer's step() -- against it.
"""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_bagon_tiny.npz")


def _load():
    f = dict(np.load(GOLDEN, allow_pickle=False))
    cfg = {str(k): int(v) for k, v in zip(f["cfg_keys"], f["cfg_vals"])}
    params = {k[2:]: torch.from_numpy(v) for k, v in f.items() if k.startswith("p:")}
    grads = {k[2:]: v for k, v in f.items() if k.startswith("g:")}
    return f, cfg, params, grads


def _load_into(model, params):
    for pre, m in (("encoder.", model.encoder), ("decoder.", model.decoder)):
        sd = {k[len(pre):]: v for k, v in params.items() if k.startswith(pre)}
        missing = m.load_state_dict(sd, strict=False)
        assert not [k for k in missing.missing_keys if "position_ids" not in k], missing


def test_fixture_config_and_ids():
    from models.bagon.Bagon import LOCAL_BERT_CONFIGS
    f, cfg, _, grads = _load()
    assert cfg == LOCAL_BERT_CONFIGS["kvq-bert-fixture"]
    assert (f["ids_enc"] != f["ids_dec"]).sum() > 10                       # the two sides really see different ids
    assert "encoder.embeddings.word_embeddings.weight" in grads and "decoder.bert.embeddings.word_embeddings.weight" in grads


def test_bagon_step_oracle_reproduces_the_fixture():
    from oracle import step_oracle as SO
    torch.set_num_threads(1)
    f, cfg, params, grads = _load()
    m = SO.OracleBagon(cfg).eval()
    _load_into(m, params)
    t = lambda k: torch.from_numpy(f[k])
    out = SO.bagon_step(m, None, t("ids_enc"), t("mask_enc"), t("ids_dec"), t("mask_dec"), cfg["vocab_size"])
    assert np.array_equal(out["recon_ids"].numpy(), f["recon_ids"])
    np.testing.assert_allclose(out["loss_recon"].item(), f["loss_recon"], rtol=1e-6)
    np.testing.assert_allclose(float(out["acc_batch"]), f["acc_batch"], atol=1e-7)
    np.testing.assert_allclose(out["acc_sentence"].numpy(), f["acc_sentence"], atol=1e-7)
    np.testing.assert_allclose(out["logits"].numpy(), f["logits"], rtol=1e-5, atol=1e-5)


def _gpu_model(dtype=torch.float32):
    from models.bagon.Bagon import Bagon
    f, cfg, params, grads = _load()
    model = Bagon("kvq-bert-fixture", "kvq-bert-fixture", True, compute_dtype=dtype)
    _load_into(model, params)
    return model.cuda().eval(), f, grads


def _ids(f):
    return tuple(torch.from_numpy(f[k]).cuda() for k in ("ids_enc", "mask_enc", "ids_dec", "mask_dec"))


@pytest.mark.gpu
def test_bagon_forward_on_gpu_matches_the_fixture():
    """model(enc ids, enc mask, dec ids, dec mask) -> logits (Bagon.py:40-55) on the HIP path, f32."""
    model, f, _ = _gpu_model()
    with torch.no_grad():
        logits = model(*_ids(f))
    np.testing.assert_allclose(logits.float().cpu().numpy(), f["logits"], rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
def test_train_engine_bagon_step_matches_the_fixture():
    """TrainEngine.forward_backward with the decoder's own ids (f32, dropout off): loss, recon ids, both accuracies and the
    gradients of the fixture -- among them the two word-embedding tables, which are summed over different id sets."""
    from kvq.engine import TrainEngine
    model, f, grads = _gpu_model()
    ids_e, mask_e, ids_d, mask_d = _ids(f)
    eng = TrainEngine(model, lr=1e-3)
    eng.sync_from_model()
    out = eng.forward_backward(ids_e, mask_e, training=False, compute_grads=True, dec_ids=ids_d, dec_mask=mask_d)
    assert np.array_equal(out["recon_ids"].cpu().numpy(), f["recon_ids"])
    np.testing.assert_allclose(out["loss_recon"].item(), f["loss_recon"], rtol=2e-5)
    np.testing.assert_allclose(out["acc"].item(), f["acc_batch"], atol=1e-7)
    np.testing.assert_allclose(out["acc_per_sentence"].cpu().numpy(), f["acc_sentence"], atol=1e-7)
    assert out["loss_vq"] is None and out["indices"] is None
    name_of = {id(p): n for n, p in model.named_parameters()}
    checked = 0
    for ename, p in eng.param_of.items():
        n = name_of[id(p)]
        if n in grads:
            got, ref = eng.flat.g(ename).float().cpu().numpy(), grads[n]
            np.testing.assert_allclose(got[: ref.shape[0]], ref, rtol=5e-3, atol=5e-6, err_msg=n)
            checked += 1
    assert checked == len(grads), (checked, sorted(grads))


@pytest.mark.gpu
@pytest.mark.parametrize("use_engine", [True, False])
def test_trainer_step_contract_on_the_fixture(use_engine):
    """models/bagon/Trainer.step: the reference's stats keys and 5-tuple (Trainer.py:124-130), target = the decoder's ids, on the
    engine and on the autograd path; pre-tokenised batches (input_ids_encoder / input_ids_decoder) stand in for the tokenizers."""
    from kvq.engine import TrainEngine
    from models.bagon import Trainer as T
    model, f, _ = _gpu_model()
    ids_e, mask_e, ids_d, mask_d = _ids(f)
    batch = {"input_ids_encoder": ids_e, "attention_mask_encoder": mask_e, "input_ids_decoder": ids_d, "attention_mask_decoder": mask_d,
             "latent_classes_labels": torch.zeros(ids_e.shape[0], 9, dtype=torch.int64)}
    eng = TrainEngine(model, lr=1e-3) if use_engine else None
    with torch.no_grad():
        stats, e_ids, d_ids, recon, labels = T.step(
            device=ids_e.device, model=model, tokenizer_encoder=None, tokenizer_decoder=None,
            tokenizer_encoder_add_special_tokens=False, tokenized_encoder_sentence_max_length=12,
            tokenizer_decoder_add_special_tokens=False, tokenized_decoder_sentence_max_length=12,
            encoder_perturb_pct=0.0, decoder_perturb_pct=0.0, opt=None, lr_sched=None, batch=batch,
            vocab_size_encoder=512, vocab_size_decoder=512, console=None, engine=eng)
    assert set(stats) == {"loss_recon_step", "loss_full_step", "metric_acc_step_per_batch", "metric_acc_step_per_sentence",
                          "padding_tokens_pct_step"}
    assert stats["padding_tokens_pct_step"] == -69 and torch.equal(e_ids, ids_e) and torch.equal(d_ids, ids_d) and labels is batch["latent_classes_labels"]
    np.testing.assert_allclose(stats["loss_recon_step"].item(), f["loss_recon"], rtol=2e-5)
    np.testing.assert_allclose(stats["loss_full_step"].item(), f["loss_recon"], rtol=2e-5)
    np.testing.assert_allclose(stats["metric_acc_step_per_batch"].item(), f["acc_batch"], atol=1e-7)
    np.testing.assert_allclose(stats["metric_acc_step_per_sentence"].cpu().numpy(), f["acc_sentence"], atol=1e-7)
    assert np.array_equal(recon.cpu().numpy(), f["recon_ids"])


# ---- the PRODUCT dtype under the same fixture (round 5): see tests/test_step_golden.py for the stated bf16 bounds ----------------
BF16_LOGIT_ATOL = 0.03


@pytest.mark.gpu
def test_bagon_bf16_forward_and_step_match_the_fixture(monkeypatch):
    """Bagon.forward and TrainEngine.forward_backward(dec_ids=...) in bf16 with every torch GEMM entry point patched to raise:
    logits within 0.03, recon ids equal wherever the fixture's top-2 logit margin exceeds that, loss rtol 2e-2, both accuracies
    within the tokens that may flip, gradients by cosine."""
    from _gemm_guard import forbid_vendor_gemms
    from kvq.engine import TrainEngine
    model, f, grads = _gpu_model(torch.bfloat16)
    ids_e, mask_e, ids_d, mask_d = _ids(f)
    eng = TrainEngine(model, lr=1e-3)
    eng.sync_from_model()
    forbid_vendor_gemms(monkeypatch)
    with torch.no_grad():
        logits = model(ids_e, mask_e, ids_d, mask_d)
    assert logits.dtype == torch.bfloat16
    err = np.abs(logits.float().cpu().numpy() - f["logits"]).max()
    assert err <= BF16_LOGIT_ATOL, err
    out = eng.forward_backward(ids_e, mask_e, training=False, compute_grads=True, dec_ids=ids_d, dec_mask=mask_d)
    s = np.sort(f["logits"].reshape(-1, f["logits"].shape[-1]), axis=-1)
    margin = s[:, -1] - s[:, -2]
    differ = out["recon_ids"].cpu().numpy().reshape(-1) != f["recon_ids"].reshape(-1)
    assert not (differ & (margin > BF16_LOGIT_ATOL)).any() and differ.mean() <= 0.25, (differ.sum(), margin[differ])
    np.testing.assert_allclose(out["loss_recon"].item(), f["loss_recon"], rtol=2e-2)
    n_tok = differ.size
    assert abs(out["acc"].item() - float(f["acc_batch"])) <= differ.sum() / n_tok + 1e-6
    name_of = {id(p): n for n, p in model.named_parameters()}
    cos = []
    for ename, p in eng.param_of.items():
        n = name_of[id(p)]
        if n in grads and not n.endswith("key.bias") and np.linalg.norm(grads[n]) > 0:
            got = eng.flat.g(ename).float().cpu().numpy()
            got, ref = got[: grads[n].shape[0]].reshape(-1), grads[n].reshape(-1)
            cos.append((float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30)), n))
    print("bf16 Bagon engine vs the f32 fixture: worst gradient cosine", min(cos), "mean", np.mean([c for c, _ in cos]))
    assert min(cos)[0] > 0.98 and np.mean([c for c, _ in cos]) > 0.995, sorted(cos)[:5]
