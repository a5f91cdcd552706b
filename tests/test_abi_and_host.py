"""CPU-side checks: the C-ABI library exports exactly what include/kvq.h declares, the ctypes table matches it,
and the host logic mirrored from the reference (config, tokenizer, dataset, stats, freeze modes, BERT plan) behaves."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kindergarten-vq-vae_amd")


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "kvq.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kvq_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    so = os.path.join(PKG, "lib", "libkvq.so")
    if not os.path.exists(so):
        g.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = set(re.findall(r" T (kvq_[a-z0-9_]+)", out))
    declared = set(_header_functions())
    assert declared, "no declarations parsed from include/kvq.h"
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert exported <= declared, f"exported but not declared in include/kvq.h: {sorted(exported - declared)}"


def test_ctypes_table_matches_header_and_library_loads():
    from kvq import _ffi
    assert sorted(_ffi.SIGNATURES) == _header_functions()
    lib = _ffi.lib()                                   # loads on a GPU-less host too (no compute calls here)
    assert lib.kvq_version() == 100
    assert lib.kvq_vq_workspace_bytes(8192, 512, 768, 1) > 0
    assert lib.kvq_vq_uses_mfma(8192, 512, 768) == 1 and lib.kvq_vq_uses_mfma(20, 16, 8) == 0
    # argument validation happens before any HIP call: error code + message, no crash
    rc = lib.kvq_vq_forward(None, None, 0, 0, 0, 0, 0, 0.25, None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.kvq_last_error()


def test_ops_refuse_cpu_tensors_loudly():
    import kvq
    from kvq._ffi import KvqError
    with pytest.raises(KvqError):
        kvq.vector_quantize(torch.randn(4, 8), torch.randn(3, 8), 0.25)
    with pytest.raises(KvqError):
        kvq.fused_cross_entropy(torch.randn(4, 10), torch.zeros(4, dtype=torch.long))


def test_product_package_never_imports_the_oracle():
    bad = []
    for dp, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "libkvq_oracle" in src:
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product code references the oracle: {bad}"


def test_config_surface():
    sys.path.insert(0, os.path.join(PKG, "models", "shelgon3"))
    try:
        import importlib
        os.environ["KVQ_VQ_N_E"] = "64"
        cfg = importlib.import_module("config")
        importlib.reload(cfg)
        c = cfg.get_config()
        assert c["vq_n_e"] == 64 and c["vq_e_dim"] == 768 and c["encoder_model_name"] == "bert-base-uncased"
        for key in ("sentences_path", "batch_size", "model_mode", "lr", "milestones", "loss_vq_weight", "export_checkpoint",
                    "tokenized_sentence_max_length", "tokenizer_add_special_tokens", "num_workers", "pin_memory"):
            assert key in c
        import json
        json.dumps(c)
    finally:
        os.environ.pop("KVQ_VQ_N_E", None)
        sys.path.pop(0)
        sys.modules.pop("config", None)


def test_synthetic_corpus_dataset_and_tokenizer(tmp_path):
    from dsentences.dataset import dSentencesDataset
    from dsentences.synthetic import FACTOR_SIZES, random_token_batch, write_corpus
    from kvq.tokenizer import load_tokenizer
    paths = write_corpus(str(tmp_path), 200, seed=69)
    ds = dSentencesDataset(*paths)
    assert len(ds) == 200
    item = ds[3]
    assert isinstance(item["sentence"], str) and item["latent_classes_labels"].shape == (9,)
    assert item["latent_classes_one_hot"].shape == (sum(FACTOR_SIZES),) and item["latent_classes_one_hot"].sum() == 9
    assert len(dSentencesDataset(paths[0])[0]) == 1
    tok = load_tokenizer("bert-base-uncased")
    enc = tok([ds[i]["sentence"] for i in range(5)], return_tensors="pt", padding="max_length", max_length=12, add_special_tokens=False)
    assert enc.input_ids.shape == (5, 12) and torch.equal(enc.attention_mask, (enc.input_ids != 0).long())
    assert (enc.input_ids[enc.attention_mask.bool()] >= 1000).all()          # every corpus word is in the vocabulary
    assert tok.batch_decode(enc.input_ids)[0].startswith(ds[0]["sentence"])
    enc2 = tok(["he accepted the payment"], add_special_tokens=True, padding=True)
    assert enc2.input_ids[0, 0] == 101 and enc2.input_ids[0, -1] == 102
    ids, mask = random_token_batch(4, 32, torch.Generator().manual_seed(69))
    assert ids.shape == (4, 32) and ((ids == 0) == (mask == 0)).all() and mask.sum(1).min() >= 4 and mask.sum(1).max() <= 12


def test_seq_acc_and_stats_bookkeeping():
    from common.metrics import seq_acc
    from models.shelgon3 import Trainer as T
    a = torch.tensor([[1, 2, 3], [4, 5, 6]]); b = torch.tensor([[1, 0, 3], [4, 5, 0]])
    per_batch, per_sentence = seq_acc(a, b)
    assert abs(per_batch.item() - 4 / 6) < 1e-6 and torch.allclose(per_sentence, torch.tensor([2 / 3, 2 / 3]))
    with pytest.raises(AssertionError):
        seq_acc(a.float(), b)
    run, best = T.init_stats_run(), T.init_stats_best()
    for loss, n in ((2.0, 4), (1.0, 12)):
        step = {"loss_recon_step": torch.tensor(loss), "loss_vq_step": torch.tensor(0.5), "metric_perp_step": torch.tensor(8.0),
                "loss_full_step": torch.tensor(loss + 0.5), "metric_acc_step": torch.tensor(0.25), "padding_tokens_pct_step": -69}
        run = T.end_of_step_stats_update(run, step, n)
    run, best = T.end_of_epoch_stats_update(run, best, 16, 2)
    assert abs(run["loss_recon_run"] - 1.25) < 1e-6 and abs(run["metric_acc_run"] - 25.0) < 1e-5
    assert best["loss_recon_is_best"] and best["metric_perp_is_best"] and best["loss_recon_best"] == run["loss_recon_run"]
    run2 = dict(run); run2["loss_recon_run"] = 9.0 * 16
    for k in ("loss_vq_run", "metric_perp_run", "loss_full_run", "metric_acc_run"):
        run2[k] = run2[k] * 16
    _, best = T.end_of_epoch_stats_update(run2, best, 16, 2)
    assert not best["loss_recon_is_best"]
    d = T.create_wandb_log_dict(3, run, "val")
    assert d["epoch"] == 3 and "val/loss_vq" in d and "padding_tokens_pct/val" in d


def test_freeze_modes_match_reference_counts():
    """set_mode semantics of models/bagon/Bagon.py:126-179 on a tiny bert2bert."""
    from models.bagon.Bagon import Bagon
    from common.model_utils import n_trainable_params
    m = Bagon("kvq-bert-tiny", "kvq-bert-tiny", cross_attn_make_trainable=True)
    full = n_trainable_params(m)
    m.set_mode("dec-head-ft")
    assert n_trainable_params(m.encoder) == 0
    head = m.decoder.cls.predictions
    expect = sum(p.numel() for p in {id(p): p for mod in (head.transform.dense, head.decoder) for p in mod.parameters()}.values())
    expect += sum(p.numel() for layer in m.decoder.bert.encoder.layer for p in layer.crossattention.parameters())
    assert n_trainable_params(m.decoder) == expect
    assert m.decoder.bert.embeddings.word_embeddings.weight.requires_grad            # tied to the LM head
    m.set_mode("enc-head-ft-dec-head-ft")
    assert all(p.requires_grad for p in m.encoder.encoder.layer[-1].parameters())
    assert not any(p.requires_grad for p in m.encoder.encoder.layer[0].parameters())
    m.set_mode("vq-ft")
    assert n_trainable_params(m) == 0 and full > 0
    with pytest.raises(ValueError):
        m.set_mode("nope")
    summary = m.model_params_summary_dict()
    assert set(summary) == {"encoder", "decoder"} and summary["encoder"]["n_params"] > 0


def test_bert_plan_equals_huggingface_forward():
    """kvq.bert (fused QKV / KV projections, explicit masks) against HF's own forward -- the third-party part of the
    reference -- in f32 eval mode on CPU: same logits."""
    from models.bagon.Bagon import Bagon
    torch.manual_seed(0)
    m = Bagon("kvq-bert-tiny", "kvq-bert-tiny", compute_dtype=torch.float32).eval()
    ids = torch.randint(1, 2048, (3, 12)); ids[0, 7:] = 0; ids[1, 4:] = 0
    mask = (ids != 0).long()
    with torch.no_grad():
        ours = m(ids, mask, ids, mask)
        m.backend = "hf"
        ref = m(ids, mask, ids, mask)
    assert ours.shape == (3, 12, 2048)
    torch.testing.assert_close(ours, ref, rtol=1e-5, atol=1e-5)
    # gradient flows to every parameter that HF's forward touches
    m.backend = "kvq"; m.train()
    m(ids, mask, ids, mask).square().mean().backward()
    missing = [n for n, p in m.named_parameters() if p.grad is None and "pooler" not in n]
    assert not missing, missing


def test_bench_refuses_to_report_a_number_for_a_job_size_it_did_not_run():
    """`python bench.py --gpus 4` without a launcher must start 4 ranks itself or FAIL -- never print a 1-GPU line with rc 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KVQ_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() < 4:
        assert r.returncode != 0 and "GPU(s) are visible" in r.stderr and '"metric"' not in r.stdout
    # launched by a launcher with a different world size: refuse as well
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout) and '"metric"' not in r.stdout
