"""End-to-end runs of the reference-shaped entry points (models/shelgon3/main.py, models/bagon/main.py) on a tiny
configuration: dataset synthesis -> split -> loaders -> model -> train (engine / autograd path) -> best-val checkpoint ->
reload -> test -> artefacts on disk with the reference's names and keys."""
import glob
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kindergarten-vq-vae_amd")


def _run(script, tmp_path, extra):
    env = dict(os.environ)
    env.update({
        "PYTHONPATH": os.pathsep.join([PKG] + ([os.environ["PYTHONPATH"]] if os.environ.get("PYTHONPATH") else [])),
        "KVQ_SYNTHETIC_SENTENCES": "640", "KVQ_BATCH_SIZE": "32", "KVQ_N_EPOCHS": "2", "KVQ_TOKENIZED_SENTENCE_MAX_LENGTH": "12",
        "KVQ_ENCODER_MODEL_NAME": "'kvq-bert-tiny'", "KVQ_DECODER_MODEL_NAME": "'kvq-bert-tiny'", "KVQ_LR": "1e-3",
        "KVQ_RUNS_DIR": repr(str(tmp_path / "runs")),
    })
    data = str(tmp_path / "data")
    env.update(extra(data))
    r = subprocess.run([sys.executable, os.path.join(PKG, script)], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    runs = glob.glob(str(tmp_path / "runs" / "*"))
    assert len(runs) == 1
    return runs[0]


@pytest.mark.parametrize("use_engine", [True, False])
def test_shelgon_main_end_to_end(tmp_path, use_engine):
    run = _run("models/shelgon3/main.py", tmp_path, lambda d: {
        "KVQ_SENTENCES_PATH": repr(d + "/dSentences_sentences_clean.npy"),
        "KVQ_LATENT_CLASSES_LABELS_PATH": repr(d + "/dSentences_latent_classes_labels_clean.npy"),
        "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(d + "/dSentences_latent_classes_one_hot_clean.npy"),
        "KVQ_VQ_N_E": "32", "KVQ_VQ_E_DIM": "128", "KVQ_USE_ENGINE": str(use_engine),
        "KVQ_TOKEN_CACHE": str(use_engine)})          # engine run: splits pre-tokenised in HBM; autograd run: DataLoader + per-step tokenizer
    conf = json.load(open(run + "/run_conf.json"))
    assert conf["token_cache"] == use_engine
    assert conf["vq_n_e"] == 32 and conf["encoder_model_name"] == "kvq-bert-tiny" and "n_params" in conf
    ckpt = torch.load(run + "/shelgon_ckpt_loss_recon_val_best.pth", map_location="cpu")
    assert set(ckpt) == {"model_state_dict", "encoder_state_dict", "decoder_state_dict"}         # Trainer.py:240-249
    assert "vector_quantizer.embedding.weight" in ckpt["model_state_dict"]
    assert ckpt["model_state_dict"]["vector_quantizer.embedding.weight"].shape == (32, 128)
    logs = [json.loads(l) for l in open(run + "/metrics.jsonl")]
    stages = {k.split("/")[0] for l in logs for k in l if "/" in k}
    assert {"train", "val", "test"} <= stages
    tr = [l["train/loss_recon"] for l in logs if "train/loss_recon" in l]
    assert len(tr) == 2 and tr[1] < tr[0]                                                     # it learns
    import pandas as pd
    df = pd.read_feather(run + "/decoded_sentences.feather")
    assert {"epoch", "stage", "input_sentence", "recon_sentence"} <= set(df.columns) and len(df) > 0


def test_shelgon_main_with_fp8_forward_gemms(tmp_path):
    """FP8_FORWARD = True in models/shelgon3/config.py (BASELINE.json configs[4], extension): the entry point trains with every forward
    GEMM of 384 x 128 activations on the fp8 matrix cores, the fp8 copies written by their producers, the weights' mirror by Adam."""
    run = _run("models/shelgon3/main.py", tmp_path, lambda d: {
        "KVQ_SENTENCES_PATH": repr(d + "/dSentences_sentences_clean.npy"),
        "KVQ_LATENT_CLASSES_LABELS_PATH": repr(d + "/dSentences_latent_classes_labels_clean.npy"),
        "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(d + "/dSentences_latent_classes_one_hot_clean.npy"),
        "KVQ_VQ_N_E": "32", "KVQ_VQ_E_DIM": "128", "KVQ_FP8_FORWARD": "True"})
    conf = json.load(open(run + "/run_conf.json"))
    assert conf["fp8_forward"] is True and conf["use_engine"]
    logs = [json.loads(l) for l in open(run + "/metrics.jsonl")]
    tr = [l["train/loss_recon"] for l in logs if "train/loss_recon" in l]
    assert len(tr) == 2 and tr[1] < tr[0]
    assert {"val", "test"} <= {k.split("/")[0] for l in logs for k in l if "/" in k}


def test_shelgon_main_with_sentences_padded_to_40_tokens(tmp_path):
    """TOKENIZED_SENTENCE_MAX_LENGTH = 40: above the 32-token attention kernels, so the engine's step runs the blocked ones; the run
    trains and writes the usual artefacts."""
    run = _run("models/shelgon3/main.py", tmp_path, lambda d: {
        "KVQ_SENTENCES_PATH": repr(d + "/dSentences_sentences_clean.npy"),
        "KVQ_LATENT_CLASSES_LABELS_PATH": repr(d + "/dSentences_latent_classes_labels_clean.npy"),
        "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(d + "/dSentences_latent_classes_one_hot_clean.npy"),
        "KVQ_VQ_N_E": "32", "KVQ_VQ_E_DIM": "128", "KVQ_TOKENIZED_SENTENCE_MAX_LENGTH": "40"})
    conf = json.load(open(run + "/run_conf.json"))
    assert conf["tokenized_sentence_max_length"] == 40 and conf["use_engine"]
    logs = [json.loads(l) for l in open(run + "/metrics.jsonl")]
    tr = [l["train/loss_recon"] for l in logs if "train/loss_recon" in l]
    assert len(tr) == 2 and tr[1] < tr[0]
    assert os.path.exists(run + "/shelgon_ckpt_loss_recon_val_best.pth")


@pytest.mark.parametrize("use_engine,perturb", [(True, 0.0), (False, 0.0), (True, 0.1)])
def test_bagon_main_end_to_end(tmp_path, use_engine, perturb):
    """models/bagon/main.py on the TrainEngine (token cache, packed batches), on the autograd path (DataLoader + per-step
    tokenizer), and on the engine with token noise on BOTH sides (encoder ids != decoder ids in every step, Trainer.py:85,94)."""
    run = _run("models/bagon/main.py", tmp_path, lambda d: {
        "KVQ_DATASET_PATH": repr(d + "/dSentences_sentences_clean.npy"),
        "KVQ_LATENT_CLASSES_LABELS_PATH": repr(d + "/dSentences_latent_classes_labels_clean.npy"),
        "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(d + "/dSentences_latent_classes_one_hot_clean.npy"),
        "KVQ_MODEL_MODE": "'dec-head-ft'", "KVQ_USE_ENGINE": str(use_engine), "KVQ_TOKEN_CACHE": str(use_engine),
        "KVQ_ENCODER_PERTURB_TRAIN_PCT": str(perturb), "KVQ_DECODER_PERTURB_TRAIN_PCT": str(perturb),
        "KVQ_DECODER_PERTURB_VAL_PCT": str(perturb)})
    ckpt = torch.load(run + "/bagon_ckpt_loss_recon_val_best.pth", map_location="cpu")
    assert set(ckpt) == {"model_state_dict", "encoder_state_dict", "decoder_state_dict"}
    conf = json.load(open(run + "/run_conf.json"))
    assert conf["model_mode"] == "dec-head-ft" and conf["n_params"]["encoder"]["n_trainable_params"] == 0
    assert conf["use_engine"] == use_engine and conf["decoder_perturb_train_pct"] == perturb
    logs = [json.loads(l) for l in open(run + "/metrics.jsonl")]
    assert any("test/loss_recon" in l for l in logs)
    assert {"train/loss_recon", "train/loss_full", "train/acc", "padding_tokens_pct/train"} <= {k for l in logs for k in l}     # Trainer.py:196-203
    tr = [l["train/loss_recon"] for l in logs if "train/loss_recon" in l]
    assert len(tr) == 2 and tr[1] < tr[0]                                                     # it learns
    import pandas as pd
    df = pd.read_feather(run + "/decoded_sentences.feather")
    assert {"epoch", "stage", "input_sentence", "recon_sentence", "sentence_acc", "sentence_type", "verb_tense"} <= set(df.columns)
    assert len(df) > 0 and df["sentence_acc"].between(0, 1).all()
    if perturb:
        return
    # common/test_checkpoint_validity.py of the reference: load that checkpoint into a fresh Bagon and reconstruct the probe sentences
    env = dict(os.environ, PYTHONPATH=PKG, KVQ_CKPT_PATH=repr(run + "/bagon_ckpt_loss_recon_val_best.pth"),
               KVQ_ENCODER_MODEL_NAME="'kvq-bert-tiny'", KVQ_DECODER_MODEL_NAME="'kvq-bert-tiny'")
    r = subprocess.run([sys.executable, os.path.join(PKG, "common/test_checkpoint_validity.py")], env=env, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "'he accepted the payment' -> " in r.stdout and "logits (3, 6, 2048)" in r.stdout and "finite: True" in r.stdout


def test_codebook_init_script_feeds_shelgon_main(tmp_path):
    """models/shelgon3/vq_codebook_init_weights.py (GPU k-means on encoder outputs) writes the reference's file format, and
    shelgon3/main.py starts from it (VQ_CODEBOOK_INIT_VALUES_PATH)."""
    data = str(tmp_path / "data")
    common = lambda d: {
        "KVQ_SENTENCES_PATH": repr(d + "/dSentences_sentences_clean.npy"),
        "KVQ_LATENT_CLASSES_LABELS_PATH": repr(d + "/dSentences_latent_classes_labels_clean.npy"),
        "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(d + "/dSentences_latent_classes_one_hot_clean.npy"),
        "KVQ_VQ_N_E": "9", "KVQ_VQ_E_DIM": "128"}                                                   # N_E = 9 as in the reference script
    env = dict(os.environ)
    env.update({"PYTHONPATH": PKG, "KVQ_SYNTHETIC_SENTENCES": "640", "KVQ_TOKENIZED_SENTENCE_MAX_LENGTH": "12",
                "KVQ_ENCODER_MODEL_NAME": "'kvq-bert-tiny'", "KVQ_DECODER_MODEL_NAME": "'kvq-bert-tiny'",
                "KVQ_CODEBOOK_INIT_OUT": str(tmp_path / "init.pth")})
    env.update(common(data))
    r = subprocess.run([sys.executable, os.path.join(PKG, "models/shelgon3/vq_codebook_init_weights.py")], env=env, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    blob = torch.load(str(tmp_path / "init.pth"), map_location="cpu")
    assert set(blob) == {"codebook_init_values", "encoder_model_name", "decoder_model_name", "tokenizer_name"}     # reference :104-111
    cb = blob["codebook_init_values"]
    assert cb.shape == (9, 128) and cb.dtype == torch.float32 and torch.isfinite(cb).all() and cb.unique(dim=0).shape[0] == 9
    run = _run("models/shelgon3/main.py", tmp_path, lambda d: dict(common(d), KVQ_N_EPOCHS="1",
                                                                   KVQ_VQ_CODEBOOK_INIT_VALUES_PATH=repr(str(tmp_path / "init.pth"))))
    conf = json.load(open(run + "/run_conf.json"))
    assert conf["vq_n_e"] == 9 and conf["vq_codebook_init_values_path"].endswith("init.pth")


def test_shelgon_main_gumbel_mode(tmp_path):
    """VQ_MODE = "GumbelQuantizer" (reference main.py:68-73): the run trains on the TrainEngine (Gumbel row kernel between the two BERT stacks) and writes the usual artefacts."""
    run = _run("models/shelgon3/main.py", tmp_path, lambda d: {
        "KVQ_SENTENCES_PATH": repr(d + "/dSentences_sentences_clean.npy"),
        "KVQ_LATENT_CLASSES_LABELS_PATH": repr(d + "/dSentences_latent_classes_labels_clean.npy"),
        "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(d + "/dSentences_latent_classes_one_hot_clean.npy"),
        "KVQ_VQ_MODE": "'GumbelQuantizer'", "KVQ_VQ_N_E": "16", "KVQ_VQ_E_DIM": "128", "KVQ_ENC_OUT_SIZE": "128", "KVQ_N_EPOCHS": "1"})
    ckpt = torch.load(run + "/shelgon_ckpt_loss_recon_val_best.pth", map_location="cpu")
    sd = ckpt["model_state_dict"]
    assert sd["vector_quantizer.proj.weight"].shape == (16, 128, 1) and sd["vector_quantizer.embed.weight"].shape == (16, 128)
    logs = [json.loads(l) for l in open(run + "/metrics.jsonl")]
    perp = [l["train/metric_perp"] for l in logs if "train/metric_perp" in l]
    assert perp and 1 <= perp[0] <= 16            # "perplexity" of this mode = number of codes in use (Shelgon.py:63-65)


def test_shelgon_main_multi_codebook_with_ema(tmp_path):
    """Extensions of BASELINE.json configs[4], selectable from config.py and off by default: VQ_MODE = "MultiVectorQuantizer"
    (4 factor slices -> one grouped launch) with VQ_EMA_DECAY: the run trains on the TrainEngine, the codebook moves by EMA only."""
    run = _run("models/shelgon3/main.py", tmp_path, lambda d: {
        "KVQ_SENTENCES_PATH": repr(d + "/dSentences_sentences_clean.npy"),
        "KVQ_LATENT_CLASSES_LABELS_PATH": repr(d + "/dSentences_latent_classes_labels_clean.npy"),
        "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(d + "/dSentences_latent_classes_one_hot_clean.npy"),
        "KVQ_VQ_MODE": "'MultiVectorQuantizer'", "KVQ_VQ_N_FACTORS": "4", "KVQ_VQ_EMA_DECAY": "0.99", "KVQ_VQ_N_E": "16",
        "KVQ_VQ_E_DIM": "128", "KVQ_N_EPOCHS": "1"})
    conf = json.load(open(run + "/run_conf.json"))
    assert conf["vq_mode"] == "MultiVectorQuantizer" and conf["vq_n_factors"] == 4 and conf["vq_ema_decay"] == 0.99
    sd = torch.load(run + "/shelgon_ckpt_loss_recon_val_best.pth", map_location="cpu")["model_state_dict"]
    assert sd["vector_quantizer.embedding.weight"].shape == (4 * 16, 32) and sd["vector_quantizer.ema_n"].shape == (4, 16)
    assert not torch.allclose(sd["vector_quantizer.ema_n"], torch.ones(4, 16))          # the EMA statistics moved
    logs = [json.loads(l) for l in open(run + "/metrics.jsonl")]
    assert any("train/loss_vq" in l for l in logs)


@pytest.mark.parametrize("script,ckpt", [("models/bagon/main.py", "bagon_ckpt_loss_recon_val_best.pth"),
                                         ("models/shelgon3/main.py", "shelgon_ckpt_loss_recon_val_best.pth")])
def test_main_with_two_ranks(tmp_path, script, ckpt):
    """The advertised multi-GPU launch line of both entry points, two ranks (gloo, sharing the one GPU): train, rank 0 writes the
    best-val checkpoint, EVERY rank reloads it and runs the test stage on its shard -- the stage's all-reduce of the statistics
    then has a partner on every rank (only rank 0 used to enter it and the run ended in mismatched collectives) --
    and rank 0 writes all ranks' decoded sentences."""
    env = dict(os.environ)
    data = str(tmp_path / "data")
    env.update({
        "PYTHONPATH": PKG, "KVQ_DIST_BACKEND": "gloo",
        "KVQ_SYNTHETIC_SENTENCES": "640", "KVQ_BATCH_SIZE": "16", "KVQ_N_EPOCHS": "2", "KVQ_TOKENIZED_SENTENCE_MAX_LENGTH": "12",
        "KVQ_ENCODER_MODEL_NAME": "'kvq-bert-tiny'", "KVQ_DECODER_MODEL_NAME": "'kvq-bert-tiny'", "KVQ_LR": "1e-3",
        "KVQ_RUNS_DIR": repr(str(tmp_path / "runs")),
        "KVQ_DATASET_PATH": repr(data + "/dSentences_sentences_clean.npy"), "KVQ_SENTENCES_PATH": repr(data + "/dSentences_sentences_clean.npy"),
        "KVQ_LATENT_CLASSES_LABELS_PATH": repr(data + "/dSentences_latent_classes_labels_clean.npy"),
        "KVQ_LATENT_CLASSES_ONE_HOT_PATH": repr(data + "/dSentences_latent_classes_one_hot_clean.npy"),
        "KVQ_VQ_N_E": "32", "KVQ_VQ_E_DIM": "128", "KVQ_MODEL_MODE": "'full'"})
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(PKG, script)], env=env, cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    runs = glob.glob(str(tmp_path / "runs" / "*"))
    assert len(runs) == 1, runs                                      # one run directory for both ranks (the run id is rank 0's)
    run = runs[0]
    conf = json.load(open(run + "/run_conf.json"))
    assert conf["world_size"] == 2 and os.path.exists(f"{run}/{ckpt}")
    logs = [json.loads(l) for l in open(run + "/metrics.jsonl")]
    assert any("test/loss_recon" in l for l in logs)
    tr = [l["train/loss_recon"] for l in logs if "train/loss_recon" in l]
    assert len(tr) == 2 and tr[1] < tr[0]
    import pandas as pd
    df = pd.read_feather(run + "/decoded_sentences.feather")
    assert len(df) > 0 and (df["stage"] == "test").any()
