"""step / train / test of the plain Bagon run -- counterpart of models/bagon/Trainer.py:65-510.

Same skeleton as the Shelgon trainer minus the VQ terms: two tokenizers (encoder / decoder side), optional token
noise on either side (`replace_pct_rand_values`, off at 0 %), reconstruction loss and token accuracy from the fused
loss kernel, checkpoints on best val loss_recon / metric_acc (`bagon_ckpt_{metric}_{stage}_best.pth`,
Trainer.py:290-296)."""
from __future__ import annotations

from itertools import islice

import numpy as np
import torch
from torch import no_grad, save

from common.consts import *  # noqa: F401,F403
from common.tensor_utils import replace_pct_rand_values


def step(device, model, tokenizer_encoder, tokenizer_decoder, tokenizer_add_special_tokens: bool, opt, lr_sched, batch,
         encoder_perturb_pct: float, decoder_perturb_pct: float, vocab_size_encoder: int, vocab_size_decoder: int,
         stage: str, console=None, max_length: int = 12, grad_sync=None):
    sentences = batch["sentence"]
    enc = tokenizer_encoder(sentences, return_tensors="pt", padding="max_length", max_length=max_length,
                            add_special_tokens=tokenizer_add_special_tokens)
    dec = enc if tokenizer_decoder is tokenizer_encoder else tokenizer_decoder(
        sentences, return_tensors="pt", padding="max_length", max_length=max_length, add_special_tokens=tokenizer_add_special_tokens)
    enc_ids, enc_mask = enc.input_ids.to(device), enc.attention_mask.to(device)
    dec_ids, dec_mask = dec.input_ids.to(device), dec.attention_mask.to(device)
    enc_in = replace_pct_rand_values(enc_ids, encoder_perturb_pct, 0, vocab_size_encoder)      # Trainer.py:85
    dec_in = replace_pct_rand_values(dec_ids, decoder_perturb_pct, 0, vocab_size_decoder)      # Trainer.py:94
    if dec_in is dec_ids:
        loss_recon_step, acc_step, recon_ids = model.forward_loss(enc_in, enc_mask, dec_ids, dec_mask)
    else:   # noisy decoder input, clean target: score against the clean ids
        from kvq import bert as kbert
        from kvq.functional import fused_cross_entropy
        hidden = model.decode_hidden(model.encode(enc_in, enc_mask), dec_in, dec_mask)
        loss_recon_step, acc_step, recon_ids = fused_cross_entropy(
            kbert.lm_head_logits(model.decoder, hidden, model.compute_dtype), dec_ids, inplace_backward=True)
    if opt is not None:
        grad_sync.zero_grad() if grad_sync is not None else opt.zero_grad()
        loss_recon_step.backward()
        if grad_sync is not None:
            grad_sync.finish()
        opt.step()
        if lr_sched is not None:
            lr_sched.step()
    return {"loss_recon_step": loss_recon_step.detach(), "metric_acc_step": acc_step.detach()}, dec_ids, recon_ids


def init_stats_best():
    return {"loss_recon_best": np.inf, "loss_recon_is_best": False, "metric_acc_best": 0, "metric_acc_is_best": False}


def init_stats_run():
    return {"loss_recon_run": 0, "metric_acc_run": 0}


def end_of_step_stats_update(run, stats_step, n):
    run["loss_recon_run"] += stats_step["loss_recon_step"] * n
    run["metric_acc_run"] += stats_step["metric_acc_step"] * n * 1e2
    return run


def end_of_epoch_stats_update(run, best, n_els_epoch, n_steps):
    run = {k: float(v) / max(n_els_epoch, 1) for k, v in run.items()}
    best["loss_recon_is_best"] = run["loss_recon_run"] < best["loss_recon_best"]
    best["metric_acc_is_best"] = run["metric_acc_run"] > best["metric_acc_best"]
    if best["loss_recon_is_best"]:
        best["loss_recon_best"] = run["loss_recon_run"]
    if best["metric_acc_is_best"]:
        best["metric_acc_best"] = run["metric_acc_run"]
    return run, best


def _save_ckpt(model, path):
    save({"model_state_dict": model.state_dict(), "encoder_state_dict": model.encoder.state_dict(),
          "decoder_state_dict": model.decoder.state_dict()}, path)


def checkpoint(best, model, checkpoint_dir, stage):
    if best["loss_recon_is_best"]:
        _save_ckpt(model, f"{checkpoint_dir}/bagon_ckpt_loss_recon_{stage}_best.pth")
    if best["metric_acc_is_best"]:
        _save_ckpt(model, f"{checkpoint_dir}/bagon_ckpt_metric_acc_{stage}_best.pth")


def _stage(stage, device, loader, n_batches, model, toks, add_special, opt, lr_sched, pcts, vocabs, decode_into, epoch,
           max_length, grad_sync):
    run, n_els, n_steps = init_stats_run(), 0, 0
    for batch in islice(loader, n_batches):
        n = len(batch["sentence"])
        n_els += n
        n_steps += 1
        with (torch.enable_grad() if opt is not None else no_grad()):
            st, ids, recon = step(device, model, toks[0], toks[1], add_special, opt, lr_sched, batch, pcts[0], pcts[1],
                                  vocabs[0], vocabs[1], stage, max_length=max_length, grad_sync=grad_sync)
        if decode_into is not None:
            for i, r in zip(toks[1].batch_decode(ids.cpu()), toks[1].batch_decode(recon.cpu())):
                decode_into.append({"epoch": epoch, "stage": stage, "input_sentence": i, "recon_sentence": r})
        run = end_of_step_stats_update(run, st, n)
    return run, n_els, n_steps


def train(console, device, dl_train, dl_val, n_batches_train, n_batches_val, model, tokenizer_encoder, tokenizer_decoder,
          tokenizer_add_special_tokens, n_epochs_to_decode_after, decoded_sentences, opt, lr_sched, n_epochs,
          encoder_perturb_train_pct, decoder_perturb_train_pct, encoder_perturb_val_pct, decoder_perturb_val_pct,
          vocab_size_encoder, vocab_size_decoder, wandb_run, run_path, export_checkpoint, max_length=12, grad_sync=None,
          is_main=True):
    best_tr, best_va = init_stats_best(), init_stats_best()
    toks, vocabs = (tokenizer_encoder, tokenizer_decoder), (vocab_size_encoder, vocab_size_decoder)
    hist = []
    for epoch in range(1, n_epochs + 1):
        dec = decoded_sentences if epoch % n_epochs_to_decode_after == 0 else None
        model.train()
        run, n, s = _stage("train", device, dl_train, n_batches_train, model, toks, tokenizer_add_special_tokens, opt, lr_sched,
                           (encoder_perturb_train_pct, decoder_perturb_train_pct), vocabs, dec, epoch, max_length, grad_sync)
        tr, best_tr = end_of_epoch_stats_update(run, best_tr, n, s)
        wandb_run.log({"epoch": epoch, "train/loss_recon": tr["loss_recon_run"], "train/acc": tr["metric_acc_run"]})
        model.eval()
        run, n, s = _stage("val", device, dl_val, n_batches_val, model, toks, tokenizer_add_special_tokens, None, None,
                           (encoder_perturb_val_pct, decoder_perturb_val_pct), vocabs, dec, epoch, max_length, None)
        va, best_va = end_of_epoch_stats_update(run, best_va, n, s)
        wandb_run.log({"epoch": epoch, "val/loss_recon": va["loss_recon_run"], "val/acc": va["metric_acc_run"]})
        if console is not None:
            console.print(f"[bold {COLOR_EPOCH}]{epoch:03d}[/bold {COLOR_EPOCH}] | train loss_recon {tr['loss_recon_run']:.6f} "
                          f"acc {tr['metric_acc_run']:.4f}% | val loss_recon {va['loss_recon_run']:.6f} acc {va['metric_acc_run']:.4f}%")
        if export_checkpoint and is_main:
            checkpoint(best_va, model, run_path, "val")
        hist.append((tr, va))
    return hist


def test(console, device, dl_test, n_batches_test, model, tokenizer_encoder, tokenizer_decoder, tokenizer_add_special_tokens,
         encoder_perturb_test_pct, decoder_perturb_test_pct, vocab_size_encoder, vocab_size_decoder, decoded_sentences, epoch,
         wandb_run, max_length=12):
    model.eval()
    run, n, s = _stage("test", device, dl_test, n_batches_test, model, (tokenizer_encoder, tokenizer_decoder),
                       tokenizer_add_special_tokens, None, None, (encoder_perturb_test_pct, decoder_perturb_test_pct),
                       (vocab_size_encoder, vocab_size_decoder), decoded_sentences, epoch, max_length, None)
    te, _ = end_of_epoch_stats_update(run, init_stats_best(), n, s)
    wandb_run.log({"epoch": epoch, "test/loss_recon": te["loss_recon_run"], "test/acc": te["metric_acc_run"]})
    if console is not None:
        console.print(f"    | test loss_recon {te['loss_recon_run']:.6f} acc {te['metric_acc_run']:.4f}%")
    return te
