"""step / train / test of the plain Bagon run -- counterpart of models/bagon/Trainer.py:65-510.

Kept from the reference: function names and keyword lists (two tokenizers with their own add-special-tokens flag and padded
length, the four perturbation percentages, vocabulary sizes), the per-step stats dict and its keys (`loss_recon_step`,
`loss_full_step`, `metric_acc_step_per_batch`, `metric_acc_step_per_sentence`, `padding_tokens_pct_step` = -69), the 5-tuple
`step()` returns (stats, encoder ids, decoder ids, recon ids, latent class labels), running / best bookkeeping and its keys,
`decode_sentences` rows (with the per-sentence accuracy and the explicit latent labels), checkpoint names
(`bagon_ckpt_{loss_recon,metric_acc}_{stage}_best.pth`) and keys.  As in the reference (Trainer.py:94,103-106) the loss and the
accuracy score the logits against the decoder's input ids AFTER their perturbation.

Two execution paths for one step:
  * engine (default, `engine=` a kvq.engine.TrainEngine): forward, loss, backward, gradient exchange, Adam and the per-step
    scheduler tick are the engine's hand-written HIP schedule (kvq/engine.py), replayed from hipGraphs; the encoder and decoder
    sides keep their own ids (the decoder's feed the decoder's embedding table, the loss target and its word-embedding gradient);
  * autograd (`engine=None`): kvq/bert.py under torch autograd + the fused loss kernel + torch.optim.Adam -- the checker.

Deviations from reference quirks, documented in DESIGN.md §6: running means are weighted by the true batch size (the reference
weights by len(batch-dict) == 3, which cancels out in the means), stats stay on the device until the end of the epoch, batches
are consumed lazily instead of `list(dl_train)`, the "val" checkpoint is decided by the VAL best flags (the reference passes the
TRAIN flags, Trainer.py:437), the progress display is optional (`prg=None`)."""
from __future__ import annotations

import time as _time
from itertools import islice

import numpy as np
import torch
from torch import Tensor, no_grad, save

from common.consts import *  # noqa: F401,F403
from common.tensor_utils import replace_pct_rand_values


def _tokenize(batch, tokenizer, add_special_tokens, max_length, device, side):
    """Trainer.py:78-84 / :87-93.  Batches that already carry ids (pre-tokenised cache) skip the tokenizer."""
    key = f"input_ids_{side}" if f"input_ids_{side}" in batch else "input_ids"
    if key in batch:
        ids, mask = batch[key], batch[key.replace("input_ids", "attention_mask")]
    else:
        tok = tokenizer(batch["sentence"], return_tensors="pt", padding="max_length", max_length=max_length,
                        add_special_tokens=add_special_tokens)
        ids, mask = tok.input_ids, tok.attention_mask
    return ids.to(device, non_blocking=True), mask.to(device, non_blocking=True)


def step(device, model, tokenizer_encoder, tokenizer_decoder,
         tokenizer_encoder_add_special_tokens: bool, tokenized_encoder_sentence_max_length: int,
         tokenizer_decoder_add_special_tokens: bool, tokenized_decoder_sentence_max_length: int,
         encoder_perturb_pct: float, decoder_perturb_pct: float, opt, lr_sched, batch,
         vocab_size_encoder: int, vocab_size_decoder: int, console=None, grad_sync=None, engine=None):
    input_ids_encoder, attention_mask_encoder = _tokenize(batch, tokenizer_encoder, tokenizer_encoder_add_special_tokens,
                                                          tokenized_encoder_sentence_max_length, device, "encoder")
    same_side = tokenizer_decoder is tokenizer_encoder and tokenizer_decoder_add_special_tokens == tokenizer_encoder_add_special_tokens \
        and tokenized_decoder_sentence_max_length == tokenized_encoder_sentence_max_length and "input_ids_decoder" not in batch
    if same_side:                                   # one tokenizer call serves both sides (bert2bert, the reference's default)
        input_ids_decoder, attention_mask_decoder = input_ids_encoder, attention_mask_encoder
    else:
        input_ids_decoder, attention_mask_decoder = _tokenize(batch, tokenizer_decoder, tokenizer_decoder_add_special_tokens,
                                                              tokenized_decoder_sentence_max_length, device, "decoder")
    input_ids_encoder = replace_pct_rand_values(input_ids_encoder, encoder_perturb_pct, 0, vocab_size_encoder)      # Trainer.py:85
    input_ids_decoder = replace_pct_rand_values(input_ids_decoder, decoder_perturb_pct, 0, vocab_size_decoder)      # Trainer.py:94
    labels = batch.get("latent_classes_labels") if isinstance(batch, dict) else None

    if engine is not None:
        # the decoder side is handed over only when it differs from the encoder's: the autoencoding step then shares one sort
        dec = {} if input_ids_decoder is input_ids_encoder else dict(dec_ids=input_ids_decoder, dec_mask=attention_mask_decoder)
        if opt is not None:
            packed = batch.get("packed") if isinstance(batch, dict) and not dec and encoder_perturb_pct == 0 else None
            out = engine.train_step(input_ids_encoder, attention_mask_encoder, prepared=packed, **dec)
        else:
            out = engine.eval_step(input_ids_encoder, attention_mask_encoder, **dec)
        loss_recon_step, acc_batch, acc_sentence, recon_ids = out["loss_recon"], out["acc"], out["acc_per_sentence"], out["recon_ids"]
    else:
        loss_recon_step, acc_batch, recon_ids = model.forward_loss(input_ids_encoder, attention_mask_encoder,
                                                                   input_ids_decoder, attention_mask_decoder)
        acc_sentence = torch.eq(recon_ids, input_ids_decoder).to(torch.float32).mean(dim=-1)      # common/metrics.py:32
        if opt is not None:                                                                        # Trainer.py:115-122
            grad_sync.zero_grad() if grad_sync is not None else opt.zero_grad()
            loss_recon_step.backward()
            if grad_sync is not None:
                grad_sync.finish()
            opt.step()
            if lr_sched is not None:
                lr_sched.step()
    loss_recon_step = loss_recon_step.detach()
    return {
        "loss_recon_step": loss_recon_step,
        "loss_full_step": loss_recon_step,                                                         # Trainer.py:111
        "metric_acc_step_per_batch": acc_batch.detach(),
        "metric_acc_step_per_sentence": acc_sentence.detach(),
        "padding_tokens_pct_step": -69,
    }, input_ids_encoder, input_ids_decoder, recon_ids, labels


def end_of_step_stats_update(stats_stage_run: dict, stats_step: dict, n_els_batch: int):
    stats_stage_run["loss_recon_run"] += stats_step["loss_recon_step"] * n_els_batch
    stats_stage_run["loss_full_run"] += stats_step["loss_full_step"] * n_els_batch
    stats_stage_run["metric_acc_run"] += stats_step["metric_acc_step_per_batch"] * n_els_batch * 1e2
    stats_stage_run["padding_tokens_pct_run"] += stats_step["padding_tokens_pct_step"]
    return stats_stage_run


_LOWER_IS_BETTER = {"loss_recon": True, "loss_full": True, "metric_acc": False}


def _sum_over_ranks(stats_stage_run: dict, n_els_epoch: int):
    """Data-parallel runs: one small all-reduce per stage and epoch turns every rank's sums into sums over the whole split."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return stats_stage_run, n_els_epoch
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    keys = list(_LOWER_IS_BETTER)
    vec = torch.stack([torch.as_tensor(stats_stage_run[f"{k}_run"], dtype=torch.float64, device=dev).reshape(()) for k in keys] +
                      [torch.tensor(float(n_els_epoch), dtype=torch.float64, device=dev)])
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    for k, v in zip(keys, vec[:-1].tolist()):
        stats_stage_run[f"{k}_run"] = v
    return stats_stage_run, int(round(vec[-1].item()))


def end_of_epoch_stats_update(stats_stage_run: dict, stats_stage_best: dict, n_els_epoch: int, n_steps: int):
    stats_stage_run, n_els_epoch = _sum_over_ranks(stats_stage_run, n_els_epoch)
    for key in _LOWER_IS_BETTER:
        stats_stage_run[f"{key}_run"] = float(stats_stage_run[f"{key}_run"]) / max(n_els_epoch, 1)      # one sync per epoch
    stats_stage_run["padding_tokens_pct_run"] /= max(n_steps, 1)
    for key, lower in _LOWER_IS_BETTER.items():
        cur, best = stats_stage_run[f"{key}_run"], stats_stage_best[f"{key}_best"]
        is_best = cur < best if lower else cur > best
        stats_stage_best[f"{key}_is_best"] = is_best
        if is_best:
            stats_stage_best[f"{key}_best"] = cur
    return stats_stage_run, stats_stage_best


def end_of_epoch_print(stats_stage_run, stats_stage_best, console, epoch, print_epoch, stat_color, stat_emojis, print_new_line):
    if console is None:
        return
    head = f"[bold {COLOR_EPOCH}]{epoch:03d}[/bold {COLOR_EPOCH}] | " if print_epoch else "    | "
    console.print(
        head +
        f"loss_recon: [bold {stat_color}] {stats_stage_run['loss_recon_run']:08.6f}[/bold {stat_color}] "
        f"{stat_emojis[1] if stats_stage_best['loss_recon_is_best'] else '  '} | "
        f"acc: [bold {stat_color}]{stats_stage_run['metric_acc_run']:08.6f}%[/bold {stat_color}] "
        f"{stat_emojis[2] if stats_stage_best['metric_acc_is_best'] else '  '} | " + ("\n" if print_new_line else ""))


def init_stats_best():
    return {"loss_recon_best": np.inf, "loss_recon_is_best": False, "loss_full_best": np.inf, "loss_full_is_best": False,
            "metric_acc_best": 0, "metric_acc_is_best": False}


def init_stats_run():
    return {"loss_recon_run": 0, "loss_full_run": 0, "metric_acc_run": 0, "padding_tokens_pct_run": 0}


def create_wandb_log_dict(epoch: int, stats_stage_run: dict, stage: str):
    return {"epoch": epoch, f"{stage}/loss_recon": stats_stage_run["loss_recon_run"], f"{stage}/loss_full": stats_stage_run["loss_full_run"],
            f"{stage}/acc": stats_stage_run["metric_acc_run"], f"padding_tokens_pct/{stage}": stats_stage_run["padding_tokens_pct_run"]}


# the five generative factors the reference spells out per decoded sentence (Trainer.py:205-253): (row key, {label: name})
_EXPLICIT = [("sentence_type", {0: "declarative", 1: "interrogative"}),
             ("grammatical_number_person", {0: "1st", 1: "2nd", 2: "3rd"}),
             ("sentence_negation", {0: "affirmative", 1: "negative"}),
             ("verb_tense", {0: "past", 1: "present", 2: "future"}),
             ("sentence_style", {0: "not_progressive", 1: "progressive"})]


def explicit_latent_classes_labels(latent_classes_labels: Tensor, console=None):
    """Labels outside the reference's tables (the synthetic corpus has other factor cardinalities) are kept as their number."""
    vals = [int(v) for v in latent_classes_labels[:len(_EXPLICIT)]]
    return {key: names.get(v, str(v)) for (key, names), v in zip(_EXPLICIT, vals)}


def decode_sentences(input_ids_encoder, input_ids_decoder, recon_ids, latent_classes_labels, stats_step, tokenizer_encoder,
                     tokenizer_decoder, decoded_sentences: list, epoch: int, stage: str, console=None):
    def _decode(tok, ids):
        try:
            return tok.batch_decode(sequences=ids.cpu(), skip_special_tokens=True)
        except TypeError:                                           # the offline word tokenizer has no such switch
            return tok.batch_decode(ids.cpu())
    input_decoded, recon_decoded = _decode(tokenizer_encoder, input_ids_encoder), _decode(tokenizer_decoder, recon_ids)
    accs = stats_step["metric_acc_step_per_sentence"].cpu().tolist()
    labels = latent_classes_labels.cpu() if latent_classes_labels is not None else [None] * len(accs)
    for i, r, a, l in zip(input_decoded, recon_decoded, accs, labels):
        row = {"epoch": epoch, "stage": stage, "input_sentence": i, "recon_sentence": r, "sentence_acc": a}
        if l is not None:
            row.update(explicit_latent_classes_labels(l, console))
        decoded_sentences.append(row)


def _save_ckpt(model, checkpoint_file_path: str, stage: str = ""):
    save({"model_state_dict": model.state_dict(), "encoder_state_dict": model.encoder.state_dict(),
          "decoder_state_dict": model.decoder.state_dict()}, checkpoint_file_path)


def checkpoint(stats_best: dict, model, checkpoint_dir: str, stage: str):
    if stats_best["loss_recon_is_best"]:
        _save_ckpt(model, f"{checkpoint_dir}/bagon_ckpt_loss_recon_{stage}_best.pth", stage)
    if stats_best["metric_acc_is_best"]:
        _save_ckpt(model, f"{checkpoint_dir}/bagon_ckpt_metric_acc_{stage}_best.pth", stage)


def _batch_size(batch) -> int:
    """Sentences in a batch, whichever form it takes: the DataLoader's dict of strings, the token cache's dict of ids, or
    ids tokenised per side (input_ids_encoder / input_ids_decoder, which step() accepts as well)."""
    if isinstance(batch, dict):
        if "sentence" in batch:
            return len(batch["sentence"])
        for key in ("input_ids", "input_ids_encoder", "input_ids_decoder"):
            if key in batch:
                return int(batch[key].shape[0])
        raise KeyError(f"batch without sentences or ids: keys {sorted(batch)}")
    return len(batch)


def _stage(stage, loader, n_batches, step_kw, opt, lr_sched, pcts, decode_into, epoch, grad_sync, engine, prg=None, task=None):
    run, n_els, n_steps = init_stats_run(), 0, 0
    for batch in islice(loader, n_batches):
        n = _batch_size(batch)
        n_els += n
        n_steps += 1
        with (torch.enable_grad() if opt is not None else no_grad()):
            st, ids_enc, ids_dec, recon, labels = step(opt=opt, lr_sched=lr_sched, batch=batch, encoder_perturb_pct=pcts[0],
                                                       decoder_perturb_pct=pcts[1], grad_sync=grad_sync, engine=engine, **step_kw)
        if decode_into is not None:
            decode_sentences(ids_enc, ids_dec, recon, labels, st, step_kw["tokenizer_encoder"], step_kw["tokenizer_decoder"],
                             decode_into, epoch, stage, step_kw.get("console"))
        run = end_of_step_stats_update(run, st, n)
        if prg is not None and task is not None:
            prg.advance(task, 1)
    return run, n_els, n_steps


def train(prg, console, device, dl_train, dl_val, n_batches_train, n_batches_val, model, tokenizer_encoder, tokenizer_decoder,
          tokenizer_encoder_add_special_tokens, tokenized_encoder_sentence_max_length,
          tokenizer_decoder_add_special_tokens, tokenized_decoder_sentence_max_length,
          encoder_perturb_train_pct, encoder_perturb_val_pct, decoder_perturb_train_pct, decoder_perturb_val_pct,
          n_epochs_to_decode_after, decoded_sentences, opt, lr_sched, n_epochs, vocab_size_encoder, vocab_size_decoder,
          wandb_run, run_path, export_checkpoint=True, grad_sync=None, engine=None, is_main=True):
    stats_train_best, stats_val_best = init_stats_best(), init_stats_best()
    step_kw = dict(device=device, model=model, tokenizer_encoder=tokenizer_encoder, tokenizer_decoder=tokenizer_decoder,
                   tokenizer_encoder_add_special_tokens=tokenizer_encoder_add_special_tokens,
                   tokenized_encoder_sentence_max_length=tokenized_encoder_sentence_max_length,
                   tokenizer_decoder_add_special_tokens=tokenizer_decoder_add_special_tokens,
                   tokenized_decoder_sentence_max_length=tokenized_decoder_sentence_max_length,
                   vocab_size_encoder=vocab_size_encoder, vocab_size_decoder=vocab_size_decoder, console=console)
    tasks = None
    if prg is not None:
        prg.start()
        tasks = (prg.add_task(f"[bold {COLOR_TRAIN}] Train batches", total=n_batches_train),
                 prg.add_task(f"[bold {COLOR_VAL}] Val   batches", total=n_batches_val))
    hist = []
    for epoch in range(1, n_epochs + 1):
        if prg is not None:
            prg.reset(tasks[0]); prg.reset(tasks[1])
        dec = decoded_sentences if epoch % n_epochs_to_decode_after == 0 else None
        model.train()
        t_stage = _time.perf_counter()
        run, n, s = _stage("train", dl_train, n_batches_train, step_kw, opt, lr_sched,
                           (encoder_perturb_train_pct, decoder_perturb_train_pct), dec, epoch, grad_sync, engine, prg, tasks and tasks[0])
        stats_train_run, stats_train_best = end_of_epoch_stats_update(run, stats_train_best, n, s)
        # sentences/s of THIS rank's train stage, loop and all (end_of_epoch_stats_update has just turned the device sums into
        # floats: the stage's kernels have finished).  An extra log entry, not one of the reference's keys.
        wandb_run.log({"epoch": epoch, "perf/train_s": _time.perf_counter() - t_stage, "perf/train_steps": s,
                       "perf/train_sentences_per_s": n / max(_time.perf_counter() - t_stage, 1e-9)})
        end_of_epoch_print(stats_train_run, stats_train_best, console, epoch, True, COLOR_TRAIN, STATS_EMOJI_TRAIN, False)
        wandb_run.log(create_wandb_log_dict(epoch, stats_train_run, "train"))
        if export_checkpoint and is_main:
            checkpoint(stats_train_best, model, run_path, "train")
        model.eval()
        run, n, s = _stage("val", dl_val, n_batches_val, step_kw, None, None, (encoder_perturb_val_pct, decoder_perturb_val_pct),
                           dec, epoch, None, engine, prg, tasks and tasks[1])
        stats_val_run, stats_val_best = end_of_epoch_stats_update(run, stats_val_best, n, s)
        end_of_epoch_print(stats_val_run, stats_val_best, console, epoch, False, COLOR_VAL, STATS_EMOJI_VAL, epoch != n_epochs)
        wandb_run.log(create_wandb_log_dict(epoch, stats_val_run, "val"))
        if export_checkpoint and is_main:
            checkpoint(stats_val_best, model, run_path, "val")
        hist.append((stats_train_run, stats_val_run))
    if prg is not None:
        prg.stop()
    return hist


def test(prg, console, device, dl_test, n_batches_test, model, tokenizer_encoder, tokenizer_decoder,
         tokenizer_encoder_add_special_tokens, tokenized_encoder_sentence_max_length,
         tokenizer_decoder_add_special_tokens, tokenized_decoder_sentence_max_length,
         encoder_perturb_test_pct, decoder_perturb_test_pct, decoded_sentences, vocab_size_encoder, vocab_size_decoder, epoch,
         wandb_run, engine=None):
    step_kw = dict(device=device, model=model, tokenizer_encoder=tokenizer_encoder, tokenizer_decoder=tokenizer_decoder,
                   tokenizer_encoder_add_special_tokens=tokenizer_encoder_add_special_tokens,
                   tokenized_encoder_sentence_max_length=tokenized_encoder_sentence_max_length,
                   tokenizer_decoder_add_special_tokens=tokenizer_decoder_add_special_tokens,
                   tokenized_decoder_sentence_max_length=tokenized_decoder_sentence_max_length,
                   vocab_size_encoder=vocab_size_encoder, vocab_size_decoder=vocab_size_decoder, console=console)
    model.eval()
    run, n, s = _stage("test", dl_test, n_batches_test, step_kw, None, None, (encoder_perturb_test_pct, decoder_perturb_test_pct),
                       decoded_sentences, epoch, None, engine)
    stats_test_run, stats_test_best = end_of_epoch_stats_update(run, init_stats_best(), n, s)
    end_of_epoch_print(stats_test_run, stats_test_best, console, epoch, False, COLOR_TEST, STATS_EMOJI_TEST, True)
    wandb_run.log(create_wandb_log_dict(epoch, stats_test_run, "test"))
    return stats_test_run
