"""step / train / test of the Shelgon (VQ) run -- counterpart of models/shelgon3/Trainer.py:65-462.

Kept: the function names and argument lists, the per-step stats dict and its keys, running / best bookkeeping,
checkpoint file names and dict keys (`model_state_dict`, `encoder_state_dict`, `decoder_state_dict`), the per-STEP
scheduler tick, validation under no_grad with opt=None.

Changed on purpose (hot path):
  * the loss block `kl_div(log_softmax(logits), one_hot(ids))` + `argmax(softmax(logits))` + `seq_acc`
    (Trainer.py:94-101) is one fused kernel pass over the logits (Shelgon.forward_loss -> kvq_ce_forward);
  * stats stay on the device (no .item() per step), batches are consumed lazily instead of `list(dl_train)`;
  * optional `grad_sync` (kvq.ddp.GradSync) overlaps the RCCL gradient all-reduce with backward.
Deviations from reference quirks, documented: running means are weighted by the true batch size (the reference
weights by len(batch-dict) == 2, which cancels out), `metric_acc_step` is the per-batch accuracy (the reference
stores the (batch, per-sentence) tuple by mistake), the "val" checkpoint is decided by the VAL best flags.
"""
from __future__ import annotations

import time as _time
from itertools import islice
from math import isclose

import numpy as np
import torch
from torch import Tensor, no_grad, save

from common.consts import *  # noqa: F401,F403  (colours / emoji)


def tokenize_batch(batch, tokenizer, tokenizer_add_special_tokens: bool, max_length: int, device):
    """Trainer.py:78-84.  Batches that already carry `input_ids` (pre-tokenised cache) skip the tokenizer."""
    if "input_ids" in batch:
        ids, mask = batch["input_ids"], batch["attention_mask"]
    else:
        tok = tokenizer(batch["sentence"], return_tensors="pt", padding="max_length", max_length=max_length,
                        add_special_tokens=tokenizer_add_special_tokens)
        ids, mask = tok.input_ids, tok.attention_mask
    return ids.to(device, non_blocking=True), mask.to(device, non_blocking=True)


def step(device, model, tokenizer, tokenizer_add_special_tokens: bool, opt,
         loss_recon_rescale_factor: float, loss_recon_weight: float,
         loss_vq_rescale_factor: float, loss_vq_weight: float,
         loss_perp_rescale_factor: float, loss_perp_weight: float,
         lr_sched, batch, vocab_size: int, stage: str, console=None, max_length: int = 12, grad_sync=None, engine=None):
    input_ids, attention_mask = tokenize_batch(batch, tokenizer, tokenizer_add_special_tokens, max_length, device)

    if engine is not None:
        # fast path: kvq.engine.TrainEngine runs forward, backward, gradient exchange and Adam (+ the per-step scheduler
        # tick) itself; the loss weights were given to its constructor
        out = engine.train_step(input_ids, attention_mask, prepared=batch.get("packed") if isinstance(batch, dict) else None) \
            if opt is not None else engine.eval_step(input_ids, attention_mask)
        return {
            "loss_recon_step": out["loss_recon"].detach(), "loss_vq_step": out["loss_vq"].detach(),
            "metric_perp_step": out["perplexity"].detach(), "loss_full_step": (out["loss_recon"] + out["loss_vq"]).detach(),
            "metric_acc_step": out["acc"].detach(), "padding_tokens_pct_step": -69,
        }, input_ids, out["recon_ids"]

    loss_vq_step, metric_perp_step, _indices, loss_recon_step, acc_step, recon_ids = \
        model.forward_loss(input_ids, attention_mask)

    loss_recon_step = loss_recon_step * (loss_recon_rescale_factor * loss_recon_weight)      # Trainer.py:103
    loss_vq_step = loss_vq_step * (loss_vq_rescale_factor * loss_vq_weight)                  # Trainer.py:104
    loss_full_step: Tensor = loss_recon_step + loss_vq_step                                  # Trainer.py:105

    if opt is not None:                                                                       # Trainer.py:109-115
        if grad_sync is not None:
            grad_sync.zero_grad()
        else:
            opt.zero_grad()
        loss_full_step.backward()
        if grad_sync is not None:
            grad_sync.finish()
        opt.step()
        if lr_sched is not None:
            lr_sched.step()

    return {
        "loss_recon_step": loss_recon_step.detach(),
        "loss_vq_step": loss_vq_step.detach(),
        "metric_perp_step": metric_perp_step.detach(),
        "loss_full_step": loss_full_step.detach(),
        "metric_acc_step": acc_step.detach(),
        "padding_tokens_pct_step": -69,
    }, input_ids, recon_ids


def end_of_step_stats_update(stats_stage_run: dict, stats_step: dict, n_els_batch: int):
    stats_stage_run["loss_recon_run"] += stats_step["loss_recon_step"] * n_els_batch
    stats_stage_run["loss_vq_run"] += stats_step["loss_vq_step"] * n_els_batch
    stats_stage_run["metric_perp_run"] += stats_step["metric_perp_step"] * n_els_batch
    stats_stage_run["loss_full_run"] += stats_step["loss_full_step"] * n_els_batch
    stats_stage_run["metric_acc_run"] += stats_step["metric_acc_step"] * n_els_batch * 1e2
    stats_stage_run["padding_tokens_pct_run"] += stats_step["padding_tokens_pct_step"]
    return stats_stage_run


_LOWER_IS_BETTER = {"loss_recon": True, "loss_vq": True, "metric_perp": False, "loss_full": True, "metric_acc": False}


def _sum_over_ranks(stats_stage_run: dict, n_els_epoch: int):
    """Data-parallel runs: every rank has accumulated the sums of ITS shard; one small all-reduce per stage and epoch turns
    them into sums over the whole split, so the printed / logged means and the best-checkpoint decision are the ones a single
    process on the global batch would make (the reference is single-process, Trainer.py:127-143)."""
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
        stats_stage_run[f"{key}_run"] = float(stats_stage_run[f"{key}_run"]) / max(n_els_epoch, 1)   # one sync per epoch
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

    def cell(label, key, emoji, pct=""):
        mark = emoji if stats_stage_best[f"{key}_is_best"] else "  "
        return f"{label}: [bold {stat_color}] {stats_stage_run[key + '_run']:08.6f}{pct}[/bold {stat_color}] {mark} | "

    line = head + cell("loss_recon", "loss_recon", stat_emojis[1]) + cell("loss_vq", "loss_vq", stat_emojis[0])
    if not isclose(stats_stage_run["metric_perp_run"], -69):
        line += cell("perp", "metric_perp", stat_emojis[3])
    line += cell("acc", "metric_acc", stat_emojis[2], "%")
    console.print(line + ("\n" if print_new_line else ""))


def init_stats_best():
    return {"loss_recon_best": np.inf, "loss_recon_is_best": False, "loss_vq_best": np.inf, "loss_vq_is_best": False,
            "metric_perp_best": 0, "metric_perp_is_best": False, "loss_full_best": np.inf, "loss_full_is_best": False,
            "metric_acc_best": 0, "metric_acc_is_best": False}


def init_stats_run():
    return {"loss_recon_run": 0, "loss_vq_run": 0, "metric_perp_run": 0, "loss_full_run": 0, "metric_acc_run": 0,
            "padding_tokens_pct_run": 0}


def create_wandb_log_dict(epoch: int, stats_stage_run: dict, stage: str):
    return {"epoch": epoch,
            f"{stage}/loss_recon": stats_stage_run["loss_recon_run"], f"{stage}/loss_vq": stats_stage_run["loss_vq_run"],
            f"{stage}/metric_perp": stats_stage_run["metric_perp_run"], f"{stage}/loss_full": stats_stage_run["loss_full_run"],
            f"{stage}/acc": stats_stage_run["metric_acc_run"],
            f"padding_tokens_pct/{stage}": stats_stage_run["padding_tokens_pct_run"]}


def decode_sentences(input_ids, recon_ids, tokenizer, decoded_sentences: list, epoch: int, stage: str, console=None):
    for i, r in zip(tokenizer.batch_decode(sequences=input_ids.cpu()), tokenizer.batch_decode(sequences=recon_ids.cpu())):
        decoded_sentences.append({"epoch": epoch, "stage": stage, "input_sentence": i, "recon_sentence": r})


def _save_ckpt(model, checkpoint_file_path: str, stage: str):
    save({"model_state_dict": model.state_dict(), "encoder_state_dict": model.encoder.state_dict(),
          "decoder_state_dict": model.decoder.state_dict()}, checkpoint_file_path)        # Trainer.py:240-249


def checkpoint(stats_best: dict, model, checkpoint_dir: str, stage: str):
    if stats_best["loss_recon_is_best"]:
        _save_ckpt(model, f"{checkpoint_dir}/shelgon_ckpt_loss_recon_{stage}_best.pth", stage)
    if stats_best["loss_vq_is_best"]:
        _save_ckpt(model, f"{checkpoint_dir}/shelgon_ckpt_loss_vq_{stage}_best.pth", stage)


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


def _run_stage(stage, device, loader, n_batches, model, tokenizer, tokenizer_add_special_tokens, opt, lr_sched, weights,
               vocab_size, decode_into, epoch, console, max_length, grad_sync, on_batch=None, engine=None):
    stats_run = init_stats_run()
    n_els_epoch = n_steps = 0
    for batch in islice(loader, n_batches):
        n_els_batch = _batch_size(batch)
        n_els_epoch += n_els_batch
        n_steps += 1
        ctx = torch.enable_grad() if opt is not None else no_grad()
        with ctx:
            stats_step, input_ids, recon_ids = step(
                device=device, model=model, tokenizer=tokenizer, tokenizer_add_special_tokens=tokenizer_add_special_tokens,
                opt=opt, lr_sched=lr_sched, batch=batch, vocab_size=vocab_size, stage=stage, console=console,
                max_length=max_length, grad_sync=grad_sync, engine=engine, **weights)
        if decode_into is not None:
            decode_sentences(input_ids, recon_ids, tokenizer, decode_into, epoch, stage, console)
        stats_run = end_of_step_stats_update(stats_run, stats_step, n_els_batch)
        if on_batch:
            on_batch()
    return stats_run, n_els_epoch, n_steps


def train(prg, console, device, dl_train, dl_val, n_batches_train: int, n_batches_val: int, model, tokenizer,
          tokenizer_add_special_tokens: bool, n_epochs_to_decode_after: int, decoded_sentences: list, opt,
          loss_recon_rescale_factor: float, loss_recon_weight: float, loss_vq_rescale_factor: float, loss_vq_weight: float,
          loss_perp_rescale_factor: float, loss_perp_weight: float, lr_sched, n_epochs: int, vocab_size: int,
          wandb_run, run_path: str, export_checkpoint: bool, max_length: int = 12, grad_sync=None, is_main: bool = True,
          engine=None):
    if not export_checkpoint and console is not None:
        console.print(f"[bold {COLOR_WARNING}]Warning[/bold {COLOR_WARNING}] checkpoint exporting is [bold {COLOR_OFF}]OFF[/bold {COLOR_OFF}]!\n")
    weights = dict(loss_recon_rescale_factor=loss_recon_rescale_factor, loss_recon_weight=loss_recon_weight,
                   loss_vq_rescale_factor=loss_vq_rescale_factor, loss_vq_weight=loss_vq_weight,
                   loss_perp_rescale_factor=loss_perp_rescale_factor, loss_perp_weight=loss_perp_weight)
    tasks = None
    if prg is not None:
        prg.start()
        tasks = (prg.add_task(f"[bold {COLOR_EPOCH}] Epochs", total=n_epochs),
                 prg.add_task(f"[bold {COLOR_TRAIN}] Train batches", total=n_batches_train),
                 prg.add_task(f"[bold {COLOR_VAL}] Val   batches", total=n_batches_val))
    stats_train_best, stats_val_best = init_stats_best(), init_stats_best()
    history = []
    for epoch in range(1, n_epochs + 1):
        if tasks:
            prg.reset(tasks[1]); prg.reset(tasks[2])
        decode_now = decoded_sentences if epoch % n_epochs_to_decode_after == 0 else None

        model.train()
        tick = (lambda: (prg.advance(tasks[1], 1), prg.advance(tasks[0], 1 / (n_batches_train + n_batches_val)))) if tasks else None
        t_stage = _time.perf_counter()
        run, n_els, n_steps = _run_stage("train", device, dl_train, n_batches_train, model, tokenizer, tokenizer_add_special_tokens,
                                         opt, lr_sched, weights, vocab_size, decode_now, epoch, console, max_length, grad_sync, tick,
                                         engine=engine)
        stats_train_run, stats_train_best = end_of_epoch_stats_update(run, stats_train_best, n_els, n_steps)
        # sentences/s of THIS rank's train stage, loop and all (end_of_epoch_stats_update has just turned the device sums into
        # floats: the stage's kernels have finished).  An extra log entry, not one of the reference's keys.
        wandb_run.log({"epoch": epoch, "perf/train_s": _time.perf_counter() - t_stage, "perf/train_steps": n_steps,
                       "perf/train_sentences_per_s": n_els / max(_time.perf_counter() - t_stage, 1e-9)})
        end_of_epoch_print(stats_train_run, stats_train_best, console, epoch, True, COLOR_TRAIN, STATS_EMOJI_TRAIN, False)
        wandb_run.log(create_wandb_log_dict(epoch, stats_train_run, "train"))

        model.eval()
        tick = (lambda: (prg.advance(tasks[2], 1), prg.advance(tasks[0], 1 / (n_batches_train + n_batches_val)))) if tasks else None
        run, n_els, n_steps = _run_stage("val", device, dl_val, n_batches_val, model, tokenizer, tokenizer_add_special_tokens,
                                         None, None, weights, vocab_size, decode_now, epoch, console, max_length, None, tick,
                                         engine=engine)
        stats_val_run, stats_val_best = end_of_epoch_stats_update(run, stats_val_best, n_els, n_steps)
        end_of_epoch_print(stats_val_run, stats_val_best, console, epoch, False, COLOR_VAL, STATS_EMOJI_VAL, epoch != n_epochs)
        wandb_run.log(create_wandb_log_dict(epoch, stats_val_run, "val"))
        if export_checkpoint and is_main:
            checkpoint(stats_val_best, model, run_path, "val")
        history.append((dict(stats_train_run), dict(stats_val_run)))
    return history


def test(prg, console, device, dl_test, n_batches_test, model, tokenizer, tokenizer_add_special_tokens: bool,
         loss_recon_rescale_factor: float, loss_recon_weight: float, loss_vq_rescale_factor: float, loss_vq_weight: float,
         loss_perp_rescale_factor: float, loss_perp_weight: float, decoded_sentences: list, vocab_size: int, epoch: int,
         wandb_run, max_length: int = 12, engine=None):
    weights = dict(loss_recon_rescale_factor=loss_recon_rescale_factor, loss_recon_weight=loss_recon_weight,
                   loss_vq_rescale_factor=loss_vq_rescale_factor, loss_vq_weight=loss_vq_weight,
                   loss_perp_rescale_factor=loss_perp_rescale_factor, loss_perp_weight=loss_perp_weight)
    task = prg.add_task(f"[bold {COLOR_TEST}] Test  batches", total=n_batches_test) if prg is not None else None
    model.eval()
    run, n_els, n_steps = _run_stage("test", device, dl_test, n_batches_test, model, tokenizer, tokenizer_add_special_tokens,
                                     None, None, weights, vocab_size, decoded_sentences, epoch, console, max_length, None,
                                     (lambda: prg.advance(task, 1)) if task is not None else None, engine=engine)
    stats_test_run, stats_test_best = end_of_epoch_stats_update(run, init_stats_best(), n_els, n_steps)
    end_of_epoch_print(stats_test_run, stats_test_best, console, epoch, False, COLOR_TEST, STATS_EMOJI_TEST, True)
    wandb_run.log(create_wandb_log_dict(epoch, stats_test_run, "test"))
    return stats_test_run
