"""TEST INFRASTRUCTURE -- CPU (f32, PyTorch ATen) restatement of one training step of the reference's VQ run.

Used only by tests/ and by bench.py's cpu_baseline leg ("port": timed on the GPU box's host cores).
Follows, expression for expression:
    models/shelgon3/Shelgon.py:50-73   encoder -> VectorQuantizer -> decoder(encoder_hidden_states=z_q).logits
    models/shelgon3/Trainer.py:82-115  pad to max_length, kl_div(log_softmax, one_hot) "batchmean", argmax(softmax),
                                       seq_acc, weighted sum, zero_grad / backward / Adam step / scheduler tick
    models/shelgon3/main.py:91         Adam over ALL model.parameters()
    models/bagon/Bagon.py:40-55 + models/bagon/Trainer.py:96-130   the plain Bagon step (OracleBagon / bagon_step)
The BERT blocks are HuggingFace's own classes (third-party part of the reference, transformers 5.15 in this image)
built from a local BertConfig: encoder = BertModel, decoder = BertLMHeadModel(is_decoder, add_cross_attention)
(what EncoderDecoderModel.from_encoder_decoder_pretrained builds, Bagon.py:24-31).
Parity status: the VQ part is pinned by tests/golden/vq_*.npz (the reference's own module); the compositions are pinned by
tests/golden/step_tiny.npz and step_bagon_tiny.npz, which tests/golden/make_step*_golden.py produce from the pieces of the
reference that run here (its VectorQuantizer and seq_acc, imported; HuggingFace's BERT classes) wired as the reference wires
them -- the reference's Shelgon / Bagon / Trainer classes themselves do not import as checked in.
"""
from __future__ import annotations

import time

import torch
import torch.nn as nn
from torch.nn.functional import kl_div, log_softmax, one_hot, softmax

from .vq_oracle import torch_expr_forward


class OracleVQ(nn.Module):
    """VectorQuantizer.py:19-29 parameters + :31-93 forward (via torch_expr_forward)."""

    def __init__(self, n_e, e_dim, beta, init=None):
        super().__init__()
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.embedding = nn.Embedding(n_e, e_dim)
        if init is not None:
            self.embedding.weight.data.copy_(init)
        else:
            self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)

    def forward(self, z, device=None):
        return torch_expr_forward(z, self.embedding.weight, self.beta)


class OracleShelgon(nn.Module):
    def __init__(self, bert_cfg: dict, n_e=512, e_dim=768, beta=0.25, codebook_init=None):
        super().__init__()
        from transformers import BertConfig, BertLMHeadModel, BertModel
        self.encoder = BertModel(BertConfig(**bert_cfg))
        self.decoder = BertLMHeadModel(BertConfig(**bert_cfg, is_decoder=True, add_cross_attention=True))
        self.vector_quantizer = OracleVQ(n_e, e_dim, beta, codebook_init)

    def forward(self, input_ids, attention_mask):
        embeds = self.encoder(input_ids, attention_mask=attention_mask).last_hidden_state          # Shelgon.py:52
        vq_loss, z_q, perplexity, _enc, idx = self.vector_quantizer(embeds)                          # Shelgon.py:58
        logits = self.decoder(encoder_hidden_states=z_q, input_ids=input_ids, attention_mask=attention_mask).logits  # :71
        return vq_loss, perplexity, idx, logits


class OracleBagon(nn.Module):
    """models/bagon/Bagon.py:24-31 (what from_encoder_decoder_pretrained builds, from a local config) + :40-55 forward."""

    def __init__(self, bert_cfg: dict):
        super().__init__()
        from transformers import BertConfig, BertLMHeadModel, BertModel
        self.encoder = BertModel(BertConfig(**bert_cfg))
        self.decoder = BertLMHeadModel(BertConfig(**bert_cfg, is_decoder=True, add_cross_attention=True))

    def forward(self, encoder_input_ids, encoder_attention_mask, decoder_input_ids, decoder_attention_mask):
        encoder_output = self.encoder(encoder_input_ids, attention_mask=encoder_attention_mask).last_hidden_state      # Bagon.py:46-48
        return self.decoder(encoder_hidden_states=encoder_output, input_ids=decoder_input_ids,
                            attention_mask=decoder_attention_mask).logits                                             # Bagon.py:50-53


def bagon_step(model, opt, ids_enc, mask_enc, ids_dec, mask_dec, vocab_size, lr_sched=None):
    """models/bagon/Trainer.py:96-130 (tokenisation and perturbation happen upstream: both id sets are given).  The target of
    the loss and of the accuracies is the decoder's input as given (i.e. after its perturbation, Trainer.py:94,103)."""
    logits = model(ids_enc, mask_enc, ids_dec, mask_dec)
    loss_recon = kl_div(input=log_softmax(logits.reshape(-1, vocab_size), dim=-1),
                        target=one_hot(ids_dec, vocab_size).reshape(-1, vocab_size).float(), reduction="batchmean")
    recon_ids = torch.argmax(softmax(logits, dim=-1), dim=-1)
    hits = (recon_ids - ids_dec) == 0                                                                  # common/metrics.py:18-22
    acc_batch, acc_sentence = hits.sum() / recon_ids.numel(), torch.mean(hits.float(), dim=-1)         # common/metrics.py:30-34
    if opt is not None:
        opt.zero_grad()
        loss_recon.backward()
        opt.step()
        if lr_sched is not None:
            lr_sched.step()
    return dict(loss_recon=loss_recon.detach(), loss_full=loss_recon.detach(), acc_batch=acc_batch, acc_sentence=acc_sentence,
                recon_ids=recon_ids, logits=logits.detach())


def step(model, opt, input_ids, attention_mask, vocab_size, w_recon=1.0, w_vq=1.0, lr_sched=None):
    """Trainer.py:87-124 (tokenisation happens upstream: ids are given)."""
    loss_vq, perp, idx, logits = model(input_ids, attention_mask)
    loss_recon = kl_div(input=log_softmax(logits.reshape(-1, vocab_size), dim=-1),
                        target=one_hot(input_ids, vocab_size).reshape(-1, vocab_size).float(), reduction="batchmean")
    recon_ids = torch.argmax(softmax(logits, dim=-1), dim=-1)
    acc = (recon_ids == input_ids).sum() / input_ids.numel()                                          # common/metrics.py:25-30
    loss_recon = loss_recon * w_recon
    loss_vq = loss_vq * w_vq
    loss_full = loss_recon + loss_vq
    if opt is not None:
        opt.zero_grad()
        loss_full.backward()
        opt.step()
        if lr_sched is not None:
            lr_sched.step()
    return dict(loss_recon=loss_recon.detach(), loss_vq=loss_vq.detach(), perplexity=perp.detach(), loss_full=loss_full.detach(),
                acc=acc, idx=idx, recon_ids=recon_ids)


def host_cores() -> int:
    """CPU cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (a GPU box hands a
    container a share of a much larger host; os.cpu_count() would oversubscribe it by an order of magnitude)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def time_cpu_steps(bert_cfg: dict, batch: int, seq_len: int, n_e: int, e_dim: int, beta: float, vocab_size: int,
                   warmup: int, steps: int, seed: int = 0, threads: int | None = None, budget_s: float = 30.0, log=None):
    """CPU baseline for bench.py: sentences/s of the restated step on this host's cores (f32, train mode, Adam on all params)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kindergarten-vq-vae_amd"))
    from dsentences.synthetic import random_token_batch
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    model = OracleShelgon(bert_cfg, n_e, e_dim, beta).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    gen = torch.Generator().manual_seed(69)
    times = []
    t_begin = time.perf_counter()
    for i in range(warmup + steps):
        ids, mask = random_token_batch(batch, seq_len, gen)
        t0 = time.perf_counter()
        step(model, opt, ids, mask, vocab_size)
        dt = time.perf_counter() - t0
        if log:
            log(f"[cpu_baseline] step {i} ({'warm-up' if i < warmup else 'timed'}): {dt:.2f} s on {torch.get_num_threads()} threads")
        if i >= warmup:
            times.append(dt)
        if len(times) >= 2 and time.perf_counter() - t_begin > budget_s:     # bounded sample
            break
    times.sort()
    med = times[len(times) // 2]
    return dict(sentences_per_s=batch / med, s_per_step=med, threads=torch.get_num_threads(), steps=len(times), batch=batch)
