"""VectorQuantizer -- drop-in for the reference's codebook bottleneck, running on libkvq.so (gfx950).

Mirrors models/shelgon3/VectorQuantizer.py:8-93 of the reference: same class name (Shelgon.forward dispatches
on it, Shelgon.py:57), same constructor, same `embedding` parameter (state-dict key `embedding.weight`), same
forward signature and 5-tuple.  What differs is where the arithmetic happens: one fused HIP kernel
(kvq_vq_forward) instead of ~20 ATen ops, no [N,K] distance / one-hot temporaries, exact-f32 MFMA distances.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from kvq.functional import vector_quantize, vq_ema_update, vq_one_hot


def ema_codebook_update(z, idx, ema_n, ema_m, E, decay, eps):
    """EMA codebook update (extension named by BASELINE.json, absent from the reference, default off): kvq_vq_ema_update on
    the local tokens, then -- data parallel -- the running statistics are averaged over the ranks (the update is linear in
    the per-batch counts / sums, so this equals the update from the global batch's mean statistics) and the codebook is
    rebuilt from them with the kernel's formula (include/kvq.h).  z [N,D] or [G,N,D]; idx [N] or [G,N]."""
    import torch.distributed as dist
    with torch.no_grad():
        vq_ema_update(z, idx, ema_n, ema_m, E, decay, eps)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for t in (ema_n, ema_m):
                dist.all_reduce(t)
                t.div_(dist.get_world_size())
            K = ema_n.shape[-1]
            tot = ema_n.sum(-1, keepdim=True)
            n = (ema_n + eps) / (tot + K * eps) * tot
            E.copy_(ema_m / n.unsqueeze(-1))


class VectorQuantizer(nn.Module):
    """Discretization bottleneck of the VQ-VAE.

    n_e: number of embeddings; e_dim: embedding dimension; beta: weight of the codebook term as written in the
    reference's loss (VectorQuantizer.py:76-77: `mean((sg[z_q]-z)^2) + beta*mean((z_q-sg[z])^2)`).
    """

    def __init__(self, n_e, e_dim, beta, vq_codebook_init_values: Tensor = None, ema_decay: float = None, ema_eps: float = 1e-5):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        if vq_codebook_init_values is not None:                      # reference :26-27
            self.embedding.weight.data.copy_(vq_codebook_init_values)
        else:                                                         # reference :29
            self.embedding.weight.data.uniform_(-1.0 / self.n_e, 1.0 / self.n_e)
        # Extension (not in the reference, off unless asked for): the codebook follows an exponential moving average of the
        # encoder outputs assigned to each code instead of the gradient of the loss term at reference :76-77.
        self.ema_decay, self.ema_eps = ema_decay, ema_eps
        if ema_decay is not None:
            self.embedding.weight.requires_grad_(False)
            self.register_buffer("ema_n", torch.ones(n_e))
            self.register_buffer("ema_m", self.embedding.weight.data.clone())
        # The reference always builds min_encodings [N,K]; its only caller drops it (Shelgon.py:58).  Keep the
        # contract by default, let the training path switch the 4*N*K-byte write off.
        self.materialize_min_encodings = True
        self.last_code_counts = None   # [K] usage histogram of the last forward (free by-product of the kernel)

    def forward(self, z: torch.Tensor, device=None):
        """z: (batch, seq_len, e_dim) encoder output -> (loss, z_q, perplexity, min_encodings, min_encoding_indices).

        `device` is accepted for signature compatibility (reference :31) and ignored: outputs live on z's device.
        """
        if z.dim() != 3:
            raise ValueError(f"not enough values to unpack: z must be (batch, seq_len, e_dim), got {tuple(z.shape)}")
        batch_size, seq_len, _ = z.shape
        if z.shape[-1] != self.e_dim or not z.is_contiguous():
            # the reference's z.view((-1, e_dim)) raises in exactly these two situations (:55)
            raise RuntimeError(f"shape '[-1, {self.e_dim}]' is invalid for input of size {z.numel()} "
                               f"or z is not contiguous (VectorQuantizer expects a contiguous (B,S,{self.e_dim}) tensor)")
        weight = self.embedding.weight
        if weight.dtype != torch.float32:
            weight = weight.float()
        loss, z_q, perplexity, idx, counts = vector_quantize(z.view(-1, self.e_dim), weight, self.beta)
        self.last_code_counts = counts
        if self.ema_decay is not None and self.training:
            self.ema_update(z.detach().view(-1, self.e_dim), idx)
        z_q = z_q.view(z.shape)
        min_encodings = vq_one_hot(idx, self.n_e) if self.materialize_min_encodings else None
        min_encoding_indices = idx.reshape(batch_size, seq_len, 1)
        return loss, z_q, perplexity, min_encodings, min_encoding_indices

    def ema_update(self, z2d, idx):
        """One EMA step of the codebook from the tokens z2d [N, e_dim] and their codes idx [N] (training steps only)."""
        ema_codebook_update(z2d, idx, self.ema_n, self.ema_m, self.embedding.weight.data, float(self.ema_decay), float(self.ema_eps))
        # the kernel writes the codebook through a raw pointer (no tensor-version bump): tell whoever caches a derived copy
        # (the TrainEngine's fragment-ordered pack) that it is stale
        self.codebook_epoch = getattr(self, "codebook_epoch", 0) + 1
