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

from kvq.functional import vector_quantize, vq_one_hot


class VectorQuantizer(nn.Module):
    """Discretization bottleneck of the VQ-VAE.

    n_e: number of embeddings; e_dim: embedding dimension; beta: weight of the codebook term as written in the
    reference's loss (VectorQuantizer.py:76-77: `mean((sg[z_q]-z)^2) + beta*mean((z_q-sg[z])^2)`).
    """

    def __init__(self, n_e, e_dim, beta, vq_codebook_init_values: Tensor = None):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        if vq_codebook_init_values is not None:                      # reference :26-27
            self.embedding.weight.data.copy_(vq_codebook_init_values)
        else:                                                         # reference :29
            self.embedding.weight.data.uniform_(-1.0 / self.n_e, 1.0 / self.n_e)
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
        z_q = z_q.view(z.shape)
        min_encodings = vq_one_hot(idx, self.n_e) if self.materialize_min_encodings else None
        min_encoding_indices = idx.reshape(batch_size, seq_len, 1)
        return loss, z_q, perplexity, min_encodings, min_encoding_indices
