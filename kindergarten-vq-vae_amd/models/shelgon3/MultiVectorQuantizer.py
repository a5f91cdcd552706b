"""MultiVectorQuantizer -- G independent codebooks, one per slice of the encoder output (extension; BASELINE.json configs[4]).

Not in the reference: each token's e_dim-vector is cut into n_factors slices of e_dim / n_factors
columns and every slice is quantised by its own K-entry codebook with the arithmetic of VectorQuantizer.forward
(models/shelgon3/VectorQuantizer.py:55-93) -- i.e. the result equals n_factors calls of that forward on the slices.  All
factors run as ONE grouped launch of kvq_vq_forward / kvq_vq_backward (G = n_factors).  Composition of the per-factor scalars:
loss = mean_g loss_g (= (1 + beta) * MSE over all N * e_dim elements, the single-codebook normalisation), perplexity = mean_g.
Surface kept from VectorQuantizer: `embedding` parameter (rows g*n_e .. (g+1)*n_e are factor g's codebook), `forward(z, device)`
5-tuple with indices [B, S, n_factors]; `ema_decay` as in VectorQuantizer.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from kvq.functional import vector_quantize
from models.shelgon3.VectorQuantizer import ema_codebook_update


class MultiVectorQuantizer(nn.Module):
    def __init__(self, n_factors, n_e, e_dim, beta, ema_decay: float = None, ema_eps: float = 1e-5):
        super().__init__()
        if e_dim % n_factors != 0:
            raise ValueError(f"e_dim={e_dim} must be a multiple of n_factors={n_factors}")
        self.n_factors, self.n_e, self.e_dim, self.beta = n_factors, n_e, e_dim, beta
        self.d_factor = e_dim // n_factors
        self.embedding = nn.Embedding(n_factors * n_e, self.d_factor)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)                 # VectorQuantizer.py:29 per codebook
        self.ema_decay, self.ema_eps = ema_decay, ema_eps
        if ema_decay is not None:
            self.embedding.weight.requires_grad_(False)
            self.register_buffer("ema_n", torch.ones(n_factors, n_e))
            self.register_buffer("ema_m", self.embedding.weight.data.clone().view(n_factors, n_e, self.d_factor))
        self.last_code_counts = None

    def codebooks(self):
        return self.embedding.weight.view(self.n_factors, self.n_e, self.d_factor)

    def split(self, z2d):
        """[N, e_dim] -> [G, N, e_dim / G] (contiguous: the grouped kernels take one matrix per factor)."""
        N = z2d.shape[0]
        return z2d.view(N, self.n_factors, self.d_factor).permute(1, 0, 2).contiguous()

    def merge(self, zg):
        """[G, N, e_dim / G] -> [N, e_dim]"""
        return zg.permute(1, 0, 2).reshape(zg.shape[1], self.e_dim)

    def forward(self, z: torch.Tensor, device=None):
        if z.dim() != 3 or z.shape[-1] != self.e_dim or not z.is_contiguous():
            raise RuntimeError(f"MultiVectorQuantizer expects a contiguous (B, S, {self.e_dim}) tensor, got {tuple(z.shape)}")
        B, S, _ = z.shape
        zg = self.split(z.view(-1, self.e_dim))
        E = self.codebooks()
        if E.dtype != torch.float32:
            E = E.float()
        loss, zq, perp, idx, counts = vector_quantize(zg, E.contiguous(), self.beta)
        self.last_code_counts = counts
        if self.ema_decay is not None and self.training:
            self.ema_update(zg.detach(), idx)
        z_q = self.merge(zq).view(B, S, self.e_dim)
        return loss.mean(), z_q, perp.mean(), None, idx.t().reshape(B, S, self.n_factors)

    def ema_update(self, zg, idx):
        ema_codebook_update(zg, idx, self.ema_n, self.ema_m, self.codebooks().data, float(self.ema_decay), float(self.ema_eps))
