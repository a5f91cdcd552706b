"""MultiVectorQuantizer -- G independent codebooks, one per slice of the encoder output (extension; BASELINE.json configs[4]).

Not in the reference: each token's e_dim-vector is cut into n_factors contiguous slices and every
slice is quantised by its own K-entry codebook with the arithmetic of VectorQuantizer.forward
(models/shelgon3/VectorQuantizer.py:55-93) -- i.e. indices and z_q equal n_factors calls of that forward on the slices.  All
factors run as ONE grouped launch of kvq_vq_forward / kvq_vq_backward (G = n_factors).  When n_factors does not divide e_dim
(9 factors on bert-base's 768) the slices are 86 / 85 columns wide and are zero-padded to a common width (a multiple of 32: 96);
zero columns add nothing to a distance, the codebooks' pad columns start at zero and receive zero gradient.
Composition of the per-factor scalars: loss = (1 + beta) * MSE over all N * e_dim elements (the single-codebook
normalisation; = mean_g loss_g for equal slices), perplexity = mean_g.
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
        if not 1 <= n_factors <= e_dim:
            raise ValueError(f"n_factors={n_factors} must be between 1 and e_dim={e_dim}")
        self.n_factors, self.n_e, self.e_dim, self.beta = n_factors, n_e, e_dim, beta
        base, rem = divmod(e_dim, n_factors)
        widths = [base + (1 if g < rem else 0) for g in range(n_factors)]
        self.ragged = rem != 0
        self.d_factor = base if not self.ragged else (max(widths) + 31) // 32 * 32           # common (padded) slice width
        self.loss_weight = self.d_factor / e_dim                                              # loss = loss_weight * sum_g loss_g
        col = torch.full((n_factors, self.d_factor), e_dim, dtype=torch.long)                 # e_dim = the appended zero column
        inv = torch.empty(e_dim, dtype=torch.long)
        o = 0
        for g, w in enumerate(widths):
            col[g, :w] = torch.arange(o, o + w)
            inv[o:o + w] = g * self.d_factor + torch.arange(w)
            o += w
        self.register_buffer("col_map", col.view(-1), persistent=False)
        self.register_buffer("inv_map", inv, persistent=False)
        self.embedding = nn.Embedding(n_factors * n_e, self.d_factor)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)                 # VectorQuantizer.py:29 per codebook
        if self.ragged:
            self.embedding.weight.data.view(n_factors, n_e, self.d_factor)[col.view(n_factors, 1, -1).expand(-1, n_e, -1) == e_dim] = 0.0
        self.ema_decay, self.ema_eps = ema_decay, ema_eps
        if ema_decay is not None:
            self.embedding.weight.requires_grad_(False)
            self.register_buffer("ema_n", torch.ones(n_factors, n_e))
            self.register_buffer("ema_m", self.embedding.weight.data.clone().view(n_factors, n_e, self.d_factor))
        self.last_code_counts = None

    def codebooks(self):
        return self.embedding.weight.view(self.n_factors, self.n_e, self.d_factor)

    def split(self, z2d):
        """[N, e_dim] -> [G, N, d_factor] (contiguous: the grouped kernels take one matrix per factor)."""
        N = z2d.shape[0]
        if self.ragged:
            z2d = torch.nn.functional.pad(z2d, (0, 1)).index_select(1, self.col_map)
        return z2d.reshape(N, self.n_factors, self.d_factor).permute(1, 0, 2).contiguous()

    def merge(self, zg):
        """[G, N, d_factor] -> [N, e_dim]"""
        flat = zg.permute(1, 0, 2).reshape(zg.shape[1], self.n_factors * self.d_factor)
        return flat.index_select(1, self.inv_map) if self.ragged else flat

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
        return loss.sum() * self.loss_weight, z_q, perp.mean(), None, idx.t().reshape(B, S, self.n_factors)

    def ema_update(self, zg, idx):
        ema_codebook_update(zg, idx, self.ema_n, self.ema_m, self.codebooks().data, float(self.ema_decay), float(self.ema_eps))
        self.codebook_epoch = getattr(self, "codebook_epoch", 0) + 1          # see VectorQuantizer.ema_update
