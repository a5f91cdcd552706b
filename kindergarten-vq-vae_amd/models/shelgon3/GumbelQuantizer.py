"""GumbelQuantizer -- the reference's second VQ_MODE on libkvq.so (gfx950); mirrors models/shelgon3/GumbelQuantizer.py:16-83.

Same class name (Shelgon.forward dispatches on it, Shelgon.py:60), constructor arguments, parameters (`proj` = Conv1d(enc_out_size,
n_embed, 1), `embed` = Embedding(n_embed, embedding_dim): state-dict keys proj.weight [K,H,1], proj.bias, embed.weight) and the
(z_q, diff, ind) return.  The 1x1 convolution and the einsum are the two GEMMs they really are; everything between them
(Gumbel noise, softmax at temperature tau, straight-through one-hot, arg-max, KL to the uniform prior) is one row kernel
(kvq_gumbel_forward) instead of ~15 ATen ops over [B, K, S] tensors.  The noise comes from the library's Philox stream, not from
torch's generator: samples differ from the reference's by construction; with the noise passed in explicitly the results are
checked against the reference module (tests/golden/gumbel_*.npz).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from kvq.functional import gumbel_quantize


class GumbelQuantizer(nn.Module):
    """Gumbel-softmax quantiser (Jang et al. 2016), text-sequence variant of the reference."""

    def __init__(self, enc_out_size, n_embed, embedding_dim, temperature: float, kl_div_scale: float, straight_through: bool):
        super().__init__()
        self.e_dim = embedding_dim
        self.n_embed = n_embed
        self.straight_through = straight_through
        self.temperature = temperature
        self.kld_scale = kl_div_scale
        self.proj = nn.Conv1d(enc_out_size, n_embed, 1)
        self.embed = nn.Embedding(n_embed, embedding_dim)
        self.seed = 0x9e3779b9          # base of the Philox stream; every forward call advances `calls`
        self.calls = 0

    def forward(self, z: torch.Tensor, is_training: bool, noise: torch.Tensor = None):
        """z (batch, seq_len, enc_out_size) -> (z_q (batch, seq_len, embedding_dim), diff (0-d), ind (batch, seq_len) int64).

        `noise` ((batch*seq_len, n_embed) float32 Gumbel(0,1) samples) overrides the internal generator (parity tests)."""
        if z.dim() != 3:
            raise ValueError(f"z must be (batch, seq_len, enc_out_size), got {tuple(z.shape)}")
        B, S, H = z.shape
        hard = self.straight_through if is_training else True                    # reference :54
        w = self.proj.weight.squeeze(-1)                                         # [K, H]: a 1x1 Conv1d is a linear layer
        logits = F.linear(z.reshape(B * S, H), w.to(z.dtype), self.proj.bias.to(z.dtype))
        self.calls += 1
        y, diff, ind = gumbel_quantize(logits, self.temperature, hard, self.kld_scale, noise=noise, seed=self.seed, site=self.calls)
        z_q = (y @ self.embed.weight.to(y.dtype)).view(B, S, self.e_dim)         # einsum('b n s, n d -> b d s') transposed back
        return z_q, diff, ind.view(B, S)
