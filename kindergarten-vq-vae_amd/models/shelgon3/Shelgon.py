"""Shelgon -- Bagon with a vector-quantised bottleneck between encoder and decoder.

Surface kept from the reference (models/shelgon3/Shelgon.py:26-179): Shelgon(Bagon) with the `vector_quantizer`
attribute, `forward(input_ids, attention_mask, device, is_training) -> (vq_loss, perplexity, indices, logits)`,
dispatch on the quantizer's class name, the e_dim assertion, `from_pretrained_bagon` checkpoint loading
(keys encoder_state_dict / decoder_state_dict) and the freeze modes.

`forward_loss` is the fused training path: it stops at the 768-d head states and lets kvq_ce_forward produce the
reconstruction loss, accuracy and recon ids straight from the vocabulary projection, so neither the [N,V]
log-softmax nor the reference's 1 GB one-hot target (Trainer.py:94-98) exist.
"""
from __future__ import annotations

from typing import Union

import torch

from common.consts import COLOR_FROZEN, COLOR_TOT, COLOR_TRAIN, SUPPORTED_VQ_MODES  # noqa: F401
from kvq import bert as kbert
from kvq.functional import fused_cross_entropy
from models.bagon.Bagon import Bagon
from models.shelgon3.VectorQuantizer import VectorQuantizer

SUPPORTED_MODEL_MODES = ["full", "dec-head-ft", "enc-head-ft-dec-head-ft", "vq-ft"]


class Shelgon(Bagon):
    def __init__(self, encoder_model_name: str, vector_quantizer: VectorQuantizer, decoder_model_name: str,
                 from_pretrained_bagon: Union[str, None] = None, cross_attn_make_trainable: bool = False,
                 compute_dtype: torch.dtype = torch.bfloat16, backend: str = "kvq"):
        super().__init__(encoder_model_name=encoder_model_name, decoder_model_name=decoder_model_name,
                         cross_attn_make_trainable=cross_attn_make_trainable, compute_dtype=compute_dtype, backend=backend)
        self.vector_quantizer = vector_quantizer
        if from_pretrained_bagon is not None:                                     # Shelgon.py:41-45
            ckpt = torch.load(from_pretrained_bagon, map_location="cpu")
            self.encoder.load_state_dict(ckpt["encoder_state_dict"])
            self.decoder.load_state_dict(ckpt["decoder_state_dict"])
        self.model_mode = "full"

    def _quantize(self, embeds, device):
        assert embeds.shape[-1] == self.vector_quantizer.e_dim, \
            "embedding dim of encoder output must match e_dim (for now)!"          # Shelgon.py:54
        kind = type(self.vector_quantizer).__name__                                # Shelgon.py:57
        if kind == "VectorQuantizer":
            vq_loss, z_q, perplexity, _enc, indices = self.vector_quantizer.forward(embeds.contiguous(), device)
            return vq_loss, z_q, perplexity, indices
        if kind == "MultiVectorQuantizer":                                         # extension: G codebooks, one grouped launch
            vq_loss, z_q, perplexity, _enc, indices = self.vector_quantizer.forward(embeds.contiguous(), device)
            return vq_loss, z_q, perplexity, indices
        if kind == "GumbelQuantizer":                                              # Shelgon.py:60-65
            z_q, vq_loss, indices = self.vector_quantizer.forward(embeds, self.training)
            # "not the actual perplexity computation, but still informative" (Shelgon.py:63): number of codes in use,
            # counted on the device instead of through a .cpu() copy
            perplexity = torch.unique(indices).numel()
            return vq_loss, z_q.to(embeds.dtype), torch.tensor(float(perplexity), device=embeds.device), indices
        raise ValueError(f"{kind} vector quantizer mode NOT supported. Supported modalities: VectorQuantizer, GumbelQuantizer, MultiVectorQuantizer")

    def forward(self, input_ids, attention_mask, device=None, is_training: bool = True):
        if self._engine_forward_ok(input_ids):                                     # no autograd graph wanted: HIP end to end
            from kvq.engine import engine_of
            out = engine_of(self).forward_logits(input_ids, attention_mask, training=self.training, quantizer_training=is_training)
            return out["loss_vq_raw"], out["perplexity"], out["indices"], out["logits"]
        if self._engine_autograd_ok(input_ids):                                    # autograd wanted AND opted in: still the engine
            from kvq.engine import engine_autograd_forward
            logits, vq_loss, perplexity, indices = engine_autograd_forward(self, input_ids, attention_mask, q_training=is_training)
            return vq_loss, perplexity, indices, logits
        embeds = self.encode(input_ids, attention_mask)                            # Shelgon.py:52
        vq_loss, z_q, perplexity, indices = self._quantize(embeds, device)
        logits = self.decode(z_q, input_ids, attention_mask)                       # Shelgon.py:71
        return vq_loss, perplexity, indices, logits

    def code_indices(self, input_ids, attention_mask, device=None):
        """min_encoding_indices of a batch without the decoder: what the index consumers use of forward()
        (analyses/unsupervised_vq_disentanglement/unsupervised_vq_disentanglement.py:164 keeps only that element)."""
        if self._engine_forward_ok(input_ids):
            from kvq.engine import engine_of
            return engine_of(self).code_indices(input_ids, attention_mask)["indices"]
        with torch.no_grad():
            return self._quantize(self.encode(input_ids, attention_mask), device)[3]

    def forward_loss(self, input_ids, attention_mask):
        """Fused step body: (vq_loss, perplexity, indices, loss_recon, acc_per_batch, recon_ids)."""
        embeds = self.encode(input_ids, attention_mask)
        vq_loss, z_q, perplexity, indices = self._quantize(embeds, None)
        hidden = self.decode_hidden(z_q, input_ids, attention_mask)
        logits = kbert.lm_head_logits(self.decoder, hidden, self.compute_dtype)
        loss_recon, acc, recon_ids = fused_cross_entropy(logits, input_ids, inplace_backward=True)
        return vq_loss, perplexity, indices, loss_recon, acc, recon_ids

    def _summary_parts(self):
        return [("encoder", "Encoder", self.encoder), ("vector_quantizer", "Vector Quantizer", self.vector_quantizer),
                ("decoder", "Decoder", self.decoder)]
