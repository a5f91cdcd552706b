"""Bagon -- BERT encoder -> BERT decoder (cross-attending) sentence autoencoder, MI355X execution.

Surface kept from the reference (models/bagon/Bagon.py:15-179): constructor arguments, `.encoder` (BertModel) /
`.decoder` (BertLMHeadModel with is_decoder + cross-attention) attributes and their state-dict keys,
`forward(encoder_input_ids, encoder_attention_mask, decoder_input_ids, decoder_attention_mask) -> logits`,
`set_mode`, `model_params_summary_{dict,print}`.

Differences by design:
  * model names are resolved OFFLINE: a local directory is loaded with from_pretrained, a known name
    ("bert-base-uncased", ...) is built from its architecture config with random init -- the reference fetches
    weights by name (Bagon.py:25-27), there is no network here;
  * forward without autograd runs on the TrainEngine's HIP schedule (kvq/engine.py: own MFMA GEMMs, MFMA attention, fused
    LayerNorm / GELU kernels, bf16 with f32 master weights); with autograd it runs kvq/bert.py, an ATen restatement of HF's math
    that torch can differentiate;
    `backend="hf"` keeps HF's own forward reachable (it is the oracle in tests/test_abi_and_host.py::test_bert_plan_equals_huggingface_forward and tests/test_engine_gpu.py::test_hf_forward_kvq_path_and_engine_agree_on_gpu).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from common.consts import COLOR_FROZEN, COLOR_TOT, COLOR_TRAIN
from common.model_utils import n_not_trainable_params, n_params, n_trainable_params, print_module_params_summary
from kvq import bert as kbert

SUPPORTED_MODEL_MODES = ["full", "dec-head-ft", "enc-head-ft-dec-head-ft", "vq-ft"]

# architecture of the names the reference uses; extra tiny entries are for tests / smoke
LOCAL_BERT_CONFIGS = {
    "bert-base-uncased": dict(),   # BertConfig() defaults ARE bert-base: 768/12/12/3072, vocab 30522, 512 positions
    "kvq-bert-small": dict(hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024),
    "kvq-bert-tiny": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                          vocab_size=2048, max_position_embeddings=64),
    "kvq-bert-tiny-nodrop": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                 vocab_size=2048, max_position_embeddings=64, hidden_dropout_prob=0.0,
                                 attention_probs_dropout_prob=0.0),
    # nine 64-wide heads = nine factor slices for the 9-codebook quantiser of BASELINE.json configs[4]
    "kvq-bert-9x64": dict(hidden_size=576, num_hidden_layers=2, num_attention_heads=9, intermediate_size=2304,
                          vocab_size=2048, max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
    # bert-base widths (768 / 12 heads / 3072 / vocab 30522) with two layers: every GEMM shape of the benchmarked step
    # at a size a parity test can afford (tests/test_engine_base_shapes_gpu.py)
    "kvq-bert-base-2l": dict(num_hidden_layers=2),
    # architecture of tests/golden/step_tiny.npz (tests/golden/make_step_golden.py::CFG)
    "kvq-bert-fixture": dict(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256,
                             vocab_size=512, max_position_embeddings=32),
}


def build_bert_pair(encoder_model_name: str, decoder_model_name: str):
    """What EncoderDecoderModel.from_encoder_decoder_pretrained(enc, dec) yields (Bagon.py:25-31), offline:
    encoder = BertModel, decoder = BertLMHeadModel(is_decoder=True, add_cross_attention=True)."""
    from transformers import BertConfig, BertLMHeadModel, BertModel

    def cfg_of(name, **extra):
        if os.path.isdir(name):
            return BertConfig.from_pretrained(name, **extra), name
        if name not in LOCAL_BERT_CONFIGS:
            raise ValueError(f"unknown model name {name!r}: give a local directory or one of {sorted(LOCAL_BERT_CONFIGS)} "
                             f"(no network: names cannot be fetched)")
        return BertConfig(**LOCAL_BERT_CONFIGS[name], **extra), None

    ecfg, edir = cfg_of(encoder_model_name)
    dcfg, ddir = cfg_of(decoder_model_name, is_decoder=True, add_cross_attention=True)
    encoder = BertModel.from_pretrained(edir, config=ecfg) if edir else BertModel(ecfg)
    decoder = BertLMHeadModel.from_pretrained(ddir, config=dcfg) if ddir else BertLMHeadModel(dcfg)
    return encoder, decoder


class Bagon(nn.Module):
    def __init__(self, encoder_model_name: str, decoder_model_name: str, cross_attn_make_trainable: bool = False,
                 compute_dtype: torch.dtype = torch.bfloat16, backend: str = "kvq"):
        super().__init__()
        self.encoder, self.decoder = build_bert_pair(encoder_model_name, decoder_model_name)
        self.encoder_model_name = encoder_model_name
        self.decoder_model_name = decoder_model_name
        self.cross_attn_make_trainable = cross_attn_make_trainable
        self.compute_dtype = compute_dtype
        self.backend = backend
        self.model_mode = "full"

    # ---- forward -------------------------------------------------------------------------------------------------
    def encode(self, input_ids, attention_mask):
        if self.backend == "hf":
            return self.encoder(input_ids, attention_mask=attention_mask).last_hidden_state
        return kbert.encoder_forward(self.encoder, input_ids, attention_mask, self.compute_dtype)

    def decode_hidden(self, encoder_hidden_states, input_ids, attention_mask):
        """768-d prediction-head states; pair with `kvq.fused_cross_entropy(lm logits)` to avoid [N,V] temporaries."""
        return kbert.decoder_hidden_forward(self.decoder, input_ids, attention_mask, encoder_hidden_states, self.compute_dtype)

    def decode(self, encoder_hidden_states, input_ids, attention_mask):
        if self.backend == "hf":
            return self.decoder(encoder_hidden_states=encoder_hidden_states, input_ids=input_ids,
                                attention_mask=attention_mask).logits
        return kbert.decoder_forward(self.decoder, input_ids, attention_mask, encoder_hidden_states, self.compute_dtype)

    def _engine_forward_ok(self, *ids) -> bool:
        """Forward on the TrainEngine's HIP schedule (own GEMMs, MFMA attention, fused LayerNorm / GELU kernels): whenever no
        autograd graph is wanted and the shapes are the ones the kernels are written for.  With gradients enabled the call goes
        through kvq/bert.py, the ATen restatement that torch autograd can differentiate (it doubles as the engine's checker)."""
        if self.backend != "kvq" or torch.is_grad_enabled() or not all(t.is_cuda for t in ids):
            return False
        from kvq.engine import TrainEngine
        return TrainEngine.supports(self, max(t.shape[1] for t in ids))

    def _engine_autograd_ok(self, *ids) -> bool:
        """autograd_backend = "engine" (default "aten"): a forward WITH autograd also runs on the engine's kernels, its backward is
        the engine's backward schedule seeded with the caller's d L / d logits (kvq.engine.engine_autograd_forward).  Covers the
        autoencoding call (decoder input = encoder input), one process, eager."""
        if self.backend != "kvq" or not torch.is_grad_enabled() or getattr(self, "autograd_backend", "aten") != "engine" \
                or not all(t.is_cuda for t in ids) or any(t is not ids[0] for t in ids):
            return False
        from kvq.engine import TrainEngine
        return TrainEngine.supports(self, max(t.shape[1] for t in ids))

    def forward(self, encoder_input_ids, encoder_attention_mask, decoder_input_ids, decoder_attention_mask):
        if self._engine_autograd_ok(encoder_input_ids, decoder_input_ids):
            from kvq.engine import engine_autograd_forward
            return engine_autograd_forward(self, encoder_input_ids, encoder_attention_mask)[0]
        if self._engine_forward_ok(encoder_input_ids, decoder_input_ids):
            from kvq.engine import engine_of
            same = decoder_input_ids is encoder_input_ids
            out = engine_of(self).forward_logits(encoder_input_ids, encoder_attention_mask, None if same else decoder_input_ids,
                                                 None if same else decoder_attention_mask, training=self.training)
            return out["logits"]
        encoder_output = self.encode(encoder_input_ids, encoder_attention_mask)            # Bagon.py:46-48
        return self.decode(encoder_output, decoder_input_ids, decoder_attention_mask)      # Bagon.py:50-55

    def forward_loss(self, encoder_input_ids, encoder_attention_mask, decoder_input_ids, decoder_attention_mask):
        """Fused step body of the plain autoencoder: (loss_recon, acc_per_batch, recon_ids) -- the loss block of
        models/bagon/Trainer.py:103-110 in one pass over the logits."""
        from kvq.functional import fused_cross_entropy
        enc = self.encode(encoder_input_ids, encoder_attention_mask)
        hidden = self.decode_hidden(enc, decoder_input_ids, decoder_attention_mask)
        logits = kbert.lm_head_logits(self.decoder, hidden, self.compute_dtype)
        return fused_cross_entropy(logits, decoder_input_ids, inplace_backward=True)

    # ---- bookkeeping ---------------------------------------------------------------------------------------------
    def _summary_parts(self):
        return [("encoder", "Encoder", self.encoder), ("decoder", "Decoder", self.decoder)]

    def model_params_summary_dict(self):
        return {key: {"n_trainable_params": n_trainable_params(m), "n_not_trainable_params": n_not_trainable_params(m),
                      "n_params": n_params(m)} for key, _, m in self._summary_parts()}

    def model_params_summary_print(self):
        for _, title, m in self._summary_parts():
            print_module_params_summary(m, title, COLOR_TRAIN, COLOR_FROZEN, COLOR_TOT)

    # ---- trainability modes (Bagon.py:87-179) ---------------------------------------------------------------------
    @staticmethod
    def _module_make_trainable(module, flag: bool):
        for p in module.parameters():
            p.requires_grad = flag

    def _encoder_make_trainable(self, flag: bool):
        self._module_make_trainable(self.encoder, flag)

    def _decoder_make_trainable(self, flag: bool):
        self._module_make_trainable(self.decoder, flag)

    def _decoder_lm_head_make_trainable(self, flag: bool):
        # decoder.cls.predictions.{transform.dense, decoder}; the latter's weight is TIED to the decoder word
        # embeddings, so the embedding table trains in dec-head-ft mode too
        head = self.decoder.cls.predictions
        self._module_make_trainable(head.transform.dense, flag)
        self._module_make_trainable(head.decoder, flag)

    def _decoder_cross_attn_make_trainable(self, flag: bool):
        for layer in self.decoder.bert.encoder.layer:
            self._module_make_trainable(layer.crossattention, flag)

    def _set_mode_dec_head_ft(self):
        self.model_mode = "dec-head-ft"
        self._encoder_make_trainable(False)
        self._decoder_make_trainable(False)
        self._decoder_lm_head_make_trainable(True)
        self._decoder_cross_attn_make_trainable(self.cross_attn_make_trainable)

    def _set_mode_enc_head_dec_head_ft(self):
        # Order as in the reference (models/bagon/Bagon.py:139-146): the label is assigned first and _set_mode_dec_head_ft() then
        # overwrites it, so `model_mode` reads "dec-head-ft" in this mode too -- kept, a consumer of the attribute sees what the
        # reference shows; the parameter sets are what the mode name says (tests/test_engine_gpu.py::test_freeze_modes_match_reference_counts)
        self.model_mode = "enc-dec-head-ft"
        self._set_mode_dec_head_ft()
        self._module_make_trainable(self.encoder.encoder.layer[-1], True)
        if self.encoder.pooler is not None:
            self._module_make_trainable(self.encoder.pooler, True)

    def set_mode(self, model_mode: str):
        if model_mode == "full":
            self.model_mode = "full"
        elif model_mode == "dec-head-ft":
            self._set_mode_dec_head_ft()
        elif model_mode == "enc-head-ft-dec-head-ft":
            self._set_mode_enc_head_dec_head_ft()
        elif model_mode == "vq-ft":
            self.model_mode = "vq-ft"
            self._encoder_make_trainable(False)
            self._decoder_make_trainable(False)
        else:
            raise ValueError(f"Invalid model mode {model_mode}, please use one of the following: {', '.join(SUPPORTED_MODEL_MODES)}")
