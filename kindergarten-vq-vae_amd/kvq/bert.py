"""ATen restatement of the BERT encoder / cross-attending BERT decoder used by Bagon and Shelgon -- the AUTOGRAD path.

Plain torch ops (F.linear, F.scaled_dot_product_attention, F.layer_norm, F.gelu, F.dropout) over the PARAMETERS of HuggingFace's
BertModel / BertLMHeadModel (so state dicts, init and checkpoints stay those of the reference: models/bagon/Bagon.py:24-31).
This is NOT the product's execution plan: the hand-written HIP schedule lives in kvq/engine.py (training steps, and every
Bagon / Shelgon.forward call made without autograd).  This file is what runs when torch autograd must differentiate the model
(USE_ENGINE = False, sequences longer than the engine's attention kernels take -- 128 tokens in bf16, 32 in f32 --, head widths other than 64) and it is the checker the engine's
gradients are compared with in tests/test_engine_gpu.py.  Differences from HF's module-by-module forward:

  * one fused QKV projection per self-attention ([768 -> 2304] GEMM instead of three), one fused KV projection per
    cross-attention, bf16 operands with f32 accumulation;
  * the LM head stops at the 768-d transform: the caller feeds `lm_head_logits` + the fused loss kernel
    (kvq_ce_forward) so the [N,V] log-softmax / one-hot tensors of the reference are never materialised.

Math restated from transformers/models/bert/modeling_bert.py (v5.15): embeddings :53-108, self/cross attention
:139-280, BertSelfOutput :282-293, BertIntermediate/BertOutput :325-352, BertLayer :354-417, LM head :466-497,
mask construction :688-716.  HF's own forward is the third-party part of the reference; tests use it as the
oracle for this file (tests/test_abi_and_host.py::test_bert_plan_equals_huggingface_forward).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _w(p, dtype):
    return p if p.dtype == dtype else p.to(dtype)


def _layer_norm(x, ln, dtype):
    return F.layer_norm(x, (x.shape[-1],), _w(ln.weight, dtype), _w(ln.bias, dtype), ln.eps)


def _embeddings(emb, input_ids, dtype, p_drop, training):
    """BertEmbeddings.forward (:68-108): word + token_type(0) + absolute position, LayerNorm, dropout."""
    B, S = input_ids.shape
    x = F.embedding(input_ids, emb.word_embeddings.weight, padding_idx=emb.word_embeddings.padding_idx)   # pad row: no gradient (modeling_bert.py:60)
    x = x + emb.token_type_embeddings.weight[0] + emb.position_embeddings.weight[:S]
    x = _layer_norm(x.to(dtype), emb.LayerNorm, dtype)
    return F.dropout(x, p_drop, training)


def _split_heads(t, B, S, nh):
    return t.view(B, S, nh, -1).transpose(1, 2)


def _self_attention(att, x, bias, nh, dtype, p_attn, training, causal):
    B, S, H = x.shape
    sa = att.self
    w = torch.cat([sa.query.weight, sa.key.weight, sa.value.weight], 0)
    b = torch.cat([sa.query.bias, sa.key.bias, sa.value.bias], 0)
    qkv = F.linear(x, _w(w, dtype), _w(b, dtype))
    q, k, v = (_split_heads(t, B, S, nh) for t in qkv.split(H, dim=-1))
    ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, dropout_p=p_attn if training else 0.0)
    return ctx.transpose(1, 2).reshape(B, S, H)


def _cross_attention(att, x, enc, nh, dtype, p_attn, training):
    """BertCrossAttention (:206-280) without an encoder mask: the reference passes none (Shelgon.py:71), so padded
    encoder positions are attended too."""
    B, S, H = x.shape
    Se = enc.shape[1]
    ca = att.self
    q = F.linear(x, _w(ca.query.weight, dtype), _w(ca.query.bias, dtype))
    wkv = torch.cat([ca.key.weight, ca.value.weight], 0)
    bkv = torch.cat([ca.key.bias, ca.value.bias], 0)
    kv = F.linear(enc, _w(wkv, dtype), _w(bkv, dtype))
    k, v = kv.split(H, dim=-1)
    ctx = F.scaled_dot_product_attention(_split_heads(q, B, S, nh), _split_heads(k, B, Se, nh), _split_heads(v, B, Se, nh),
                                         dropout_p=p_attn if training else 0.0)
    return ctx.transpose(1, 2).reshape(B, S, H)


def _out_block(out_mod, h, residual, dtype, p_drop, training):
    """BertSelfOutput / BertOutput (:282-293, :340-352): dense, dropout, LayerNorm(h + residual)."""
    h = F.linear(h, _w(out_mod.dense.weight, dtype), _w(out_mod.dense.bias, dtype))
    h = F.dropout(h, p_drop, training)
    return _layer_norm(h + residual, out_mod.LayerNorm, dtype)


def _layer(layer, x, bias, enc, nh, dtype, p_drop, p_attn, training, causal):
    a = _self_attention(layer.attention, x, bias, nh, dtype, p_attn, training, causal)
    x = _out_block(layer.attention.output, a, x, dtype, p_drop, training)
    if enc is not None:
        c = _cross_attention(layer.crossattention, x, enc, nh, dtype, p_attn, training)
        x = _out_block(layer.crossattention.output, c, x, dtype, p_drop, training)
    h = F.linear(x, _w(layer.intermediate.dense.weight, dtype), _w(layer.intermediate.dense.bias, dtype))
    h = F.gelu(h)                                          # hidden_act "gelu" = erf form (:325-337)
    return _out_block(layer.output, h, x, dtype, p_drop, training)


def attention_bias(attention_mask, dtype, causal):
    """Additive mask [B,1,S,S]: padding keys (and future keys when causal) get a large negative bias (:688-716)."""
    B, S = attention_mask.shape
    keep = attention_mask.bool()[:, None, None, :]
    if causal:
        tri = torch.ones(S, S, dtype=torch.bool, device=attention_mask.device).tril()
        keep = keep & tri[None, None]
    else:
        keep = keep.expand(B, 1, S, S)
    lo = torch.full((), torch.finfo(dtype).min, dtype=dtype, device=keep.device)
    return torch.where(keep, torch.zeros((), dtype=dtype, device=keep.device), lo)


def encoder_forward(encoder, input_ids, attention_mask, dtype=torch.bfloat16):
    """BertModel(...).last_hidden_state for a plain (non-decoder) BERT.  Bagon.py:46-48 / Shelgon.py:52."""
    cfg = encoder.config
    training = encoder.training
    x = _embeddings(encoder.embeddings, input_ids, dtype, cfg.hidden_dropout_prob, training)
    bias = attention_bias(attention_mask, dtype, causal=False)
    for layer in encoder.encoder.layer:
        x = _layer(layer, x, bias, None, cfg.num_attention_heads, dtype, cfg.hidden_dropout_prob,
                   cfg.attention_probs_dropout_prob, training, False)
    return x


def decoder_hidden_forward(decoder, input_ids, attention_mask, encoder_hidden_states, dtype=torch.bfloat16):
    """BertLMHeadModel up to (and including) the prediction-head transform: the 768-d states that feed the
    vocabulary projection (Bagon.py:50-53 / Shelgon.py:71; LM head :466-481)."""
    bert = decoder.bert
    cfg = decoder.config
    training = decoder.training
    x = _embeddings(bert.embeddings, input_ids, dtype, cfg.hidden_dropout_prob, training)
    bias = attention_bias(attention_mask, dtype, causal=True)
    enc = encoder_hidden_states.to(dtype)
    for layer in bert.encoder.layer:
        x = _layer(layer, x, bias, enc, cfg.num_attention_heads, dtype, cfg.hidden_dropout_prob,
                   cfg.attention_probs_dropout_prob, training, True)
    tr = decoder.cls.predictions.transform
    h = F.linear(x, _w(tr.dense.weight, dtype), _w(tr.dense.bias, dtype))
    h = F.gelu(h)
    return _layer_norm(h, tr.LayerNorm, dtype)


def lm_head_logits(decoder, hidden, dtype=torch.bfloat16):
    """Vocabulary projection (:483-497): [.., 768] -> [.., V]; weight tied to the decoder's word embeddings."""
    dec = decoder.cls.predictions.decoder
    return F.linear(hidden, _w(dec.weight, dtype), _w(dec.bias, dtype))


def decoder_forward(decoder, input_ids, attention_mask, encoder_hidden_states, dtype=torch.bfloat16):
    """Full logits, as `decoder(...).logits` of the reference."""
    return lm_head_logits(decoder, decoder_hidden_forward(decoder, input_ids, attention_mask, encoder_hidden_states, dtype), dtype)
