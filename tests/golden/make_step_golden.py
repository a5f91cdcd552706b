#!/usr/bin/env python3
"""Generate the composed-step golden vector (SURVEY.md §8(c) golden item 4: ids -> logits / loss / idx).

Run only in the build container (the reference tree does not travel):

    python3 tests/golden/make_step_golden.py

The reference's own Shelgon / Trainer do not import as checked in, so the composition is restated here
with the two pieces of the reference that DO run:
  * the reference's own `VectorQuantizer`, imported from /root/reference/models/shelgon3/VectorQuantizer.py, and
  * HuggingFace's BertModel / BertLMHeadModel (is_decoder, add_cross_attention) -- what
    EncoderDecoderModel.from_encoder_decoder_pretrained builds (models/bagon/Bagon.py:24-31) -- from a tiny local config,
wired exactly as models/shelgon3/Shelgon.py:50-73 (encoder -> quantiser -> decoder(encoder_hidden_states=z_q).logits) and
models/shelgon3/Trainer.py:94-105 (one-hot KL "batchmean", argmax(softmax), seq_acc, weighted sum, backward).
CPU, f32, eval mode (dropout off), one thread.  Stored: every parameter (so nothing depends on an RNG stream), the ids,
and the outputs: logits, losses, perplexity, indices, recon ids, accuracy, and the gradient of a few parameters.
Only numbers are stored, none of the reference's source text.
"""
import os
import sys

import numpy as np
import torch
from torch.nn.functional import kl_div, log_softmax, one_hot, softmax

REF = "/root/reference"
sys.path.insert(0, REF)
from models.shelgon3.VectorQuantizer import VectorQuantizer  # noqa: E402  (reference module)

OUT = os.path.dirname(os.path.abspath(__file__))

# must equal LOCAL_BERT_CONFIGS["kvq-bert-fixture"] in kindergarten-vq-vae_amd/models/bagon/Bagon.py
CFG = dict(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=512,
           max_position_embeddings=32)
K, BETA, B, S = 24, 0.25, 6, 12
GRAD_KEYS = ["encoder.encoder.layer.0.attention.self.query.weight", "encoder.encoder.layer.0.output.dense.bias",
             "decoder.bert.encoder.layer.0.crossattention.self.key.weight", "decoder.bert.encoder.layer.0.intermediate.dense.weight",
             "decoder.cls.predictions.transform.dense.weight", "decoder.cls.predictions.bias",
             "encoder.embeddings.LayerNorm.weight", "decoder.bert.embeddings.position_embeddings.weight"]


def main():
    from transformers import BertConfig, BertLMHeadModel, BertModel
    torch.set_num_threads(1)
    torch.manual_seed(20240)
    enc = BertModel(BertConfig(**CFG)).eval()
    dec = BertLMHeadModel(BertConfig(**CFG, is_decoder=True, add_cross_attention=True)).eval()
    # make every parameter non-trivial (HF initialises biases / LayerNorm to 0 / 1)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for m in (enc, dec):
            for n, p in m.named_parameters():
                if p.dim() == 1:
                    p.add_(0.05 * torch.randn(p.shape, generator=g))
    E0 = torch.randn(K, CFG["hidden_size"], generator=g)
    vq = VectorQuantizer(n_e=K, e_dim=CFG["hidden_size"], beta=BETA, vq_codebook_init_values=E0)

    ids = torch.randint(100, CFG["vocab_size"], (B, S), generator=g)
    lens = torch.tensor([12, 9, 5, 12, 3, 7])
    ids = ids * (torch.arange(S)[None] < lens[:, None])
    mask = (ids != 0).long()
    V = CFG["vocab_size"]

    embeds = enc(ids, attention_mask=mask).last_hidden_state                                     # Shelgon.py:52
    loss_vq, z_q, perp, _enc1h, idx = vq.forward(embeds, "cpu")                                   # Shelgon.py:58
    logits = dec(encoder_hidden_states=z_q, input_ids=ids, attention_mask=mask).logits           # Shelgon.py:71
    loss_recon = kl_div(input=log_softmax(logits.reshape(-1, V), dim=-1),
                        target=one_hot(ids, V).reshape(-1, V).float(), reduction="batchmean")    # Trainer.py:94-98
    recon_ids = torch.argmax(softmax(logits, dim=-1), dim=-1)                                     # Trainer.py:100
    acc = (recon_ids == ids).sum() / ids.numel()                                                  # common/metrics.py:25-30
    (loss_recon + loss_vq).backward()                                                             # Trainer.py:103-111

    rec = dict(cfg_keys=np.array(sorted(CFG)), cfg_vals=np.array([CFG[k] for k in sorted(CFG)], dtype=np.int64),
               K=K, beta=np.float32(BETA), ids=ids.numpy(), mask=mask.numpy(),
               logits=logits.detach().numpy(), loss_recon=np.float32(loss_recon.item()), loss_vq=np.float32(loss_vq.item()),
               perplexity=np.float32(perp.item()), idx=idx.reshape(-1).numpy().astype(np.int32),
               recon_ids=recon_ids.numpy().astype(np.int32), acc=np.float32(acc.item()),
               z_e=embeds.detach().numpy(), codebook=E0.numpy(), grad_codebook=vq.embedding.weight.grad.numpy())
    params = {}
    for pre, m in (("encoder.", enc), ("decoder.", dec)):
        for n, p in m.state_dict().items():
            if p.dtype == torch.float32:
                rec["p:" + pre + n] = p.numpy()
        params.update({pre + n: p for n, p in m.named_parameters()})
    for k in GRAD_KEYS:
        rec["g:" + k] = params[k].grad.numpy()
    # fp64 gap of the two closest codes per token: how far the indices are from a rounding decision
    z64 = embeds.detach().double().reshape(-1, CFG["hidden_size"])
    d64 = (z64 ** 2).sum(1, keepdim=True) + (E0.double() ** 2).sum(1)[None] - 2.0 * z64 @ E0.double().t()
    top2 = torch.topk(d64, 2, dim=1, largest=False).values
    rec["gap64"] = (top2[:, 1] - top2[:, 0]).numpy()
    path = os.path.join(OUT, "step_tiny.npz")
    np.savez_compressed(path, **rec)
    print(f"step_tiny: loss_recon={loss_recon.item():.6f} loss_vq={loss_vq.item():.6f} perp={perp.item():.4f} acc={acc.item():.4f} "
          f"min top-2 gap={rec['gap64'].min():.3g} codes used={len(np.unique(rec['idx']))} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
