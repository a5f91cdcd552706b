"""Word-level tokenizer with the call surface the trainers use from HF's BertTokenizer
(models/shelgon3/Trainer.py:82, :224-225): __call__(sentences, return_tensors="pt", padding, max_length,
add_special_tokens) -> .input_ids/.attention_mask, and batch_decode.

Why it exists: BertTokenizer.from_pretrained("bert-base-uncased") needs a vocab file fetched by name; there is no
network and no HF cache here.  `load_tokenizer` returns the real BertTokenizer when given a local directory that
holds a vocab, this one otherwise.  Ids follow BERT conventions: [PAD]=0, [UNK]=100, [CLS]=101, [SEP]=102, words
from 1000 up, so they are valid rows of a 30522-entry embedding table."""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

PAD, UNK, CLS, SEP = 0, 100, 101, 102
FIRST_WORD_ID = 1000


class WordTokenizer:
    def __init__(self, words):
        self.itos = {PAD: "[PAD]", UNK: "[UNK]", CLS: "[CLS]", SEP: "[SEP]"}
        self.stoi = {}
        for i, w in enumerate(sorted(set(words))):
            self.stoi[w] = FIRST_WORD_ID + i
            self.itos[FIRST_WORD_ID + i] = w
        self.vocab_size = 30522
        self.pad_token_id = PAD

    def _encode(self, s: str, add_special_tokens: bool):
        ids = [self.stoi.get(w, UNK) for w in s.lower().split()]
        return [CLS] + ids + [SEP] if add_special_tokens else ids

    def __call__(self, sentences, return_tensors="pt", padding=True, max_length=None, add_special_tokens=True,
                 truncation=False):
        if isinstance(sentences, str):
            sentences = [sentences]
        enc = [self._encode(s, add_special_tokens) for s in sentences]
        if padding == "max_length":
            if max_length is None:
                raise ValueError("padding='max_length' needs max_length")
            width = max_length
            enc = [e[:width] for e in enc] if truncation or any(len(e) > width for e in enc) else enc
        else:
            width = max(len(e) for e in enc)
        ids = torch.zeros((len(enc), width), dtype=torch.int64)
        mask = torch.zeros((len(enc), width), dtype=torch.int64)
        for r, e in enumerate(enc):
            ids[r, :len(e)] = torch.tensor(e, dtype=torch.int64)
            mask[r, :len(e)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)

    def batch_decode(self, sequences):
        out = []
        for row in sequences.tolist() if hasattr(sequences, "tolist") else sequences:
            out.append(" ".join(self.itos.get(int(i), "[UNK]") for i in row))
        return out


def load_tokenizer(name_or_path: str):
    """HF BertTokenizer from a local directory if one is given, else the offline word tokenizer."""
    if os.path.isdir(name_or_path) and os.path.exists(os.path.join(name_or_path, "vocab.txt")):
        from transformers import BertTokenizer
        return BertTokenizer.from_pretrained(name_or_path)
    from dsentences.synthetic import vocabulary
    return WordTokenizer(vocabulary())
