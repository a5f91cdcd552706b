"""dsentences/token_cache.py (pre-tokenised split kept on the device): same ids as the per-step tokenizer call of the
reference's Trainer.step (Trainer.py:82-84), epoch coverage, rank partition, and the batch layout Trainer.tokenize_batch takes."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kindergarten-vq-vae_amd"))


def _corpus(n=203):
    from dsentences.synthetic import make_corpus
    return make_corpus(n, seed=5)[0].tolist()


def test_cache_equals_per_batch_tokenisation():
    from dsentences.token_cache import TokenCache
    from kvq.tokenizer import load_tokenizer
    from models.shelgon3.Trainer import tokenize_batch
    tok = load_tokenizer("bert-base-uncased")
    sents = _corpus()
    cache = TokenCache(sents, tok, max_length=12, add_special_tokens=False, device="cpu", chunk=64)
    assert len(cache) == len(sents) and cache.input_ids.shape == (len(sents), 12) and cache.input_ids.dtype == torch.int64
    ref = tok(sents, return_tensors="pt", padding="max_length", max_length=12, add_special_tokens=False)
    assert torch.equal(cache.input_ids, ref.input_ids) and torch.equal(cache.attention_mask, ref.attention_mask)
    b = cache.batch(torch.tensor([5, 0, 77]))
    ids, mask = tokenize_batch(b, tok, False, 12, "cpu")          # the trainer takes cache batches as they are
    assert torch.equal(ids, ref.input_ids[[5, 0, 77]]) and torch.equal(mask, ref.attention_mask[[5, 0, 77]])


def test_loader_covers_the_split_once_per_epoch_and_reshuffles():
    from dsentences.token_cache import TokenCache
    from kvq.tokenizer import load_tokenizer
    cache = TokenCache(_corpus(), load_tokenizer("bert-base-uncased"), 12)
    plain = cache.loader(32, shuffle=False)
    assert len(plain) == 7 and torch.equal(torch.cat([b["input_ids"] for b in plain]), cache.input_ids)
    sh = cache.loader(32, shuffle=True, seed=3, drop_last=True)
    e1 = torch.cat([b["input_ids"] for b in sh]); e2 = torch.cat([b["input_ids"] for b in sh])
    assert len(sh) == 6 and e1.shape == (192, 12) and not torch.equal(e1, e2)
    key = lambda t: sorted(map(tuple, t.tolist()))
    full = key(cache.input_ids)
    assert all(r in full for r in key(e1)[:10])


def test_ranks_take_disjoint_equal_slices():
    from dsentences.token_cache import TokenCache
    from kvq.tokenizer import load_tokenizer
    sents = [f"sentence number {i}" for i in range(101)]
    cache = TokenCache(sents, load_tokenizer("bert-base-uncased"), 8)
    seen = []
    for rank in range(2):
        ld = cache.loader(10, shuffle=True, seed=1, rank=rank, world=2)
        assert len(ld) == 5
        seen.append(torch.cat([b["input_ids"] for b in ld]))
    assert seen[0].shape == seen[1].shape == (50, 8)
    with pytest.raises(ValueError):
        cache.loader(10, False, rank=2, world=2)


def test_cache_of_split_follows_random_split_indices(tmp_path):
    import numpy as np
    from torch.utils.data import random_split
    from dsentences.dataset import dSentencesDataset
    from dsentences.token_cache import cache_of_split
    from kvq.tokenizer import load_tokenizer
    sents = _corpus(50)
    np.save(tmp_path / "s.npy", np.array(sents))
    ds = dSentencesDataset(str(tmp_path / "s.npy"))
    a, b = random_split(ds, (30, 20), torch.Generator().manual_seed(69))
    tok = load_tokenizer("bert-base-uncased")
    ca = cache_of_split(a, tok, 12, False, "cpu")
    want = tok([sents[i] for i in a.indices], return_tensors="pt", padding="max_length", max_length=12, add_special_tokens=False).input_ids
    assert torch.equal(ca.input_ids, want)
