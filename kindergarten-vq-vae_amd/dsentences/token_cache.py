"""Pre-tokenised split resident in device memory -- SURVEY.md §8(f) rank 3 (host input pipeline).

The reference tokenises every batch in Python inside the step (`models/shelgon3/Trainer.py:82-84`) and materialises the epoch
with `list(dl_train)` (`:314`); at > 10 k sentences/s that host work is longer than the GPU step.  Here a split is tokenised
ONCE (same tokenizer call: padding='max_length', max_length, add_special_tokens), kept as one [M, L] int64 tensor -- the whole
576 k-sentence corpus at L = 32 is 147 MB, a rounding error of 288 GB of HBM -- and every batch is an index_select on the
device: no tokenizer, no collate, no H2D copy in the step.  Batches have the layout `Trainer.tokenize_batch` already accepts
({"input_ids", "attention_mask"}), so `step()` / `train()` / `test()` are unchanged.
"""
from typing import Iterator, Optional, Sequence

import torch


class TokenCache:
    def __init__(self, sentences: Sequence[str], tokenizer, max_length: int, add_special_tokens: bool = False,
                 device="cpu", chunk: int = 16384, labels: Optional[torch.Tensor] = None):
        parts = []
        for i in range(0, len(sentences), chunk):
            tok = tokenizer(list(sentences[i:i + chunk]), return_tensors="pt", padding="max_length", max_length=max_length,
                            truncation=True, add_special_tokens=add_special_tokens)
            parts.append(tok.input_ids.to(torch.int64))
        self.pad_id = int(getattr(tokenizer, "pad_token_id", 0) or 0)
        ids = torch.cat(parts) if parts else torch.empty((0, max_length), dtype=torch.int64)
        if ids.shape[1] != max_length:
            raise ValueError(f"tokenizer returned rows of {ids.shape[1]} tokens, expected max_length={max_length}")
        self.input_ids = ids.to(device)                               # [M, L], stays on the device
        self.attention_mask = (self.input_ids != self.pad_id).to(torch.int64)
        self.device = self.input_ids.device
        # the id the "packed" sort files under -1 = the padding_idx of the MODEL's word embeddings (its row receives no gradient,
        # modeling_bert.py:60), which need not be the tokenizer's pad id: the entry points set it from TrainEngine.pad_idx
        # (None = the model has no padding row: nothing is filed away; "off" = batches carry no "packed" entry)
        self.packed_pad_id = self.pad_id
        self.labels = labels.to(self.device) if labels is not None else None       # latent class labels [M, F], handed out per batch

    @classmethod
    def from_ids(cls, input_ids: torch.Tensor, pad_id: int = 0, device=None, labels: Optional[torch.Tensor] = None):
        """A cache over already tokenised rows [M, L] (bench.py's synthetic ids; a corpus tokenised elsewhere)."""
        self = cls.__new__(cls)
        self.pad_id = int(pad_id)
        self.input_ids = input_ids.to(torch.int64).to(device if device is not None else input_ids.device)
        self.attention_mask = (self.input_ids != self.pad_id).to(torch.int64)
        self.device = self.input_ids.device
        self.packed_pad_id = self.pad_id
        self.labels = labels.to(self.device) if labels is not None else None
        return self

    def __len__(self) -> int:
        return int(self.input_ids.shape[0])

    def batch(self, index: torch.Tensor) -> dict:
        """The batch of rows `index`; on the GPU also "packed" [4, B*L] = ids | mask | ids in stable sorted order (pads filed
        under -1) | that order: what kvq.engine.TrainEngine.train_step(prepared=...) takes, so that the replayed step starts
        with one device copy and contains no sort (the order is what the word-embedding gradient is summed in)."""
        index = index.to(self.device)
        ids, mask = self.input_ids.index_select(0, index), self.attention_mask.index_select(0, index)
        out = {"input_ids": ids, "attention_mask": mask}
        if self.labels is not None:
            out["latent_classes_labels"] = self.labels.index_select(0, index)
        if ids.is_cuda and self.packed_pad_id != "off":
            flat = ids.reshape(-1)
            key = flat if self.packed_pad_id is None else torch.where(flat == self.packed_pad_id, torch.full_like(flat, -1), flat)
            srt, perm = torch.sort(key, stable=True)
            out["packed"] = torch.stack([flat, mask.reshape(-1), srt, perm])
        return out

    def loader(self, batch_size: int, shuffle: bool, seed: int = 0, drop_last: bool = False, rank: int = 0, world: int = 1):
        return TokenCacheLoader(self, batch_size, shuffle, seed, drop_last, rank, world)


class TokenCacheLoader:
    """Re-iterable like a DataLoader: len() = batches per epoch, a fresh permutation per epoch (seed + epoch).
    With world > 1 every rank takes an equal, disjoint slice of the (common) permutation; the tail that does not divide is dropped."""

    def __init__(self, cache: TokenCache, batch_size: int, shuffle: bool, seed: int = 0, drop_last: bool = False,
                 rank: int = 0, world: int = 1):
        if batch_size < 1 or not (0 <= rank < world):
            raise ValueError("TokenCacheLoader: bad batch_size / rank / world")
        self.cache, self.batch_size, self.shuffle, self.seed = cache, int(batch_size), bool(shuffle), int(seed)
        self.drop_last, self.rank, self.world = bool(drop_last or world > 1), int(rank), int(world)
        self.epoch = 0

    def _per_rank(self) -> int:
        return len(self.cache) // self.world if self.world > 1 else len(self.cache)

    def __len__(self) -> int:
        n = self._per_rank()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[dict]:
        m = len(self.cache)
        if self.shuffle:
            order = torch.randperm(m, generator=torch.Generator().manual_seed(self.seed + self.epoch))
        else:
            order = torch.arange(m)
        self.epoch += 1
        n = self._per_rank()
        mine = order[self.rank * n:(self.rank + 1) * n].to(self.cache.device)
        for i in range(len(self)):
            yield self.cache.batch(mine[i * self.batch_size:(i + 1) * self.batch_size])


def cache_of_split(split, tokenizer, max_length: int, add_special_tokens: bool, device, keep_labels: bool = False) -> TokenCache:
    """TokenCache of a torch.utils.data.Subset / dataset whose items carry a "sentence" (random_split output of the mains).
    keep_labels: batches also carry the split's "latent_classes_labels" (the Bagon trainer's decode step reads them)."""
    labels = None
    if hasattr(split, "dataset") and hasattr(split, "indices") and hasattr(split.dataset, "sentences"):
        sentences = [split.dataset.sentences[i] for i in split.indices]
        if keep_labels and getattr(split.dataset, "latent_classes_labels", None) is not None:
            labels = split.dataset.latent_classes_labels[torch.as_tensor(list(split.indices), dtype=torch.int64)]
    elif hasattr(split, "sentences"):
        sentences = list(split.sentences)
        if keep_labels:
            labels = getattr(split, "latent_classes_labels", None)
    else:
        items = [split[i] for i in range(len(split))]
        sentences = [it["sentence"] for it in items]
        if keep_labels and items and "latent_classes_labels" in items[0]:
            labels = torch.stack([it["latent_classes_labels"] for it in items])
    return TokenCache(sentences, tokenizer, max_length, add_special_tokens, device, labels=labels)
