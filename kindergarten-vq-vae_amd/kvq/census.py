"""Word x code census of the quantiser's indices -- host side of `kvq_code_census` (include/kvq.h).

Boundary mirrored: the bookkeeping of analyses/unsupervised_vq_disentanglement/unsupervised_vq_disentanglement.py:156-235.
The reference walks sentence -> word -> token in Python, tokenising every word of every sentence again to learn how many
tokens it has (:174), and appends to Python lists; here

* `WordSpanIndex` turns sentences into ONE int32 per token position (-1 = no word, else (slot << 1) | first-token flag), with
  the per-word token count memoised (each distinct word is tokenised once);
* `CodeCensus` keeps counts[G][W][K] for all tokens and for first tokens on the device, one kernel per batch, no host sync;
* `CodeCensus.results()` reads the tables back once and derives what the reference writes to its three files.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence

import torch

from ._ffi import KvqError, check, lib, require_gpu, stream_ptr


class WordSpanIndex:
    """Distinct words (as written: the reference compares `word in WORDS_OF_INTEREST` on the raw string, :185) -> slots in
    first-seen order; sentences -> the per-position int32 the kernel reads."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer
        self.slot_of: Dict[str, int] = {}
        self.words: List[str] = []
        self._ntok: Dict[str, int] = {}

    def _n_tokens(self, word: str) -> int:
        n = self._ntok.get(word)
        if n is None:                         # (:174) tokenizer(word, padding=False, add_special_tokens=False).input_ids.flatten()
            ids = self.tokenizer(word, return_tensors="pt", padding=False, add_special_tokens=False).input_ids
            n = int(ids.numel())
            self._ntok[word] = n
        return n

    def slot(self, word: str) -> int:
        s = self.slot_of.get(word)
        if s is None:
            s = len(self.words)
            self.slot_of[word] = s
            self.words.append(word)
        return s

    def slot_first(self, sentences: Sequence[str], width: int) -> torch.Tensor:
        """[len(sentences), width] int32 (host): the words of a sentence laid over its token positions in order (:170-178).
        A sentence whose words need more than `width` positions is an error (the reference would index past the row)."""
        out = torch.full((len(sentences), width), -1, dtype=torch.int32)
        for r, s in enumerate(sentences):
            pos = 0
            for word in s.split(" "):
                n = self._n_tokens(word)
                if n == 0:
                    continue
                if pos + n > width:
                    raise KvqError(f"WordSpanIndex: sentence {s!r} needs more than {width} token positions")
                sl = self.slot(word) << 1
                out[r, pos] = sl | 1
                if n > 1:
                    out[r, pos + 1:pos + n] = sl
                pos += n
        return out


class CodeCensus:
    """Device tables counts_all / counts_first [G][W][K] uint32 (W = capacity in distinct words), accumulated over batches."""

    def __init__(self, n_codes: int, capacity_words: int, n_factors: int = 1, device=None):
        if n_codes < 1 or capacity_words < 1 or n_factors < 1:
            raise KvqError("CodeCensus: n_codes, capacity_words, n_factors >= 1")
        self.K, self.W, self.G = int(n_codes), int(capacity_words), int(n_factors)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise KvqError("CodeCensus: the tables live on the GPU (no CPU path)")
        # uint32 cells kept in int32 storage (torch has no uint32 arithmetic; the bits are read back as unsigned)
        self._all = torch.zeros((self.G, self.W, self.K), dtype=torch.int32, device=self.device)
        self._first = torch.zeros_like(self._all)
        self._bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.tokens = 0

    def add(self, slot_first: torch.Tensor, indices: torch.Tensor) -> None:
        """slot_first [B, S] int32, indices [B, S, 1] / [B, S] / [B, S, G] int64 (min_encoding_indices), both on the device."""
        require_gpu(slot_first, indices)
        if slot_first.dtype != torch.int32 or indices.dtype != torch.int64:
            raise KvqError("CodeCensus.add: slot_first int32, indices int64")
        n = slot_first.numel()
        if indices.numel() != n * self.G:
            raise KvqError(f"CodeCensus.add: {indices.numel()} indices for {n} positions x {self.G} factors")
        sf, ix = slot_first.contiguous(), indices.contiguous()
        check(lib().kvq_code_census(sf.data_ptr(), ix.data_ptr(), n, self.G, self.K, self.W, self._all.data_ptr(),
                                    self._first.data_ptr(), self._bad.data_ptr(), stream_ptr()), "kvq_code_census")
        self.tokens += n

    def tables(self):
        """(counts_all, counts_first) as int64 [G, W, K] on the host, and the number of skipped positions."""
        u = lambda t: t.cpu().to(torch.int64) & 0xFFFFFFFF
        return u(self._all), u(self._first), int(self._bad.item())

    def results(self, words: Sequence[str], words_of_interest: Iterable[str], factor: int = 0) -> dict:
        """What the reference writes (:208-235), for one factor's codebook:
        populated          -- set of codes that received a token                                   (seen_v_is)
        histograms[word]   -- {code: times the word's FIRST token took that code} for words of interest, every code 0..K-1 present
        words_of_code[k]   -- the distinct words with a token on code k (sorted; the reference's list(set(...)) has no order)"""
        call, cfirst, bad = self.tables()
        if bad:
            raise KvqError(f"CodeCensus: {bad} positions had a slot beyond the capacity or a code outside [0, {self.K})")
        if len(words) > self.W:
            raise KvqError("CodeCensus.results: more words than table rows")
        a, f = call[factor, :len(words)], cfirst[factor, :len(words)]
        slot_of = {w: i for i, w in enumerate(words)}
        populated = set(torch.nonzero(a.sum(0)).flatten().tolist())
        histograms = {}
        for w in words_of_interest:
            row = f[slot_of[w]].tolist() if w in slot_of else [0] * self.K
            histograms[w] = {k: int(row[k]) for k in range(self.K)}
        words_of_code = {k: sorted(words[i] for i in torch.nonzero(a[:, k]).flatten().tolist()) for k in range(self.K)}
        return {"populated": populated, "histograms": histograms, "words_of_code": words_of_code}
