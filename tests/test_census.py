"""The consumer of the code indices (analyses/unsupervised_vq_disentanglement): oracle on hand-worked cases and the host span index
on the CPU; the HIP census kernel, TrainEngine.code_indices and the analysis script on the GPU, against the oracle's walk."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kindergarten-vq-vae_amd")
sys.path.insert(0, ROOT)
sys.path.insert(0, PKG)

from oracle.census_oracle import census_results, census_walk  # noqa: E402


class _PieceTokenizer:
    """Word-level ids, but words longer than 5 letters are split into two pieces (a stand-in for WordPiece's multi-token words)."""

    def __init__(self):
        self.ids = {}

    def _pieces(self, w):
        return [w] if len(w) <= 5 else [w[:4], "##" + w[4:]]

    def __call__(self, sentences, return_tensors="pt", padding=True, add_special_tokens=False, **_):
        from types import SimpleNamespace
        if isinstance(sentences, str):
            sentences = [sentences]
        enc = [[self.ids.setdefault(p, 1000 + len(self.ids)) for w in s.split(" ") for p in self._pieces(w)] for s in sentences]
        width = max(len(e) for e in enc)
        ids = torch.zeros((len(enc), width), dtype=torch.int64)
        mask = torch.zeros_like(ids)
        for r, e in enumerate(enc):
            ids[r, :len(e)] = torch.tensor(e)
            mask[r, :len(e)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)

    def n_tokens(self, w):
        return len(self._pieces(w))


def test_oracle_on_the_reference_example():
    """The pairing the reference prints for its commented example (:133-139): one code per token, words in order, padding ignored."""
    sentences = ["I was taming the tiger", "I was appreciating the coat"]
    tok = _PieceTokenizer()
    #            I  was tami ##ng the tiger | I  was appr ##eciating the coat
    indices = [[3, 1, 4, 4, 0, 2, 8], [3, 1, 5, 6, 0, 7, 8]]                     # trailing 8s = padding positions of the batch
    woi, by_code, seen = census_walk(sentences, indices, tok.n_tokens, 9, ["I", "was", "the", "it"])
    assert woi == {"I": [3, 3], "was": [1, 1], "the": [0, 0], "it": []}
    assert seen == {0, 1, 2, 3, 4, 5, 6, 7}                                       # code 8 only sits on padding
    assert by_code[4] == ["taming", "taming"] and by_code[5] == ["appreciating"] and by_code[6] == ["appreciating"]
    res = census_results(woi, by_code, seen, 9)
    assert res["histograms"]["I"] == {0: 0, 1: 0, 2: 0, 3: 2, 4: 0, 5: 0, 6: 0, 7: 0, 8: 0}
    assert res["histograms"]["it"] == {k: 0 for k in range(9)}
    assert res["words_of_code"][0] == ["the"] and res["words_of_code"][8] == [] and res["words_of_code"][7] == ["coat"]


def _random_case(seed, n_sent, n_codes, G=1):
    rng = np.random.default_rng(seed)
    vocab = ["i", "you", "he", "was", "were", "not", "the", "a", "painting", "accepted", "holidays", "tiger", "coat", "ruining", "do"]
    sentences = [" ".join(rng.choice(vocab, size=rng.integers(2, 8))) for _ in range(n_sent)]
    tok = _PieceTokenizer()
    t = tok(sentences)
    idx = torch.from_numpy(rng.integers(0, n_codes, size=(n_sent, t.input_ids.shape[1], G)))
    return sentences, tok, t, idx


def _oracle_tables(sentences, tok, idx, n_codes, woi, g=0):
    walk = census_walk(sentences, [row[:, g].tolist() for row in idx], tok.n_tokens, n_codes, woi)
    return census_results(*walk, n_codes)


def test_span_index_reproduces_the_walk_on_the_host():
    """WordSpanIndex + a plain numpy count of (slot, code) pairs = the oracle's walk (no GPU: the kernel's arithmetic in numpy)."""
    from kvq.census import WordSpanIndex
    sentences, tok, t, idx = _random_case(0, 64, 9)
    spans = WordSpanIndex(tok)
    sf = spans.slot_first(sentences, t.input_ids.shape[1]).numpy()
    assert ((sf >= 0) == t.attention_mask.numpy().astype(bool)).all()           # every real token belongs to exactly one word
    W = len(spans.words)
    call, cfirst = np.zeros((W, 9), np.int64), np.zeros((W, 9), np.int64)
    for (r, c), v in np.ndenumerate(sf):
        if v >= 0:
            call[v >> 1, idx[r, c, 0]] += 1
            cfirst[v >> 1, idx[r, c, 0]] += v & 1
    want = _oracle_tables(sentences, tok, idx, 9, ["i", "was", "painting", "zebra"])
    assert set(np.nonzero(call.sum(0))[0].tolist()) == want["populated"]
    for w in ("i", "was", "painting"):
        assert {k: int(cfirst[spans.slot_of[w], k]) for k in range(9)} == want["histograms"][w]
    for k in range(9):
        assert sorted(spans.words[i] for i in np.nonzero(call[:, k])[0]) == want["words_of_code"][k]


def test_span_index_rejects_a_sentence_longer_than_the_row():
    from kvq._ffi import KvqError
    from kvq.census import WordSpanIndex
    with pytest.raises(KvqError):
        WordSpanIndex(_PieceTokenizer()).slot_first(["he was painting the holidays"], 4)


@pytest.mark.gpu
@pytest.mark.parametrize("n_codes,G", [(9, 1), (512, 1), (8192, 1), (32, 9)])
def test_census_kernel_equals_the_walk(n_codes, G):
    """Bit-exact counts: LDS-private tables (9 and 512 codes), global atomics (8192 codes), nine factors; two batches accumulated."""
    from kvq.census import CodeCensus, WordSpanIndex
    woi = ["i", "was", "painting", "zebra"]
    tok_shared = None
    spans = None
    census = CodeCensus(n_codes, 64, G)
    all_sent, all_idx = [], []
    for seed in (1, 2):
        sentences, tok, t, idx = _random_case(seed, 700, n_codes, G)
        tok_shared = tok_shared or tok
        spans = spans or WordSpanIndex(tok_shared)
        census.add(spans.slot_first(sentences, t.input_ids.shape[1]).cuda(), idx.cuda())
        all_sent.append(sentences); all_idx.append(idx)
    for g in range(G):
        got = census.results(spans.words, woi, factor=g)
        walks = [census_walk(s, [row[:, g].tolist() for row in i], tok_shared.n_tokens, n_codes, woi) for s, i in zip(all_sent, all_idx)]
        merged = ({w: walks[0][0][w] + walks[1][0][w] for w in woi}, {k: walks[0][1][k] + walks[1][1][k] for k in range(n_codes)},
                  walks[0][2] | walks[1][2])
        want = census_results(*merged, n_codes)
        assert got["populated"] == want["populated"]
        assert got["histograms"] == want["histograms"]
        assert got["words_of_code"] == want["words_of_code"]
    call, cfirst, bad = census.tables()
    assert bad == 0 and int(call.sum()) == G * sum(int(tok_shared(s).attention_mask.sum()) for s in all_sent)


@pytest.mark.gpu
def test_census_counts_bad_positions_and_refuses_them():
    from kvq._ffi import KvqError
    from kvq.census import CodeCensus
    c = CodeCensus(4, 2)
    sf = torch.tensor([[1, 0, 3, 9, -1]], dtype=torch.int32).cuda()             # slot 4 (9 >> 1) is beyond the two rows
    ix = torch.tensor([[0, 1, 7, 0, 0]], dtype=torch.int64).cuda()              # code 7 is outside [0, 4)
    c.add(sf, ix)
    call, cfirst, bad = c.tables()
    assert bad == 2 and call[0].tolist() == [[1, 1, 0, 0], [0, 0, 0, 0]] and cfirst[0].tolist() == [[1, 0, 0, 0], [0, 0, 0, 0]]
    with pytest.raises(KvqError):
        c.results(["a", "b"], ["a"])


@pytest.mark.gpu
def test_code_indices_equal_the_indices_of_the_whole_forward():
    """Encoder + quantiser alone (TrainEngine.code_indices) return the indices Shelgon.forward returns (bit-exact), without
    running the decoder."""
    from dsentences.synthetic import random_token_batch
    from models.shelgon3.Shelgon import Shelgon
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    torch.manual_seed(0)
    vq = VectorQuantizer(9, 128, 0.1, vq_codebook_init_values=torch.randn(9, 128))
    vq.materialize_min_encodings = False
    model = Shelgon("kvq-bert-tiny", vq, "kvq-bert-tiny", None, compute_dtype=torch.bfloat16).cuda().eval()
    ids, mask = (t.cuda() for t in random_token_batch(16, 12, torch.Generator().manual_seed(3)))
    with torch.no_grad():
        _, _, want, _ = model.forward(ids, mask, ids.device, False)
        got = model.code_indices(ids, mask, ids.device)
    assert got.shape == want.shape == (16, 12, 1) and torch.equal(got, want)


@pytest.mark.gpu
def test_analysis_script_writes_the_reference_files(tmp_path):
    env = dict(os.environ)
    data = str(tmp_path / "data")
    env.update({"PYTHONPATH": PKG, "KVQ_SYNTHETIC_SENTENCES": "2000", "KVQ_BATCH_SIZE": "50", "KVQ_LIM_BATCHES_PCT": "0.5",
                "KVQ_ENCODER_MODEL_NAME": "'kvq-bert-tiny'", "KVQ_DECODER_MODEL_NAME": "'kvq-bert-tiny'", "KVQ_VQ_E_DIM": "128",
                "KVQ_ENC_OUT_SIZE": "128", "KVQ_SENTENCES_PATH": repr(data + "/dSentences_sentences.npy"),
                "KVQ_RESULTS_DIR": repr(str(tmp_path / "results"))})
    script = os.path.join(PKG, "analyses", "unsupervised_vq_disentanglement", "unsupervised_vq_disentanglement.py")
    r = subprocess.run([sys.executable, script], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = tmp_path / "results"
    populated = open(out / "dSentences_vq_vector_populated.txt").read()
    assert populated.startswith("the following VQ latent vectors were populated: {")
    hist = json.load(open(out / "dSentences_words_of_interest_histograms.json"))
    assert set(hist) == {"i", "you", "he", "she", "it", "we", "they", "am", "are", "is", "was", "were", "not", "do", "does", "will"}
    assert all(set(h) == {str(k) for k in range(9)} for h in hist.values())
    by_code = json.load(open(out / "dSentences_vq_words_distrib.json"))
    assert set(by_code) == {str(k) for k in range(9)}
    # every sentence of the synthetic grammar carries "the", "a" or "some" once: the first-token counts of words of interest are
    # bounded by the sentences seen, and a populated code lists at least one word
    n_seen = int(2000 * 0.6 / 50 * 0.5) * 50 + 2 * int(2000 * 0.2 / 50 * 0.5) * 50
    assert 0 < sum(sum(h.values()) for h in hist.values()) <= 3 * n_seen
    codes = eval(populated.split(": ", 1)[1])
    assert codes and all(by_code[str(k)] for k in codes) and all(not by_code[str(k)] for k in range(9) if k not in codes)
