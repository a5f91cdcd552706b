"""Synthetic stand-in for the dSentences corpus (the real .npy files are git-ignored upstream and absent offline).

dSentences sentences are generated from 9 discrete factors (verb/object pair, number, person, tense, style, ...);
examples in the reference: "he accepted the payment", "are you not ruining the holidays", "they were touring the
lakes" (common/test_checkpoint_validity.py:36-38).  `make_corpus` produces sentences of that grammar with their
9 integer factor labels and a one-hot encoding, in the file layout dSentencesDataset expects.
`random_token_batch` is the id-level generator of BASELINE.md §3 used by bench.py."""
import os

import numpy as np
import torch

SUBJECTS = [("i", "am", "was"), ("you", "are", "were"), ("he", "is", "was"), ("she", "is", "was"),
            ("we", "are", "were"), ("they", "are", "were")]
VERBS = [("accept", "accepted", "accepting"), ("ruin", "ruined", "ruining"), ("tour", "toured", "touring"),
         ("paint", "painted", "painting"), ("clean", "cleaned", "cleaning"), ("visit", "visited", "visiting"),
         ("open", "opened", "opening"), ("watch", "watched", "watching"), ("count", "counted", "counting"),
         ("order", "ordered", "ordering"), ("check", "checked", "checking"), ("load", "loaded", "loading")]
OBJECTS = [("payment", "payments"), ("holiday", "holidays"), ("lake", "lakes"), ("wall", "walls"), ("room", "rooms"),
           ("museum", "museums"), ("door", "doors"), ("movie", "movies"), ("coin", "coins"), ("meal", "meals"),
           ("ticket", "tickets"), ("truck", "trucks")]
# factor cardinalities: verb, object, obj number, subject, tense(2), aspect(2), negation(2), question(2), article(2)
FACTOR_SIZES = [len(VERBS), len(OBJECTS), 2, len(SUBJECTS), 2, 2, 2, 2, 2]


def _sentence(f):
    v, o, on, s, tense, prog, neg, quest, art = f
    subj, be_now, be_past = SUBJECTS[s]
    base, past, ing = VERBS[v]
    obj = OBJECTS[o][on]
    det = "the" if art == 0 else ("a" if on == 0 else "some")
    if prog:
        be = be_past if tense else be_now
        words = ([be, subj] if quest else [subj, be]) + (["not"] if neg else []) + [ing, det, obj]
    else:
        aux = "did" if tense else ("does" if subj in ("he", "she") else "do")
        if quest or neg:
            words = ([aux, subj] if quest else [subj, aux]) + (["not"] if neg else []) + [base, det, obj]
        else:
            third = base + "s" if (not tense and subj in ("he", "she")) else base
            words = [subj, past if tense else third, det, obj]
    return " ".join(words)


def vocabulary():
    words = {"not", "the", "a", "some", "do", "does", "did"}
    for s in SUBJECTS:
        words.update(s)
    for v in VERBS:
        words.update(v)
        words.add(v[0] + "s")
    for o in OBJECTS:
        words.update(o)
    return sorted(words)


def make_corpus(n_sentences: int, seed: int = 69):
    """-> (sentences: np.ndarray[str], labels: int64 [n,9], one_hot: float32 [n, sum(FACTOR_SIZES)])"""
    rng = np.random.Generator(np.random.PCG64(seed))
    labels = np.stack([rng.integers(0, k, n_sentences) for k in FACTOR_SIZES], axis=1).astype(np.int64)
    sentences = np.array([_sentence(f) for f in labels])
    one_hot = np.zeros((n_sentences, sum(FACTOR_SIZES)), np.float32)
    off = 0
    for c, k in enumerate(FACTOR_SIZES):
        one_hot[np.arange(n_sentences), off + labels[:, c]] = 1.0
        off += k
    return sentences, labels, one_hot


def write_corpus(directory: str, n_sentences: int, seed: int = 69, suffix: str = "_clean"):
    os.makedirs(directory, exist_ok=True)
    s, l, o = make_corpus(n_sentences, seed)
    paths = (os.path.join(directory, f"dSentences_sentences{suffix}.npy"),
             os.path.join(directory, f"dSentences_latent_classes_labels{suffix}.npy"),
             os.path.join(directory, f"dSentences_latent_classes_one_hot{suffix}.npy"))
    np.save(paths[0], s); np.save(paths[1], l); np.save(paths[2], o)
    return paths


def random_token_batch(batch: int, seq_len: int, generator: torch.Generator, vocab_lo=1000, vocab_hi=30000,
                       min_len=4, max_len=12):
    """BASELINE.md §3 input recipe: L ~ U{min_len..max_len} real ids uniform in [vocab_lo, vocab_hi), [PAD]=0 up
    to seq_len; attention_mask = ids != 0.  Returns int64 CPU tensors (ids, mask)."""
    lens = torch.randint(min_len, max_len + 1, (batch,), generator=generator)
    ids = torch.randint(vocab_lo, vocab_hi, (batch, seq_len), generator=generator)
    keep = torch.arange(seq_len)[None, :] < lens[:, None]
    ids = ids * keep
    return ids, keep.long()
