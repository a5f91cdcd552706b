"""CPU-side checks of round 4's host logic: the GEMM tile rule, batch packing layout, the Bagon trainer's bookkeeping (the
reference's keys, models/bagon/Trainer.py:132-203), the token cache's labels / pad handling."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kindergarten-vq-vae_amd")


def test_pick_tile_rule_reproduces_the_measured_choices():
    """kvq.nnops.pick_tile / persistent_pays: at the benchmarked 8192 rows the rule's picks are the per-shape
    choices measured in rounds 2 - 3 (the tables it replaced); at the reference's own row counts it turns to the small tile."""
    from kvq import nnops
    name = lambda M, N, K, lay="nt": nnops.TILE_NAMES[nnops.pick_tile(M, N, K)] + ("p" if nnops.persistent_pays(nnops.pick_tile(M, N, K), M, N, K, lay) else "")
    # forward (y = x W^T): the former _OWN_FWD table, (N, K) -> tile
    assert [name(8192, n, k) for n, k in ((768, 768), (768, 3072), (2304, 768), (18432, 768), (30528, 768))] == \
        ["128x192", "128x192", "128x192p", "256x256p", "256x256"]
    assert name(8192, 3072, 768) == "256x192"                      # (FFN1: the tile its GELU epilogue exists for)
    # input gradients (gx = gy W): the former _OWN_DGRAD table; the persistent form is NT only
    assert [name(8192, n, k, "nn") for n, k in ((768, 768), (768, 2304), (768, 3072), (3072, 768), (768, 18432), (768, 30528))] == \
        ["128x192", "128x192", "128x192", "256x192", "128x192", "128x192"]
    # weight gradients with a launch of their own
    assert name(18432, 768, 8192, "tn") == "256x256" and name(30528, 768, 8192, "tn") == "128x256"
    # the reference's batches: 12 tokens x 64 / 128 sentences
    assert name(768, 768, 768) == "64x128" and name(1536, 768, 768) == "64x128" and name(768, 768, 18432, "nn") == "64x128"
    # monotone sanity of the cost model: more rows never cost less on the same tile
    for t in range(5):
        c = [nnops.tile_cost_us(t, M, 768, 768) for M in (768, 1536, 3072, 6144, 8192, 16384)]
        assert all(b >= a for a, b in zip(c, c[1:])), (t, c)


def test_pack_layout_round_trips():
    """TrainEngine.unpack_batch on a pack built by hand: the [4, N] autoencoding form and the flat 4N + 5Nd two-sided form."""
    from kvq._ffi import KvqError
    from kvq.engine import TrainEngine
    B, S, Sd = 3, 5, 4
    rows = [torch.arange(B * S) + 100 * i for i in range(4)]
    ids, mask, srt, perm, dec = TrainEngine.unpack_batch(torch.stack(rows), (B, S))
    assert dec is None and ids.shape == (B, S) and torch.equal(ids.reshape(-1), rows[0]) and torch.equal(perm, rows[3])
    drows = [torch.arange(B * Sd) + 1000 * (i + 1) for i in range(5)]
    flat = torch.cat(rows + drows)
    ids, mask, srt, perm, dec = TrainEngine.unpack_batch(flat, (B, S), (B, Sd))
    assert torch.equal(mask.reshape(-1), rows[1]) and dec[0].shape == (B, Sd) and torch.equal(dec[4].reshape(-1), drows[4])
    assert torch.equal(dec[2], drows[2]) and torch.equal(dec[3], drows[3])
    with pytest.raises(KvqError):
        TrainEngine.unpack_batch(flat, (B, S))                      # a two-sided pack handed to an autoencoding call
    with pytest.raises(KvqError):
        TrainEngine.unpack_batch(flat.to(torch.int32), (B, S), (B, Sd))


def test_bagon_trainer_bookkeeping_keys_and_best_flags():
    """models/bagon/Trainer.py:132-203 of the reference: run / best dicts, their keys, the x100 accuracy, best flags, the wandb dict."""
    sys.path.insert(0, PKG)
    from models.bagon import Trainer as T
    run, best = T.init_stats_run(), T.init_stats_best()
    assert set(run) == {"loss_recon_run", "loss_full_run", "metric_acc_run", "padding_tokens_pct_run"}
    assert set(best) == {"loss_recon_best", "loss_recon_is_best", "loss_full_best", "loss_full_is_best", "metric_acc_best", "metric_acc_is_best"}
    step = {"loss_recon_step": torch.tensor(2.0), "loss_full_step": torch.tensor(2.0), "metric_acc_step_per_batch": torch.tensor(0.25),
            "metric_acc_step_per_sentence": torch.tensor([0.5, 0.0]), "padding_tokens_pct_step": -69}
    run = T.end_of_step_stats_update(run, step, 2)
    run = T.end_of_step_stats_update(run, dict(step, loss_recon_step=torch.tensor(4.0), loss_full_step=torch.tensor(4.0)), 6)
    run, best = T.end_of_epoch_stats_update(run, best, 8, 2)
    assert run["loss_recon_run"] == pytest.approx((2 * 2 + 4 * 6) / 8) and run["metric_acc_run"] == pytest.approx(25.0)
    assert run["padding_tokens_pct_run"] == -69 and best["loss_recon_is_best"] and best["metric_acc_is_best"] and best["loss_full_best"] == run["loss_full_run"]
    log = T.create_wandb_log_dict(3, run, "val")
    assert set(log) == {"epoch", "val/loss_recon", "val/loss_full", "val/acc", "padding_tokens_pct/val"}
    worse = T.init_stats_run()
    worse = T.end_of_step_stats_update(worse, dict(step, loss_recon_step=torch.tensor(9.0), loss_full_step=torch.tensor(9.0)), 4)
    _, best = T.end_of_epoch_stats_update(worse, best, 4, 1)
    assert not best["loss_recon_is_best"] and best["loss_recon_best"] == pytest.approx(3.5)
    ex = T.explicit_latent_classes_labels(torch.tensor([1, 2, 0, 1, 1, 7, 7, 7, 7]))
    assert ex == {"sentence_type": "interrogative", "grammatical_number_person": "3rd", "sentence_negation": "affirmative",
                  "verb_tense": "present", "sentence_style": "progressive"}
    assert T.explicit_latent_classes_labels(torch.tensor([5, 0, 0, 0, 0]))["sentence_type"] == "5"        # outside the reference's table


def test_token_cache_hands_out_labels_and_respects_the_packed_pad_id():
    sys.path.insert(0, PKG)
    from dsentences.token_cache import TokenCache, cache_of_split
    from kvq.tokenizer import load_tokenizer
    from dsentences.synthetic import make_corpus
    sentences, labels, _ = make_corpus(40, seed=1)
    tok = load_tokenizer("bert-base-uncased")
    cache = TokenCache(list(sentences), tok, 12, False, "cpu", labels=torch.as_tensor(labels))
    b = cache.batch(torch.tensor([3, 7, 11]))
    assert b["latent_classes_labels"].shape == (3, 9) and torch.equal(b["latent_classes_labels"], torch.as_tensor(labels)[[3, 7, 11]])
    assert "packed" not in b                                        # packs are built on the device only
    assert cache.packed_pad_id == cache.pad_id == 0

    class Split:                                                     # what random_split hands the mains
        def __init__(self, ds, idx):
            self.dataset, self.indices = ds, idx

    class DS:
        pass
    ds = DS()
    ds.sentences, ds.latent_classes_labels = list(sentences), torch.as_tensor(labels)
    c2 = cache_of_split(Split(ds, [5, 2, 9]), tok, 12, False, "cpu", keep_labels=True)
    assert len(c2) == 3 and torch.equal(c2.labels, torch.as_tensor(labels)[[5, 2, 9]])
    assert cache_of_split(Split(ds, [5, 2, 9]), tok, 12, False, "cpu").labels is None


def test_fp8_span_table_names_the_segment_of_every_span():
    """kvq.engine.fp8_span_table (round 5): the table the Adam kernel reads to find the fp8 scale of the 8 weights it has just updated
    (kvq_adam_step_dev_fp8).  For random non-overlapping segments in multiples of 16: a span wholly inside one segment names it, a
    span no segment touches is -1, everything else -2 -- and walking the segment table for the -2 spans agrees with a brute-force map."""
    import numpy as np
    from kvq.engine import fp8_span_table
    rng = np.random.default_rng(0)
    for trial in range(20):
        n_total = int(rng.integers(3000, 40000)) // 16 * 16
        cuts = np.sort(rng.choice(np.arange(1, n_total // 16), size=int(rng.integers(2, 12)), replace=False)) * 16
        bounds = [0] + cuts.tolist() + [n_total]
        segs = [(bounds[i], bounds[i + 1] - bounds[i]) for i in range(len(bounds) - 1) if rng.random() < 0.6]      # some stretches belong to nobody
        offs, ns = [o for o, _ in segs], [c for _, c in segs]
        table = fp8_span_table(offs, ns, n_total)
        owner = np.full(n_total, -1)
        for si, (o, c) in enumerate(segs):
            owner[o:o + c] = si
        for k, t in enumerate(table):
            part = owner[2048 * k: 2048 * k + 2048]
            if t >= 0:
                assert (part == t).all()
            elif t == -1:
                assert (part == -1).all()
            else:
                assert t == -2 and len(set(part.tolist())) > 1
        # 8-element groups never straddle a segment (segments are multiples of 16): what the kernel's walk relies on
        assert all((owner[e:e + 8] == owner[e]).all() for e in range(0, n_total, 8))
