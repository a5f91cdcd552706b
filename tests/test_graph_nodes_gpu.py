"""What the captured training step consists of.  include/kvq.h: every entry point of libkvq.so enqueues KERNELS
only -- a hipMemsetAsync captured into a hipGraph becomes a node of another kind, and the one run-to-run difference this engine
ever had (round 4) was such a node losing its stream order inside a replayed graph.  kvq_graph_census (hipGraphGetNodes +
hipGraphNodeGetType) counts the nodes of every graph of the engine's chain by kind; a step graph may hold kernel nodes and the
empty / event nodes of a stream fork, nothing else."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ids(B, S, seed, hi=2000):
    from dsentences.synthetic import random_token_batch
    ids, mask = random_token_batch(B, S, torch.Generator().manual_seed(seed), vocab_hi=hi, min_len=3, max_len=S)
    return ids.cuda(), mask.cuda()


def _model(kind, name="kvq-bert-tiny", dtype=torch.bfloat16):
    from models.bagon.Bagon import Bagon, LOCAL_BERT_CONFIGS
    from models.shelgon3.Shelgon import Shelgon
    H = LOCAL_BERT_CONFIGS[name].get("hidden_size", 768)
    torch.manual_seed(0)
    if kind == "bagon":
        return Bagon(name, name, True, compute_dtype=dtype).cuda().train()
    if kind == "gumbel":
        from models.shelgon3.GumbelQuantizer import GumbelQuantizer
        q = GumbelQuantizer(enc_out_size=H, n_embed=64, embedding_dim=H, temperature=0.9, kl_div_scale=5e-4, straight_through=True)
    elif kind == "multi":
        from models.shelgon3.MultiVectorQuantizer import MultiVectorQuantizer
        q = MultiVectorQuantizer(4, 32, H, 0.25)
    else:
        from models.shelgon3.VectorQuantizer import VectorQuantizer
        q = VectorQuantizer(32, H, 0.25, vq_codebook_init_values=torch.randn(32, H))
        q.materialize_min_encodings = False
    return Shelgon(name, q, name, None, compute_dtype=dtype).cuda().train()


def _census_after_steps(model, B=16, S=12, steps=5, mode=None, **engine_kw):
    from kvq.engine import TrainEngine
    if mode:
        model.set_mode(mode)
    eng = TrainEngine(model, lr=1e-3, **engine_kw)
    ids, mask = _ids(B, S, seed=3)
    kw = {}
    if not eng.has_vq:
        dec = torch.where(torch.rand(ids.shape, device="cuda") < 0.2, torch.randint_like(ids, 1000, 2000), ids) * mask
        kw = dict(dec_ids=dec, dec_mask=mask)
    losses = [float(eng.train_step(ids, mask, **kw)["loss_recon"]) for _ in range(steps)]
    torch.cuda.synchronize()
    assert eng._graphs, "the step was not captured"
    assert np.isfinite(losses).all()
    return next(iter(eng._graphs.values())).node_census(), eng


@pytest.mark.parametrize("kind,dtype", [("vq", torch.bfloat16), ("vq", torch.float32), ("bagon", torch.bfloat16), ("multi", torch.bfloat16),
                                        ("gumbel", torch.bfloat16)])
def test_captured_step_holds_kernel_nodes_only(kind, dtype):
    census, eng = _census_after_steps(_model(kind, dtype=dtype))
    print(kind, dtype, "graphs of the step chain:", census)
    for c in census:
        assert c["memset"] == 0 and c["memcpy"] == 0 and c["other"] == 0, census
    assert sum(c["kernel"] for c in census) >= 40
    # the quantiser of the reference (one codebook) is an eager interlude between two graphs; the plain Bagon step is ONE graph
    assert len(census) == (1 if kind == "bagon" else 2) or kind in ("multi", "gumbel"), census


@pytest.mark.parametrize("mode", ["dec-head-ft", "vq-ft", "enc-head-ft-dec-head-ft"])
def test_captured_step_of_the_freeze_modes_holds_kernel_nodes_only(mode):
    """Bagon.set_mode's partial-training modes take other branches of the backward schedule (no batched cross-K/V gradient, zeroed
    encoder-output gradient): they must not bring a torch fill / copy into the graph as a memset / memcpy node either."""
    census, _ = _census_after_steps(_model("vq"), mode=mode)
    print(mode, census)
    for c in census:
        assert c["memset"] == 0 and c["memcpy"] == 0 and c["other"] == 0, census


def test_quantiser_entry_points_capture_as_kernels():
    """kvq_vq_forward / kvq_vq_forward_packed / kvq_vq_backward / kvq_vq_ema_update captured on their own: kernel nodes only, and
    the replayed graph gives the eager call's bits (round 4's header told callers NOT to capture two of them)."""
    import kvq as K
    from kvq.functional import vq_ema_update
    torch.manual_seed(1)
    N, Kc, D = 2048, 64, 128
    z = torch.randn(N, D, device="cuda").bfloat16()
    E = torch.randn(Kc, D, device="cuda")
    want = K.vector_quantize(z, E, 0.25)
    ema_n, ema_m = torch.ones(Kc, device="cuda"), E.clone()
    E2 = E.clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin()
        got = K.vector_quantize(z, E, 0.25)
        vq_ema_update(z, got[3], ema_n, ema_m, E2, 0.99)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    import ctypes
    from kvq._ffi import check, lib
    counts = (ctypes.c_int64 * 6)()
    check(lib().kvq_graph_census(g.raw_cuda_graph(), counts), "kvq_graph_census")
    kinds = dict(zip(("kernel", "memset", "memcpy", "empty", "event", "other"), counts))
    print("quantiser entry points captured:", kinds)
    assert kinds["memset"] == 0 and kinds["memcpy"] == 0 and kinds["other"] == 0 and kinds["kernel"] >= 5, kinds
    for t in got:
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)
