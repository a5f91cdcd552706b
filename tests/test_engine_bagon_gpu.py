"""The plain Bagon training step (models/bagon/Trainer.py:65-130) on the TrainEngine: the encoder and the decoder read their OWN
ids (separately tokenised and perturbed in the reference), the loss target is the decoder's input (or an explicit target).
Checked against torch autograd through HuggingFace's own forward (the third-party part of the reference) and through the ATen
restatement kvq/bert.py; eager against hipGraph replay; a packed batch against the unprepared call."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _build(dtype, name="kvq-bert-tiny", mode="full", seed=0):
    from models.bagon.Bagon import Bagon
    torch.manual_seed(seed)
    model = Bagon(name, name, True, compute_dtype=dtype).cuda()
    model.set_mode(mode)
    return model


def _two_sided_batch(B=6, S=12, Sd=None, seed=1, lo=1000, hi=2000):
    """Encoder / decoder ids that differ the way the reference makes them differ: same sentence, independent token noise."""
    from common.tensor_utils import replace_pct_rand_values
    Sd = Sd or S
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(lo, hi, (B, max(S, Sd)), generator=g)
    lens = torch.randint(3, min(S, Sd) + 1, (B,), generator=g)
    ids = ids * (torch.arange(max(S, Sd))[None] < lens[:, None])
    torch.manual_seed(seed)
    e, d = ids[:, :S].contiguous().cuda(), ids[:, :Sd].contiguous().cuda()
    e_in, d_in = replace_pct_rand_values(e, 0.2, lo, hi), replace_pct_rand_values(d, 0.3, lo, hi)
    return e_in, (e != 0).long(), d_in, (d != 0).long()


def _hf_autograd(model, e, em, d, dm, target=None):
    for p in model.parameters():
        p.grad = None
    model.backend = "hf"
    try:
        logits = model(e, em, d, dm)
        loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), (d if target is None else target).reshape(-1))
        loss.backward()
    finally:
        model.backend = "kvq"
    return dict(loss=loss.item(), recon=logits.argmax(-1).clone(),
                grads={n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})


def _engine_grads(eng, model):
    name_of = {id(p): n for n, p in model.named_parameters()}
    out = {}
    for ename, p in eng.param_of.items():
        if p.requires_grad:
            out[name_of[id(p)]] = eng.flat.g(ename).float()[: p.shape[0]].clone()
    return out


def _rel(g, r):
    return (g - r).norm().item() / max(r.norm().item(), 1e-30)


@pytest.mark.parametrize("mode", ["full", "dec-head-ft", "enc-head-ft-dec-head-ft"])
@pytest.mark.parametrize("Sd", [12, 9])
def test_bagon_engine_f32_every_parameter_against_autograd_through_huggingface(mode, Sd):
    """tiny model, f32, dropout off, encoder ids != decoder ids (and, Sd = 9, another padded length on the decoder side -- the
    reference pads the two sides separately, Trainer.py:78-93): every trainable parameter's gradient, relative L2 < 2e-3."""
    from kvq.engine import TrainEngine
    model = _build(torch.float32, mode=mode).eval()
    e, em, d, dm = _two_sided_batch(Sd=Sd)
    assert not torch.equal(e[:, :Sd], d)
    eng = TrainEngine(model, lr=1e-3)
    out = eng.forward_backward(e, em, training=False, compute_grads=True, dec_ids=d, dec_mask=dm)
    mine = _engine_grads(eng, model)
    ref = _hf_autograd(model, e, em, d, dm)
    np.testing.assert_allclose(out["loss_recon"].item(), ref["loss"], rtol=2e-5)
    assert torch.equal(out["recon_ids"], ref["recon"]) and out["recon_ids"].shape == d.shape
    hits = (ref["recon"] == d).float()
    np.testing.assert_allclose(out["acc"].item(), hits.mean().item(), atol=1e-7)
    torch.testing.assert_close(out["acc_per_sentence"], hits.mean(-1), rtol=0, atol=1e-7)
    assert set(mine) == set(ref["grads"]) and len(mine) > (30 if mode == "full" else 3)
    for n, g in mine.items():
        if not n.endswith("key.bias"):                      # softmax shift invariance: pure rounding noise
            assert _rel(g, ref["grads"][n]) < 2e-3, f"{n}: relative L2 error {_rel(g, ref['grads'][n]):.3g}"
    if mode == "full":      # the two word tables are summed over DIFFERENT id sets: each has gradient rows exactly at ITS side's ids
        for key, side in (("encoder.embeddings.word_embeddings.weight", e), ("decoder.bert.embeddings.word_embeddings.weight", None)):
            if side is not None:
                rows = set(mine[key].abs().sum(1).nonzero().flatten().tolist())
                assert rows == set(side[side != 0].unique().tolist()), key


def test_bagon_engine_separate_target_ids():
    """target_ids: the loss / accuracies score the logits against another tensor than the decoder's input (e.g. noisy decoder
    input, clean target -- not what the reference does, Trainer.py:103, but what its docstring suggests a denoising run wants)."""
    from kvq.engine import TrainEngine
    model = _build(torch.float32).eval()
    e, em, d, dm = _two_sided_batch(seed=4)
    tgt = torch.where(dm.bool(), torch.roll(d, 1, dims=1), d)
    eng = TrainEngine(model, lr=1e-3)
    out = eng.forward_backward(e, em, training=False, compute_grads=True, dec_ids=d, dec_mask=dm, target_ids=tgt)
    mine = _engine_grads(eng, model)
    ref = _hf_autograd(model, e, em, d, dm, target=tgt)
    np.testing.assert_allclose(out["loss_recon"].item(), ref["loss"], rtol=2e-5)
    np.testing.assert_allclose(out["acc"].item(), (ref["recon"] == tgt).float().mean().item(), atol=1e-7)
    for n, g in mine.items():
        if not n.endswith("key.bias"):
            assert _rel(g, ref["grads"][n]) < 2e-3, n
    # ... and with the autoencoding ids (no dec_ids): train_step accepts a target alone
    out2 = eng.eval_step(e, em, target_ids=torch.roll(e, 1, dims=1))
    assert np.isfinite(out2["loss_recon"].item())


def test_bagon_engine_bf16_at_bert_base_shapes():
    """kvq-bert-base-2l (every GEMM shape of the benchmarked step), bf16, 2048 tokens per side, encoder ids != decoder ids:
    the tolerances of tests/test_engine_base_shapes_gpu.py::test_engine_bf16_at_bert_base_shapes against f32 autograd through
    HuggingFace's forward."""
    from kvq.engine import TrainEngine
    e, em, d, dm = _two_sided_batch(B=64, S=32, seed=2, lo=1000, hi=30000)
    m32 = _build(torch.float32, "kvq-bert-base-2l").eval()
    ref = _hf_autograd(m32, e, em, d, dm)
    del m32
    model = _build(torch.bfloat16, "kvq-bert-base-2l").eval()
    eng = TrainEngine(model, lr=1e-4)
    out = eng.forward_backward(e, em, training=False, compute_grads=True, dec_ids=d, dec_mask=dm)
    np.testing.assert_allclose(out["loss_recon"].item(), ref["loss"], rtol=2e-2)
    cos = []
    for n, g in _engine_grads(eng, model).items():
        r = ref["grads"][n]
        if r.norm() > 0 and not n.endswith("key.bias"):
            cos.append((n, F.cosine_similarity(g.reshape(-1), r.reshape(-1), dim=0).item()))
    worst = min(cos, key=lambda t: t[1])
    print("bagon bf16 engine vs f32 autograd: worst gradient cosine", worst, "mean", np.mean([c for _, c in cos]))
    assert worst[1] > 0.97 and np.mean([c for _, c in cos]) > 0.997, (worst, np.mean([c for _, c in cos]))


def test_bagon_train_step_graph_replay_equals_eager_and_matches_torch_adam():
    """engine.train_step(dec_ids=...) for 8 steps on three alternating batches: (a) replayed from hipGraphs == launched eagerly
    (f32, dropout off: same kernels, same order -> same numbers), (b) both == torch.optim.Adam on autograd gradients of the
    ATen restatement, parameter by parameter after the last step."""
    from kvq.engine import TrainEngine
    batches = [_two_sided_batch(B=8, S=12, seed=s) for s in (3, 4, 5)]
    model0 = _build(torch.float32, "kvq-bert-tiny-nodrop").train()
    runs = []
    for use_graph in (False, True):
        model = copy.deepcopy(model0)
        eng = TrainEngine(model, lr=1e-3, milestones=[4], gamma=0.5)
        eng.use_graph = use_graph
        losses = []
        for i in range(8):
            e, em, d, dm = batches[i % 3]
            out = eng.train_step(e, em, dec_ids=d, dec_mask=dm)
            losses.append((float(out["loss_recon"]), float(out["acc"])))
        assert bool(eng._graphs) == use_graph and eng.step_count == 8
        runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}, out))
    (l0, p0, o0), (l1, p1, o1) = runs
    np.testing.assert_allclose(np.array(l0), np.array(l1), rtol=1e-6, atol=1e-7)
    for n in p0:
        torch.testing.assert_close(p0[n], p1[n], rtol=1e-5, atol=1e-7, msg=lambda m: f"{n}: {m}")
    assert o1["acc_per_sentence"].shape == (8,) and o1["recon_ids"].shape == (8, 12)
    # torch autograd + torch Adam on the same batches
    ref = copy.deepcopy(model0)
    opt = torch.optim.Adam([p for p in ref.parameters()], lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[4], gamma=0.5)
    lref = []
    for i in range(8):
        e, em, d, dm = batches[i % 3]
        loss, acc, _ = ref.forward_loss(e, em, d, dm)
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        lref.append(loss.item())
    np.testing.assert_allclose(np.array([l for l, _ in l1]), np.array(lref), rtol=2e-3)
    pr = dict(ref.named_parameters())
    # (key biases: their gradient is rounding noise -- softmax is shift invariant -- and Adam turns any noise into +-lr steps)
    worst = max((_rel(p1[n], pr[n].detach()), n) for n in p1 if "pooler" not in n and not n.endswith("key.bias"))
    assert worst[0] < 2e-3, worst


@pytest.mark.parametrize("two_sided", [False, True])
def test_train_step_prepared_forms_equal_the_unprepared_call(two_sided):
    """train_step(prepared=...) in every accepted form -- pack_batch()'s tensor, prepare_batch()'s tuple, a dict of tuples --
    gives the loss and the parameters of the unprepared call, eagerly (steps 1-2) and on replay (steps 3-6); a pack that holds
    other ids than the call's, or has another size, is refused (the tuple used to crash the first replay)."""
    from kvq._ffi import KvqError
    from kvq.engine import TrainEngine
    model0 = _build(torch.float32, "kvq-bert-tiny-nodrop").train()
    batches = [_two_sided_batch(B=8, S=12, seed=s) for s in (7, 8)]

    def run(form):
        model = copy.deepcopy(model0)
        eng = TrainEngine(model, lr=1e-3)
        losses = []
        for i in range(6):
            e, em, d, dm = batches[i % 2]
            dkw = dict(dec_ids=d, dec_mask=dm) if two_sided else {}
            if form == "none":
                prep = None
            elif form == "pack":
                prep = eng.pack_batch(e, em, *( (d, dm) if two_sided else ()))
            elif form == "tuple":
                prep = eng.prepare_batch(e)
            else:
                prep = dict(enc=eng.prepare_batch(e), dec=eng.prepare_batch(d) if two_sided else None)
            losses.append(float(eng.train_step(e, em, prepared=prep, **dkw)["loss_recon"]))
        assert eng._graphs
        return losses, eng.flat.master.clone(), eng

    base, pbase, eng = run("none")
    for form in ("pack", "tuple", "dict"):
        l, p, _ = run(form)
        np.testing.assert_allclose(np.array(l), np.array(base), rtol=1e-6, atol=1e-7, err_msg=form)
        torch.testing.assert_close(p, pbase, rtol=1e-5, atol=1e-7)
    e, em, d, dm = batches[0]
    dkw = dict(dec_ids=d, dec_mask=dm) if two_sided else {}
    with pytest.raises(KvqError):
        eng.train_step(e, em, prepared=eng.pack_batch(e, em)[:, :-1].contiguous() if not two_sided else eng.pack_batch(e, em), **dkw)
    with pytest.raises(KvqError):
        eng.train_step(e, em, prepared=(eng.prepare_batch(e)[0][:-1], eng.prepare_batch(e)[1][:-1]), **dkw)
    with pytest.raises(KvqError):
        eng.train_step(e, em, prepared="sorted", **dkw)
    # a pack of OTHER ids is caught while the batch shape is new to the engine (the check costs a device sync)
    fresh = TrainEngine(copy.deepcopy(model0), lr=1e-3)
    e2, em2, d2, dm2 = batches[1]
    with pytest.raises(KvqError):
        fresh.train_step(e, em, prepared=fresh.pack_batch(e2, em2, *((d2, dm2) if two_sided else ())), **dkw)


def test_seq_acc_kernel_against_the_torch_version():
    """kvq_seq_acc (common/metrics.py:32-36, second result) on random hits, B not a multiple of the 4 sentences per workgroup."""
    from common.metrics import seq_acc
    from kvq._ffi import check, lib, stream_ptr
    g = torch.Generator().manual_seed(0)
    for B, S in ((1, 1), (7, 12), (256, 32), (33, 100)):
        pred = torch.randint(0, 3, (B, S), generator=g).cuda()
        tgt = torch.randint(0, 3, (B, S), generator=g).cuda()
        out = torch.empty(B, dtype=torch.float32, device="cuda")
        check(lib().kvq_seq_acc(pred.data_ptr(), tgt.data_ptr(), B, S, out.data_ptr(), stream_ptr()), "kvq_seq_acc")
        torch.testing.assert_close(out, seq_acc(pred, tgt)[1], rtol=0, atol=1e-7)


def test_engine_refuses_mismatched_decoder_inputs():
    from kvq._ffi import KvqError
    from kvq.engine import TrainEngine
    model = _build(torch.float32).eval()
    e, em, d, dm = _two_sided_batch()
    eng = TrainEngine(model, lr=1e-3)
    with pytest.raises(KvqError):
        eng.forward_backward(e, em, dec_ids=d[:3], dec_mask=dm[:3])           # another batch size
    with pytest.raises(KvqError):
        eng.forward_backward(e, em, dec_ids=d)                                # no mask
    with pytest.raises(KvqError):
        eng.forward_backward(e, em, dec_ids=d, dec_mask=dm, target_ids=d[:, :5])


def test_bagon_engine_with_different_encoder_and_decoder_stacks():
    """The reference builds encoder and decoder from two names (models/bagon/Bagon.py:16-31; its main.py even allows a GPT-2 decoder
    tokenizer): here a 2-layer / 2048-word encoder feeds a 1-layer / 512-word / 32-position decoder (equal hidden width, which
    cross-attention needs).  Engine (f32) against autograd through HuggingFace's forward, two-sided ids, every parameter."""
    from kvq.engine import TrainEngine
    from models.bagon.Bagon import Bagon
    torch.manual_seed(0)
    model = Bagon("kvq-bert-tiny", "kvq-bert-fixture", True, compute_dtype=torch.float32).cuda().eval()
    assert model.encoder.config.vocab_size != model.decoder.config.vocab_size
    assert model.encoder.config.num_hidden_layers != model.decoder.config.num_hidden_layers
    e, em, _, _ = _two_sided_batch(B=5, S=12, seed=9, lo=1000, hi=2000)
    _, _, d, dm = _two_sided_batch(B=5, S=10, seed=9, lo=100, hi=500)
    eng = TrainEngine(model, lr=1e-3)
    out = eng.forward_backward(e, em, training=False, compute_grads=True, dec_ids=d, dec_mask=dm)
    mine = _engine_grads(eng, model)
    ref = _hf_autograd(model, e, em, d, dm)
    np.testing.assert_allclose(out["loss_recon"].item(), ref["loss"], rtol=2e-5)
    assert torch.equal(out["recon_ids"], ref["recon"]) and set(mine) == set(ref["grads"])
    for n, g in mine.items():
        if not n.endswith("key.bias"):
            assert _rel(g, ref["grads"][n]) < 2e-3, f"{n}: relative L2 error {_rel(g, ref['grads'][n]):.3g}"
    # and it trains through the replayed graphs
    model.train()
    losses = [float(eng.train_step(e, em, dec_ids=d, dec_mask=dm)["loss_recon"]) for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and eng._graphs, losses
