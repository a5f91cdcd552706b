"""The explicit forward/backward TrainEngine against the autograd path (kvq.bert + torch autograd) on the same model:
losses, every parameter gradient, and the Adam update.  f32 compute with dropout off for tight tolerances; bf16 for the
production dtype."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(dtype, mode="full", K=32):
    from models.shelgon3.Shelgon import Shelgon
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    torch.manual_seed(0)
    vq = VectorQuantizer(K, 128, 0.25, vq_codebook_init_values=torch.randn(K, 128))
    vq.materialize_min_encodings = False
    model = Shelgon("kvq-bert-tiny", vq, "kvq-bert-tiny", None, compute_dtype=dtype).cuda()
    model.set_mode(mode)
    return model


def _batch(B=6, S=12, seed=1):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, 2000, (B, S), generator=g)
    lens = torch.randint(3, S + 1, (B,), generator=g)
    ids = ids * (torch.arange(S)[None] < lens[:, None])
    return ids.cuda(), (ids != 0).long().cuda()


def _autograd_reference(model, ids, mask):
    model.eval()                                  # dropout off; the engine is run with training=False as well
    for p in model.parameters():
        p.grad = None
    l_vq, perp, idx, l_rec, acc, recon = model.forward_loss(ids, mask)
    (l_rec + l_vq).backward()
    return dict(loss_recon=l_rec.item(), loss_vq=l_vq.item(), perp=perp.item(), acc=acc.item(), idx=idx.clone(), recon=recon.clone())


@pytest.mark.parametrize("mode", ["full", "dec-head-ft", "enc-head-ft-dec-head-ft"])
def test_engine_f32_matches_autograd(mode):
    from kvq.engine import TrainEngine
    model = _build(torch.float32, mode)
    ids, mask = _batch()
    eng = TrainEngine(model, lr=1e-3)
    ref = _autograd_reference(model, ids, mask)
    out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
    np.testing.assert_allclose(out["loss_recon"].item(), ref["loss_recon"], rtol=2e-5)
    np.testing.assert_allclose(out["loss_vq"].item(), ref["loss_vq"], rtol=2e-5)
    np.testing.assert_allclose(out["perplexity"].item(), ref["perp"], rtol=1e-4)
    assert torch.equal(out["indices"], ref["idx"]) and torch.equal(out["recon_ids"], ref["recon"])
    checked = 0
    for name, p in eng.param_of.items():
        if not p.requires_grad:
            continue
        g = eng.flat.g(name).float()
        assert p.grad is not None, name
        torch.testing.assert_close(g, p.grad, rtol=2e-3, atol=2e-6, msg=lambda m: f"{name}: {m}")
        checked += 1
    assert checked > (10 if mode == 'full' else 3)
    torch.testing.assert_close(eng.gE, model.vector_quantizer.embedding.weight.grad, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("mode", ["full", "dec-head-ft"])
def test_forward_with_autograd_on_the_engine_equals_the_aten_path(mode):
    """model.autograd_backend = "engine": Shelgon.forward under autograd runs the engine's kernels, and loss.backward() on a loss
    the CALLER builds from the returned logits and quantiser loss (here a label-smoothed cross entropy, not the engine's own)
    runs the engine's backward schedule seeded with d L / d logits and d L / d loss_vq.  Same outputs and the same .grad on
    every trainable parameter as the ATen restatement (kvq/bert.py) under torch autograd; a second backward on new inputs
    accumulates into .grad like any other op."""
    import torch.nn.functional as F
    model = _build(torch.float32, mode).eval()
    ids, mask = _batch()

    def user_loss(logits, l_vq):
        return F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), ids.reshape(-1), label_smoothing=0.1) + 0.7 * l_vq

    runs = {}
    for backend in ("aten", "engine"):
        model.autograd_backend = backend
        for p in model.parameters():
            p.grad = None
        l_vq, perp, idx, logits = model.forward(ids, mask, ids.device, False)
        assert logits.requires_grad and type(logits.grad_fn).__name__.startswith("_EngineForward") == (backend == "engine")
        loss = user_loss(logits, l_vq)
        loss.backward()
        runs[backend] = (loss.item(), l_vq.item(), perp.item(), idx.clone(), logits.detach().float().clone(),
                         {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    a, e = runs["aten"], runs["engine"]
    np.testing.assert_allclose(e[0], a[0], rtol=2e-5)
    np.testing.assert_allclose(e[1], a[1], rtol=2e-5)
    np.testing.assert_allclose(e[2], a[2], rtol=1e-4)
    assert torch.equal(e[3], a[3])
    torch.testing.assert_close(e[4], a[4], rtol=1e-4, atol=1e-4)
    assert set(e[5]) == set(a[5]) and len(e[5]) > (10 if mode == "full" else 3)
    for n in a[5]:
        if not n.endswith("key.bias"):
            torch.testing.assert_close(e[5][n], a[5][n], rtol=2e-3, atol=2e-6, msg=lambda m: f"{n}: {m}")
    # accumulation: a second forward / backward on the same batch doubles every gradient
    model.autograd_backend = "engine"
    l_vq, perp, idx, logits = model.forward(ids, mask, ids.device, False)
    user_loss(logits, l_vq).backward()
    n = "vector_quantizer.embedding.weight" if mode == "full" else "decoder.cls.predictions.transform.dense.weight"
    torch.testing.assert_close(dict(model.named_parameters())[n].grad, 2 * e[5][n], rtol=1e-5, atol=1e-9)
    # a forward whose activations were consumed cannot be differentiated twice; a stale one is refused
    from kvq._ffi import KvqError
    l_vq, perp, idx, logits = model.forward(ids, mask, ids.device, False)
    loss = user_loss(logits, l_vq)
    loss.backward(retain_graph=True)
    with pytest.raises(KvqError):
        loss.backward()


def test_engine_adam_step_matches_torch_adam():
    from kvq.engine import TrainEngine
    model = _build(torch.float32)
    ref_model = copy.deepcopy(model)
    ids, mask = _batch(seed=3)
    eng = TrainEngine(model, lr=1e-3, weight_decay=0.01, amsgrad=True, milestones=[2], gamma=0.5)
    opt = torch.optim.Adam(ref_model.parameters(), lr=1e-3, weight_decay=0.01, amsgrad=True)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2], gamma=0.5)
    model.eval(); ref_model.eval()
    for _ in range(3):
        out = eng.train_step(ids, mask)
        opt.zero_grad()
        l_vq, _, _, l_rec, _, _ = ref_model.forward_loss(ids, mask)
        (l_rec + l_vq).backward()
        opt.step(); sched.step()
        np.testing.assert_allclose(out["loss_recon"].item(), l_rec.item(), rtol=1e-4)
    ref_params = dict(ref_model.named_parameters())
    for name, p in model.named_parameters():
        if "pooler" in name or "key.bias" in name:   # key-bias gradients are pure rounding noise (softmax shift invariance)
            continue
        torch.testing.assert_close(p.data, ref_params[name].data, rtol=1e-3, atol=2e-5, msg=lambda m: f"{name}: {m}")


def test_engine_bf16_close_to_f32_autograd():
    from kvq.engine import TrainEngine
    model32 = _build(torch.float32)
    ids, mask = _batch(B=8, S=32, seed=5)
    ref = _autograd_reference(model32, ids, mask)
    model = _build(torch.bfloat16)
    eng = TrainEngine(model, lr=1e-3)
    out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
    np.testing.assert_allclose(out["loss_recon"].item(), ref["loss_recon"], rtol=2e-2)
    np.testing.assert_allclose(out["loss_vq"].item(), ref["loss_vq"], rtol=5e-2)
    p32 = dict(model32.named_parameters())
    cos = []
    for name, p in eng.param_of.items():
        ref_name = [n for n, q in model.named_parameters() if q is p][0]
        g = eng.flat.g(name).float().reshape(-1)
        r = p32[ref_name].grad.reshape(-1)
        if r.norm() > 0 and not name.endswith("k.b"):      # key-bias gradients are rounding noise (softmax shift invariance)
            cos.append(torch.nn.functional.cosine_similarity(g, r, dim=0).item())
    assert min(cos) > 0.9 and np.mean(cos) > 0.99, (min(cos), np.mean(cos))


def test_engine_training_reduces_loss_with_dropout():
    from kvq.engine import TrainEngine
    model = _build(torch.bfloat16).train()
    eng = TrainEngine(model, lr=2e-3)
    ids, mask = _batch(B=16, S=32, seed=7)
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(30)]
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses[::5]
    a = eng.forward_backward(ids, mask, training=True, compute_grads=False)["loss_recon"].item()
    b = eng.forward_backward(ids, mask, training=True, compute_grads=False)["loss_recon"].item()
    assert a == b        # same step seed -> same dropout masks: the step is reproducible
    state = model.state_dict()
    assert "vector_quantizer.embedding.weight" in state and "encoder.embeddings.word_embeddings.weight" in state


def test_forward_only_engine_allocates_no_optimizer_state():
    """The engine behind a no-autograd Shelgon.forward (engine_of(model)) holds master + shadow weights only; the gradient and
    Adam-moment buffers appear with the first training step."""
    from kvq.engine import engine_of
    model = _build(torch.bfloat16).eval()
    ids, mask = _batch()
    with torch.no_grad():
        model.forward(ids, mask, ids.device, False)
    eng = engine_of(model, create=False)
    assert eng is not None and not eng.flat.optimizer_state_allocated()
    model.train()
    eng.train_step(ids, mask)
    assert eng.flat.optimizer_state_allocated() and eng.flat.m.abs().sum().item() > 0


def test_engine_rejects_long_sequences():
    """f32 stops at the 32-token kernels, bf16 at the 128 tokens of the blocked ones."""
    from kvq._ffi import KvqError
    from kvq.engine import TrainEngine
    for dtype, S in ((torch.float32, 40), (torch.bfloat16, 160)):
        model = _build(dtype)
        eng = TrainEngine(model)
        assert not TrainEngine.supports(model, S)
        ids = torch.randint(1000, 2000, (2, S)).cuda()
        with pytest.raises(KvqError):
            eng.train_step(ids, torch.ones_like(ids))


@pytest.mark.parametrize("S", [48, 64])
def test_engine_bf16_above_32_tokens(S):
    """33 .. 128 tokens: the engine runs (blocked attention kernels, everything else is per token); its bf16 gradients point where
    the f32 autograd gradients of the ATen restatement point (same bounds as the 32-token bf16 test), and it trains through the
    captured replay.  (The reference pads to 12 - 14 tokens, BASELINE.json to 32: cover, not benchmark.)"""
    from kvq.engine import TrainEngine
    model32 = _build(torch.float32)
    ids, mask = _batch(B=6, S=S, seed=5)
    ref = _autograd_reference(model32, ids, mask)
    model = _build(torch.bfloat16)
    assert TrainEngine.supports(model, S)
    eng = TrainEngine(model, lr=1e-3)
    out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
    np.testing.assert_allclose(out["loss_recon"].item(), ref["loss_recon"], rtol=2e-2)
    np.testing.assert_allclose(out["loss_vq"].item(), ref["loss_vq"], rtol=5e-2)
    p32 = dict(model32.named_parameters())
    cos = []
    for name, p in eng.param_of.items():
        ref_name = [n for n, q in model.named_parameters() if q is p][0]
        g = eng.flat.g(name).float().reshape(-1)
        r = p32[ref_name].grad.reshape(-1)
        if r.norm() > 0 and not name.endswith("k.b"):
            cos.append(torch.nn.functional.cosine_similarity(g, r, dim=0).item())
    assert min(cos) > 0.9 and np.mean(cos) > 0.99, (min(cos), np.mean(cos))
    model.train()
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and eng._graphs, losses


def test_engine_graph_replay_equals_eager_steps():
    """The two-hipGraph replay of a step (kvq.engine._StepGraphs) against the same steps launched one kernel at a time:
    per-step losses, dropout masks (device-side seed offset), learning-rate milestones and the Adam state must agree."""
    from kvq import nnops
    from kvq.engine import TrainEngine
    batches = [_batch(B=16, S=32, seed=s) for s in (3, 4, 5)]
    runs = []
    for use_graph in (False, True):
        model = _build(torch.bfloat16).train()
        eng = TrainEngine(model, lr=2e-3, milestones=[3, 6], gamma=0.5, seed=99)
        eng.use_graph = use_graph
        losses = []
        for i in range(9):
            out = eng.train_step(*batches[i % 3])
            losses.append((float(out["loss_recon"]), float(out["loss_vq"]), float(out["acc"])))
        assert bool(eng._graphs) == use_graph and eng.step_count == 9
        runs.append((losses, eng.flat.master.clone(), eng.E.detach().clone(), nnops.read_step_state(eng._state), out))
    (l0, p0, e0, st0, o0), (l1, p1, e1, st1, o1) = runs
    np.testing.assert_allclose(np.array(l0), np.array(l1), rtol=2e-2, atol=2e-3)     # embedding scatter-add order is the only noise
    assert st0 == st1 and st0[0] == 9
    np.testing.assert_allclose(st0[1], 2e-3 * 0.25, rtol=1e-6)                        # 9th step: 8 ticks >= both milestones
    np.testing.assert_allclose(st0[2], 1 - 0.9 ** 9, rtol=2e-5)
    np.testing.assert_allclose(st0[3], (1 - 0.999 ** 9) ** 0.5, rtol=2e-5)        # beta2 is f32
    cos = torch.nn.functional.cosine_similarity((p0 - p0.mean()).double(), (p1 - p1.mean()).double(), dim=0).item()
    assert cos > 0.9999, cos
    torch.testing.assert_close(e0, e1, rtol=1e-2, atol=1e-3)
    assert o1["recon_ids"].shape == o0["recon_ids"].shape and o1["indices"].dtype == torch.int64


def test_engine_early_adam_option_equals_the_default_schedule(monkeypatch):
    """KVQ_EARLY_ADAM=1 (parameters updated on a side stream while backward runs, step state prepared / committed in two
    halves) must be the same optimiser: losses, step state and parameters after six steps, eager and replayed from graphs."""
    from kvq import nnops
    from kvq.engine import TrainEngine
    batches = [_batch(B=16, S=32, seed=s) for s in (7, 8)]
    runs = []
    for early, use_graph in (("0", False), ("1", False), ("1", True)):
        monkeypatch.setenv("KVQ_EARLY_ADAM", early)
        model = _build(torch.bfloat16).train()
        eng = TrainEngine(model, lr=2e-3, milestones=[2], gamma=0.5, seed=17)
        eng.use_graph = use_graph
        assert eng._early_adam == (early == "1")
        losses = [float(eng.train_step(*batches[i % 2])["loss_recon"]) for i in range(6)]
        runs.append((losses, eng.flat.master.clone(), nnops.read_step_state(eng._state)))
    for losses, p, st in runs[1:]:
        np.testing.assert_allclose(np.array(losses), np.array(runs[0][0]), rtol=2e-2, atol=2e-3)
        assert st == runs[0][2] and st[0] == 6
        cos = torch.nn.functional.cosine_similarity((p - p.mean()).double(), (runs[0][1] - runs[0][1].mean()).double(), dim=0).item()
        assert cos > 0.9999, cos


def test_engine_step_count_setter_moves_the_device_state():
    from kvq import nnops
    from kvq.engine import TrainEngine
    model = _build(torch.bfloat16).train()
    eng = TrainEngine(model, lr=1e-3, milestones=[10], gamma=0.1)
    eng.step_count = 41                                   # e.g. resuming a run
    eng.train_step(*_batch(B=4, S=16))
    step, lr, bc1, _ = nnops.read_step_state(eng._state)
    assert step == 42 and eng.step_count == 42
    np.testing.assert_allclose(lr, 1e-4, rtol=1e-6)
    np.testing.assert_allclose(bc1, 1 - 0.9 ** 42, rtol=2e-5)


def test_engine_survives_a_failed_graph_capture(monkeypatch):
    """Capture is an optimisation: an error inside it leaves the engine on the eager path with the same results."""
    from kvq import engine as E
    monkeypatch.setenv("KVQ_GRAPH_STRICT", "0")           # (tests/conftest.py makes a failed capture fatal: this test is ABOUT the fallback)
    model = _build(torch.bfloat16).train()
    eng = E.TrainEngine(model, lr=1e-3, seed=5)
    ids, mask = _batch(B=8, S=16, seed=2)
    calls = {"n": 0}
    real = E.TrainEngine._emb_bwd

    def flaky(self, *a, **k):
        if self._cap is not None:
            calls["n"] += 1
            raise RuntimeError("injected failure during capture")
        return real(self, *a, **k)
    monkeypatch.setattr(E.TrainEngine, "_emb_bwd", flaky)
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(5)]
    assert calls["n"] == 1 and eng.use_graph is False and not eng._graphs and eng.step_count == 5
    monkeypatch.setattr(E.TrainEngine, "_emb_bwd", real)
    ref = E.TrainEngine(_build(torch.bfloat16).train(), lr=1e-3, seed=5)
    ref.use_graph = False
    want = [float(ref.train_step(ids, mask)["loss_recon"]) for _ in range(5)]
    np.testing.assert_allclose(losses, want, rtol=2e-2)


def test_engine_graphs_per_shape_interleaved_with_eval():
    """Two batch shapes (two captured step chains), evaluation steps in between, a return to the first shape: same trajectory
    as the engine that launches every kernel eagerly."""
    from kvq.engine import TrainEngine
    a, b = _batch(B=8, S=16, seed=11), _batch(B=4, S=32, seed=12)
    plan = [("t", a)] * 3 + [("e", b)] + [("t", b)] * 3 + [("e", a)] + [("t", a)] * 2 + [("t", b)]
    runs = []
    for use_graph in (False, True):
        model = _build(torch.bfloat16)
        eng = TrainEngine(model, lr=1e-3, seed=3)
        eng.use_graph = use_graph
        out = []
        for kind, (ids, mask) in plan:
            if kind == "t":
                model.train()
                out.append(float(eng.train_step(ids, mask)["loss_recon"]))
            else:
                model.eval()
                out.append(float(eng.eval_step(ids, mask)["loss_recon"]))
        assert len(eng._graphs) == (2 if use_graph else 0) and eng.step_count == 9
        runs.append(out)
    np.testing.assert_allclose(runs[0], runs[1], rtol=2e-2, atol=2e-3)


def test_hf_forward_kvq_path_and_engine_agree_on_gpu():
    """The chain that pins the composed step: HuggingFace's own BERT forward (the third-party part of the reference, Bagon.py:46-55)
    -> the kvq path of the same modules (HIP LayerNorm / attention / GELU kernels) -> the TrainEngine, all on the GPU in f32 eval
    mode on one model: same logits, same reconstruction loss."""
    from kvq.engine import TrainEngine
    from kvq.functional import fused_cross_entropy
    model = _build(torch.float32).eval()
    ids, mask = _batch(B=5, S=12, seed=21)
    with torch.no_grad():
        model.backend = "hf"
        vq_hf, perp_hf, idx_hf, logits_hf = model(ids, mask)
        model.backend = "kvq"
        vq_kv, perp_kv, idx_kv, logits_kv = model(ids, mask)
    assert torch.equal(idx_hf, idx_kv)
    torch.testing.assert_close(logits_kv, logits_hf, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(vq_kv.item(), vq_hf.item(), rtol=1e-5)
    loss_hf, _, pred_hf = fused_cross_entropy(logits_hf.reshape(-1, logits_hf.shape[-1]).contiguous(), ids.reshape(-1))
    eng = TrainEngine(model)
    out = eng.forward_backward(ids, mask, training=False, compute_grads=False)
    np.testing.assert_allclose(out["loss_recon"].item(), loss_hf.item(), rtol=2e-5)
    np.testing.assert_allclose(out["loss_vq"].item(), vq_hf.item(), rtol=1e-5)
    assert torch.equal(out["indices"], idx_hf) and torch.equal(out["recon_ids"].reshape(-1), pred_hf)


def test_engine_gradients_match_autograd_through_huggingface_forward():
    """Every parameter gradient of the engine's explicit backward against torch autograd through HuggingFace's own forward
    (encoder -> VectorQuantizer -> decoder with cross-attention -> mean token cross-entropy, Trainer.py:94-105), f32, dropout off."""
    import torch.nn.functional as F
    from kvq.engine import TrainEngine
    model = _build(torch.float32).eval()
    ids, mask = _batch(B=6, S=12, seed=5)
    eng = TrainEngine(model, lr=1e-3)
    out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
    mine = {name: eng.flat.g(name).float().clone() for name, p in eng.param_of.items() if p.requires_grad}
    gE_mine = eng.gE.clone()
    for p in model.parameters():
        p.grad = None
    model.backend = "hf"
    vq_loss, _perp, idx, logits = model(ids, mask)
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), ids.reshape(-1)) + vq_loss
    loss.backward()
    model.backend = "kvq"
    np.testing.assert_allclose(out["loss_recon"].item() + out["loss_vq"].item(), loss.item(), rtol=2e-5)
    assert torch.equal(out["indices"], idx)
    worst = 0.0
    for name, p in eng.param_of.items():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        if name.endswith("k.b"):                              # key-bias gradients are rounding noise (softmax shift invariance)
            continue
        ref = p.grad.float()
        got = mine[name]
        if ref.dim() == 2 and got.shape[0] > ref.shape[0]:     # padded LM-head rows
            got = got[: ref.shape[0]]
        torch.testing.assert_close(got, ref, rtol=5e-3, atol=5e-6, msg=lambda m: f"{name}: {m}")
        worst = max(worst, (got - ref).abs().max().item())
    torch.testing.assert_close(gE_mine, model.vector_quantizer.embedding.weight.grad, rtol=2e-3, atol=1e-7)
