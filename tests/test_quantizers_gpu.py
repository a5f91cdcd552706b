"""The quantiser variants behind Shelgon on the HIP path: the 9-factor MultiVectorQuantizer and the EMA codebook update
(extensions named by BASELINE.json, SURVEY.md section 8 row A9: oracle = G / one calls of the CPU restatement), the
GumbelQuantizer inside the TrainEngine (models/shelgon3/GumbelQuantizer.py:43-83), and Bagon / Shelgon.forward routed through
the engine's kernels (models/bagon/Bagon.py:40-55, models/shelgon3/Shelgon.py:50-73)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _build():
    O.build()


def _batch(B=6, S=12, seed=1, vocab=2000):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, vocab, (B, S), generator=g)
    lens = torch.randint(3, S + 1, (B,), generator=g)
    ids = ids * (torch.arange(S)[None] < lens[:, None])
    return ids.cuda(), (ids != 0).long().cuda()


@pytest.mark.parametrize("D", [576, 768])
def test_multi_vector_quantizer_equals_one_oracle_call_per_factor(D):
    """9 factors on 576 columns (9 x 64) and on bert-base's 768 (slices of 86 / 85 columns, zero-padded to 96 inside)."""
    from models.shelgon3.MultiVectorQuantizer import MultiVectorQuantizer
    G, K, B, S = 9, 32, 4, 7
    torch.manual_seed(0)
    mq = MultiVectorQuantizer(G, K, D, 0.25).cuda()
    W = mq.embedding.weight.data
    W.copy_(torch.randn_like(W) * (W != 0 if mq.ragged else 1))          # pad columns of the codebooks stay zero
    z = torch.randn(B, S, D, device="cuda", requires_grad=True)
    loss, z_q, perp, enc, idx = mq(z, "cuda")
    assert enc is None and idx.shape == (B, S, G) and z_q.shape == z.shape
    (loss * 1.7 + (z_q * torch.arange(D, device="cuda").float()).sum()).backward()
    base, rem = divmod(D, G)
    widths = [base + (g < rem) for g in range(G)]
    offs = np.concatenate([[0], np.cumsum(widths)])
    Ep = mq.embedding.weight.detach().cpu().numpy().reshape(G, K, mq.d_factor)
    gEp = mq.embedding.weight.grad.cpu().numpy().reshape(G, K, mq.d_factor)
    zn = z.detach().cpu().numpy().reshape(B * S, D)
    g_up = np.broadcast_to(np.arange(D, dtype=np.float32), (B * S, D))
    sse, perps = 0.0, []
    for g in range(G):
        sl = slice(offs[g], offs[g + 1])
        w = widths[g]
        zs, Eg = np.ascontiguousarray(zn[:, sl]), np.ascontiguousarray(Ep[g][:, :w])
        ora = O.vq_forward(zs, Eg, 0.25)
        assert np.array_equal(idx[..., g].reshape(-1).cpu().numpy(), ora["idx"]), f"factor {g}"
        assert np.array_equal(z_q.detach().reshape(B * S, D)[:, sl].cpu().numpy(), ora["z_q"])
        sse += ora["loss"] * w
        perps.append(ora["perplexity"])
        # d total / d loss_g' = 1.7 * w / D for the oracle's per-slice mean
        gz, gE = O.vq_backward(zs, Eg, ora["idx"], np.ascontiguousarray(g_up[:, sl]), 1.7 * w / D, 0.25)
        np.testing.assert_allclose(z.grad.reshape(B * S, D)[:, sl].cpu().numpy(), gz, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(gEp[g][:, :w], gE, rtol=2e-5, atol=1e-7)
        assert not gEp[g][:, w:].any() and not Ep[g][:, w:].any()         # pad columns: zero value, zero gradient
    np.testing.assert_allclose(loss.item(), sse / D, rtol=2e-6)
    np.testing.assert_allclose(perp.item(), np.mean(perps), rtol=1e-5)


def _model(vq, name="kvq-bert-tiny-nodrop", dtype=torch.float32):
    from models.shelgon3.Shelgon import Shelgon
    torch.manual_seed(0)
    return Shelgon(name, vq, name, None, compute_dtype=dtype).cuda()


def _compare_engine_with_autograd(model, ids, mask, eng, skip=()):
    for p in model.parameters():
        p.grad = None
    l_vq, perp, idx, l_rec, acc, recon = model.forward_loss(ids, mask)
    (l_rec + l_vq).backward()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    out = eng.forward_backward(ids, mask, training=model.training, compute_grads=True)
    np.testing.assert_allclose(out["loss_recon"].item(), l_rec.item(), rtol=2e-5)
    np.testing.assert_allclose(out["loss_vq"].item(), l_vq.item(), rtol=2e-5)
    assert torch.equal(out["indices"].reshape(-1), idx.reshape(-1))
    name_of = {id(p): n for n, p in model.named_parameters()}
    for en, p in eng.param_of.items():
        n = name_of[id(p)]
        if p.requires_grad and not n.endswith("key.bias") and n not in skip:
            torch.testing.assert_close(eng.flat.g(en).float()[: p.shape[0]] if p.dim() else eng.flat.g(en).float(), ref[n].float(),
                                       rtol=5e-3, atol=5e-6, msg=lambda m: f"{n}: {m}")
    return ref, out


@pytest.mark.parametrize("name,D", [("kvq-bert-9x64", 576), ("kvq-bert-tiny-nodrop", 128)])
def test_engine_runs_the_nine_factor_quantiser_as_one_grouped_launch(name, D):
    """Nine codebooks of 32 codes: hidden 576 = nine 64-wide slices, and hidden 128 = ragged slices (15 / 14 columns, padded to 32);
    engine == autograd through the module."""
    from kvq.engine import TrainEngine
    from models.shelgon3.MultiVectorQuantizer import MultiVectorQuantizer
    mq = MultiVectorQuantizer(9, 32, D, 0.25)
    W = mq.embedding.weight.data
    W.copy_(torch.randn_like(W) * (W != 0 if mq.ragged else 1))
    model = _model(mq, name).train()
    ids, mask = _batch(B=5, S=12, seed=3)
    eng = TrainEngine(model, lr=1e-3)
    assert eng.G == 9 and TrainEngine.supports(model, 12)
    ref, out = _compare_engine_with_autograd(model, ids, mask, eng)
    assert out["indices"].shape == (5, 12, 9)
    torch.testing.assert_close(eng.gE, ref["vector_quantizer.embedding.weight"], rtol=2e-3, atol=1e-7)
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(6)]           # eager steps, then the captured replay
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and eng._graphs


@pytest.mark.parametrize("multi", [False, True])
def test_ema_codebook_in_module_and_engine(multi):
    """ema_decay: no codebook gradient; after one training step the codebook is the textbook EMA update (oracle) of the encoder
    outputs of that step -- same result from the autograd path (module) and from the TrainEngine."""
    from kvq.engine import TrainEngine
    from models.shelgon3.MultiVectorQuantizer import MultiVectorQuantizer
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    torch.manual_seed(1)
    G = 4 if multi else 1
    vq = MultiVectorQuantizer(4, 16, 128, 0.25, ema_decay=0.9) if multi else VectorQuantizer(16, 128, 0.25, torch.randn(16, 128), ema_decay=0.9)
    if multi:
        vq.embedding.weight.data.normal_()
        vq.ema_m.copy_(vq.embedding.weight.data.view(4, 16, 32))
    model = _model(vq).train()
    ref_model = copy.deepcopy(model)
    ids, mask = _batch(B=8, S=16, seed=5)
    assert not model.vector_quantizer.embedding.weight.requires_grad
    E0 = model.vector_quantizer.embedding.weight.detach().clone()
    # autograd path: the module updates the codebook inside forward (training mode)
    with torch.no_grad():
        z = ref_model.encode(ids, mask)
    l_vq, _perp, idx, l_rec, _acc, _recon = ref_model.forward_loss(ids, mask)
    E_mod = ref_model.vector_quantizer.embedding.weight.detach().clone()
    # oracle on the same encoder outputs
    zn = z.float().cpu().numpy().reshape(-1, G, 128 // G)
    for g in range(G):
        zg = np.ascontiguousarray(zn[:, g])
        Eg = E0.cpu().numpy().reshape(G, 16, -1)[g]
        ig = idx.reshape(-1, G)[:, g].cpu().numpy() if multi else idx.reshape(-1).cpu().numpy()
        _n, _m, E_or = O.vq_ema_update(zg, ig, 0.9, 1e-5, np.ones(16, np.float32), Eg.copy(), Eg.copy())
        np.testing.assert_allclose(E_mod.cpu().numpy().reshape(G, 16, -1)[g], E_or, rtol=1e-5, atol=1e-6)
    # engine: same codebook after one optimiser step (the step does not touch E through Adam)
    eng = TrainEngine(model, lr=1e-3)
    eng.train_step(ids, mask)
    torch.testing.assert_close(model.vector_quantizer.embedding.weight.detach(), E_mod, rtol=1e-5, atol=1e-6)
    assert not torch.equal(E_mod, E0)
    for _ in range(4):                                   # through the captured replay as well: the codebook keeps moving
        eng.train_step(ids, mask)
    assert eng._graphs and not torch.equal(model.vector_quantizer.embedding.weight.detach(), E_mod)


def test_engine_trains_the_gumbel_quantiser():
    """GumbelQuantizer inside the TrainEngine: with the same Gumbel noise, losses, codes and every gradient (BERT, proj, embed)
    equal torch autograd through the module (GumbelQuantizer.py:43-83 between the two BERT stacks)."""
    from kvq.engine import TrainEngine
    from models.shelgon3.GumbelQuantizer import GumbelQuantizer
    torch.manual_seed(2)
    gq = GumbelQuantizer(128, 24, 128, temperature=0.7, kl_div_scale=5e-2, straight_through=True)
    model = _model(gq).train()
    ids, mask = _batch(B=6, S=12, seed=7)
    N = ids.numel()
    u = torch.rand(N, 24, device="cuda").clamp_(1e-9, 1 - 1e-9)
    noise = -torch.log(-torch.log(u))
    # autograd path with the noise handed to the module
    for p in model.parameters():
        p.grad = None
    z = model.encode(ids, mask)
    z_q, diff, ind = model.vector_quantizer(z, True, noise=noise)
    hidden = model.decode_hidden(z_q.to(z.dtype), ids, mask)
    from kvq import bert as kbert
    from kvq.functional import fused_cross_entropy
    l_rec, _acc, _pred = fused_cross_entropy(kbert.lm_head_logits(model.decoder, hidden, model.compute_dtype), ids)
    (l_rec + diff).backward()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    eng = TrainEngine(model, lr=1e-3)
    eng.gumbel_noise = noise
    out = eng.forward_backward(ids, mask, training=True, compute_grads=True)
    assert torch.equal(out["indices"].reshape(-1), ind.reshape(-1)) and out["indices"].shape == ids.shape
    np.testing.assert_allclose(out["loss_recon"].item(), l_rec.item(), rtol=2e-5)
    np.testing.assert_allclose(out["loss_vq"].item(), diff.item(), rtol=2e-5)
    assert out["perplexity"].item() == torch.unique(ind).numel()
    torch.testing.assert_close(eng.g_pw, ref["vector_quantizer.proj.weight"], rtol=5e-3, atol=1e-6)
    torch.testing.assert_close(eng.g_pb, ref["vector_quantizer.proj.bias"], rtol=5e-3, atol=1e-6)
    torch.testing.assert_close(eng.g_emb, ref["vector_quantizer.embed.weight"], rtol=5e-3, atol=1e-6)
    name_of = {id(p): n for n, p in model.named_parameters()}
    for en, p in eng.param_of.items():
        n = name_of[id(p)]
        if not n.endswith("key.bias"):
            torch.testing.assert_close(eng.flat.g(en).float(), ref[n].float(), rtol=5e-3, atol=5e-6, msg=lambda m: f"{n}: {m}")
    eng.gumbel_noise = None
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and eng._graphs


def test_gumbel_products_on_the_own_gemm_at_step_sizes(monkeypatch):
    """bert-base widths, 2048 tokens, 512 codes: the six products of the Gumbel mode (logits, z_q and their four gradient GEMMs)
    run on csrc/kvq_gemm2.hip and agree with the same step on the library (KVQ_OWN_GEMM=0 routes every GEMM to torch): same Gumbel
    noise, nine tenths of the codes equal, losses within 1e-2 relative, quantiser gradients cosine > 0.99 (both paths round to
    bf16 with f32 accumulation; the accumulation order differs)."""
    from dsentences.synthetic import random_token_batch
    from kvq.engine import TrainEngine
    from models.shelgon3.GumbelQuantizer import GumbelQuantizer
    from models.shelgon3.Shelgon import Shelgon
    ids, mask = (t.cuda() for t in random_token_batch(64, 32, torch.Generator().manual_seed(8)))
    u = torch.rand(ids.numel(), 512, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).clamp_(1e-9, 1 - 1e-9)
    noise = -torch.log(-torch.log(u))
    runs = {}
    for own in ("1", "0"):
        monkeypatch.setenv("KVQ_OWN_GEMM", own)
        torch.manual_seed(3)
        gq = GumbelQuantizer(768, 512, 768, temperature=0.9, kl_div_scale=5e-4, straight_through=True)
        model = Shelgon("kvq-bert-base-2l", gq, "kvq-bert-base-2l", None, compute_dtype=torch.bfloat16).cuda().train()
        eng = TrainEngine(model, lr=1e-4)
        eng.gumbel_noise = noise
        out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
        assert eng._gumbel_own == (6 if own == "1" else 0)
        runs[own] = (out["loss_recon"].item(), out["loss_vq"].item(), out["indices"].clone(), eng.g_pw.float().clone(),
                     eng.g_emb.float().clone(), eng.flat.g("enc.1.f2.w").float().clone())
    a, b = runs["1"], runs["0"]
    np.testing.assert_allclose(a[0], b[0], rtol=1e-2)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-2)
    assert (a[2] == b[2]).float().mean().item() > 0.9
    for x, y in zip(a[3:], b[3:]):
        assert F.cosine_similarity(x.reshape(-1), y.reshape(-1), dim=0).item() > 0.99


def test_bagon_forward_without_autograd_runs_on_the_engine_and_matches_huggingface():
    """Bagon.forward(enc_ids, enc_mask, dec_ids, dec_mask) -> logits (models/bagon/Bagon.py:40-55), decoder input != encoder input."""
    from kvq import engine as E
    from models.bagon.Bagon import Bagon
    torch.manual_seed(3)
    model = Bagon("kvq-bert-tiny", "kvq-bert-tiny", compute_dtype=torch.float32).cuda().eval()
    enc_ids, enc_mask = _batch(B=5, S=12, seed=11)
    dec_ids, dec_mask = _batch(B=5, S=9, seed=12)
    with torch.no_grad():
        model.backend = "hf"
        want = model(enc_ids, enc_mask, dec_ids, dec_mask)
        model.backend = "kvq"
        assert E.engine_of(model, create=False) is None
        got = model(enc_ids, enc_mask, dec_ids, dec_mask)
        assert E.engine_of(model, create=False) is not None          # the call went through the TrainEngine's schedule
    assert got.shape == want.shape == (5, 9, 2048)
    torch.testing.assert_close(got.float(), want, rtol=2e-4, atol=2e-4)
    # parameters changed from outside (e.g. an optimiser of the autograd path): the bf16 shadow follows
    mb = Bagon("kvq-bert-tiny", "kvq-bert-tiny", compute_dtype=torch.bfloat16).cuda().eval()
    with torch.no_grad():
        a = mb(enc_ids, enc_mask, enc_ids, enc_mask)
        mb.decoder.cls.predictions.bias.add_(1.0)
        b = mb(enc_ids, enc_mask, enc_ids, enc_mask)
    assert (b.float() - a.float()).mean().item() > 0.9


def test_engine_forward_sees_a_codebook_moved_by_the_module_ema_update():
    """The autograd path with `ema_decay` moves the codebook through a raw pointer inside the module's forward (no tensor version
    bump).  A following no_grad Shelgon.forward runs on the engine, which keeps a fragment-ordered copy of the codebook: the copy
    must follow (VectorQuantizer.codebook_epoch; forward_logits repacks in any case) -- indices / loss / logits of the engine
    forward equal the module path on the moved codebook."""
    from kvq.engine import engine_of
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    torch.manual_seed(3)
    vq = VectorQuantizer(16, 128, 0.25, torch.randn(16, 128) * 0.5, ema_decay=0.5)
    model = _model(vq).train()
    ids, mask = _batch(B=8, S=16, seed=9)
    with torch.no_grad():
        model.eval()
        model(ids, mask, "cuda", False)                        # builds the engine and its codebook pack from the INITIAL codebook
        model.train()
    eng = engine_of(model, create=False)
    assert eng is not None and eng._epack is not None
    E0 = vq.embedding.weight.detach().clone()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    for _ in range(2):                                          # two autograd steps: the module's forward moves the codebook
        l_vq, _perp, _idx, l_rec, _acc, _recon = model.forward_loss(ids, mask)
        opt.zero_grad(); (l_rec + l_vq).backward(); opt.step()
    assert (vq.embedding.weight.detach() - E0).abs().max().item() > 1e-3 and vq.codebook_epoch == 2
    model.eval()
    with torch.no_grad():
        l_e, p_e, idx_e, logits_e = model(ids, mask, "cuda", False)             # engine
        z = model.encode(ids, mask)
        l_m, zq_m, p_m, _enc, idx_m = vq.forward(z.contiguous(), "cuda")        # module path on the same encoder output
    assert torch.equal(idx_e.reshape(-1), idx_m.reshape(-1))
    torch.testing.assert_close(l_e.float(), l_m.float(), rtol=1e-5, atol=1e-7)
    # and the stamp alone (without forward_logits' unconditional repack) notices the move
    eng._repack_codebook()
    stamp = eng._E_version
    vq.ema_update(z.reshape(-1, 128).detach(), idx_m.reshape(-1))
    assert eng._codebook_stamp() != stamp
