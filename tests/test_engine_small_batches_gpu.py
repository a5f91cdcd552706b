"""The bf16 TrainEngine at the row counts of the REFERENCE's own batches -- sentences padded to 12 tokens
(models/shelgon3/Trainer.py:82), 64 / 128 / 512 of them = 768 / 1536 / 6144 rows -- and at odd sizes (72 rows, a 9-code Gumbel
quantiser): every matrix product of the step runs in libkvq.so.  torch.mm / addmm / bmm / matmul / nn.functional.linear are
patched to raise while the engine steps; every launch of the GEMM family is judged against an f32 matmul of the SAME bf16
operands, as tests/test_engine_base_shapes_gpu.py does at 2048 rows."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


from _gemm_guard import forbid_vendor_gemms as _forbid_vendor_gemms  # noqa: E402


def _judged_gemms(monkeypatch, tol=4e-3):
    """Wrap the GEMM entry points of kvq.nnops: every launch is compared with the f32 product of its own operands."""
    from kvq import nnops
    seen = {"nt": 0, "nn": 0, "tn": 0, "gelu": 0, "dgelu": 0, "dropres": 0, "grouped": 0, "any": 0, "tiles": set()}
    worst = [0.0, ""]
    real = dict(gemm=nnops.gemm, gemm_gelu=nnops.gemm_gelu, gemm_dgelu=nnops.gemm_dgelu, gemm_problem=nnops.gemm_problem,
                gemm_grouped=nnops.gemm_grouped, gemm_dropres=nnops.gemm_dropres)
    recorded = {}

    def ref(a, b, layout):                                # (the saved torch.mm: the checker may multiply with the library)
        a, b = a.float(), b.float()
        return _REAL["mm"](a.t() if layout == "tn" else a, b.t() if layout == "nt" else b)

    def judge(out, want, what):
        err = ((out.float() - want).norm() / want.norm().clamp_min(1e-30)).item()
        if err > worst[0]:
            worst[0], worst[1] = err, what
        assert err < tol, f"{what}: relative L2 error {err:.3g} against the f32 product of the same operands"

    def gemm(a, b, layout="nt", bias=None, out=None, accumulate=False, tile=None):
        before = out.float().clone() if accumulate else None
        M, N, K = nnops._gemm_dims(a, b, layout)
        o = real["gemm"](a, b, layout, bias=bias, out=out, accumulate=accumulate, tile=tile)
        want = ref(a, b, layout) + (bias.float() if bias is not None else 0.0)
        if accumulate:
            want = want.to(torch.bfloat16).float() + before
        judge(o, want, f"gemm {layout} [{M}, {N}] K={K} tile={tile} acc={accumulate}")
        seen[layout] += 1
        if nnops.gemm_mfma_ok(a, b, o, layout, bias):
            seen["tiles"].add(nnops.TILE_NAMES[nnops.pick_tile(M, N, K)] if tile is None else tile)
        else:
            seen["any"] += 1
        return o

    def gemm_gelu(x, w, bias, tile="256x192"):
        h, g = real["gemm_gelu"](x, w, bias, tile=tile)
        judge(h, ref(x, w, "nt") + bias.float(), f"gemm_gelu h tile={tile}")
        judge(g, F.gelu(h.float()), f"gemm_gelu gelu(h) tile={tile}")
        seen["gelu"] += 1
        return h, g

    def gemm_dgelu(gy, w, h, tile="256x192"):
        g_h, part = real["gemm_dgelu"](gy, w, h, tile=tile)
        with torch.enable_grad():
            hf = h.float().requires_grad_(True)
            F.gelu(hf).backward(ref(gy, w, "nn").to(torch.bfloat16).float())
        judge(g_h, hf.grad, "gemm_dgelu")
        seen["dgelu"] += 1
        return g_h, part

    def gemm_dropres(x, w, bias, resid, p_drop, seed, site, tile=None):
        # the dense layer in front of a residual LayerNorm with dropout + residual in its epilogue (round 5); judged where the
        # step runs without dropout (training=False), launched unjudged otherwise
        pre = real["gemm_dropres"](x, w, bias, resid, p_drop, seed, site, tile=tile)
        if p_drop == 0.0:
            judge(pre, (ref(x, w, "nt") + bias.float()).to(torch.bfloat16).float() + resid.float(), f"gemm_dropres {tuple(pre.shape)} K={x.shape[1]}")
        seen["dropres"] += 1
        seen["tiles"].add(nnops.TILE_NAMES[nnops.pick_tile(x.shape[0], w.shape[0], x.shape[1], candidates=nnops.DROPRES_TILES)])
        return pre

    def gemm_problem(a, b, out, layout, bias=None, accumulate=False):
        pr = real["gemm_problem"](a, b, out, layout, bias=bias, accumulate=accumulate)
        recorded[out.data_ptr()] = (a, b, out, layout)
        return pr

    def gemm_grouped(problems, layout, tile):
        real["gemm_grouped"](problems, layout, tile)
        for pr in problems:
            if pr.C in recorded:
                a, b, out, lay = recorded.pop(pr.C)
                judge(out, ref(a, b, lay), f"grouped {lay} {tuple(out.shape)} tile={tile}")
                seen["grouped"] += 1

    for k, f in dict(gemm=gemm, gemm_gelu=gemm_gelu, gemm_dgelu=gemm_dgelu, gemm_problem=gemm_problem, gemm_grouped=gemm_grouped,
                     gemm_dropres=gemm_dropres).items():
        monkeypatch.setattr(nnops, k, f)
    return seen, worst, recorded


_REAL = {"mm": torch.mm}


def _shelgon(dtype, name="kvq-bert-base-2l", K=512, seed=0):
    from models.bagon.Bagon import LOCAL_BERT_CONFIGS
    from models.shelgon3.Shelgon import Shelgon
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    H = LOCAL_BERT_CONFIGS[name].get("hidden_size", 768)
    torch.manual_seed(seed)
    vq = VectorQuantizer(K, H, 0.25, vq_codebook_init_values=torch.randn(K, H))
    vq.materialize_min_encodings = False
    model = Shelgon(name, vq, name, None, compute_dtype=dtype).cuda()
    model.set_mode("full")
    return model.eval()


def _batch(B, S, seed, hi=30000):
    from dsentences.synthetic import random_token_batch
    ids, mask = random_token_batch(B, S, torch.Generator().manual_seed(seed), vocab_hi=hi, min_len=3, max_len=S)
    return ids.cuda(), mask.cuda()


@pytest.mark.parametrize("B", [64, 128, 512])
def test_engine_bf16_at_the_reference_batch_shapes_runs_no_vendor_gemm(B, monkeypatch):
    """kvq-bert-base-2l, bf16, S = 12, B sentences: forward + backward with every torch GEMM entry point patched to raise; every
    own GEMM launch within 4e-3 (relative L2) of the f32 product of its operands; the loss within bf16 tolerance of f32 autograd
    through HuggingFace's forward (computed BEFORE the patch: that oracle multiplies with torch)."""
    from kvq.engine import TrainEngine
    ids, mask = _batch(B, 12, seed=B)
    m32 = _shelgon(torch.float32)
    m32.backend = "hf"
    with torch.no_grad():
        _, _, _, logits = m32(ids, mask)
        ref_loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), ids.reshape(-1)).item()
    del m32, logits
    model = _shelgon(torch.bfloat16)
    eng = TrainEngine(model, lr=1e-4)
    seen, worst, recorded = _judged_gemms(monkeypatch)
    _forbid_vendor_gemms(monkeypatch)
    out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
    torch.cuda.synchronize()
    assert not recorded and np.isfinite(out["loss_recon"].item())
    np.testing.assert_allclose(out["loss_recon"].item(), ref_loss, rtol=2e-2)
    print(f"B={B}: own GEMM launches inside the step:", {k: v for k, v in seen.items() if k != "tiles"}, "tiles", sorted(seen["tiles"]),
          "worst", worst)
    assert seen["nt"] >= 8 and seen["dropres"] == 10 and seen["nn"] >= 14 and seen["tn"] >= 1 and seen["grouped"] >= 20 and seen["any"] == 0
    if B <= 128:
        assert "64x128" in seen["tiles"]                       # the small tile is what these row counts run on


def test_engine_bf16_train_steps_replayed_without_vendor_gemm(monkeypatch):
    """Six optimiser steps at S = 12, B = 128 (eager, capture, replay) with the torch GEMM entry points patched to raise: the loss
    falls and the replayed steps run from hipGraphs."""
    from kvq.engine import TrainEngine
    model = _shelgon(torch.bfloat16).train()
    eng = TrainEngine(model, lr=2e-4)
    ids, mask = _batch(128, 12, seed=5)
    _forbid_vendor_gemms(monkeypatch)
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and eng._graphs, losses


@pytest.mark.parametrize("kind", ["vq72", "vq66", "gumbel9", "bagon72"])
def test_engine_bf16_odd_sizes_run_on_the_any_shape_kernel(kind, monkeypatch):
    """Sizes the MFMA kernel refuses still never reach torch.  6 x 12 = 72 tokens: the weight gradients contract over 72 rows,
    zero-padded to 128 since round 5 (nnops.tn_operands_k64) -- the step stays on the MFMA kernels.  6 x 11 = 66 tokens: 64 rows of
    every forward / input-gradient product on the MFMA kernel, the last 2 on csrc/kvq_gemm_any.hip.  9 Gumbel codes (the reference
    analysis' N_E = 9, unsupervised_vq_disentanglement.py:58): the any-shape kernel takes the products over the codes."""
    from kvq.engine import TrainEngine
    if kind == "gumbel9":
        from models.shelgon3.GumbelQuantizer import GumbelQuantizer
        from models.shelgon3.Shelgon import Shelgon
        torch.manual_seed(0)
        gq = GumbelQuantizer(enc_out_size=128, n_embed=9, embedding_dim=128, temperature=0.9, kl_div_scale=5e-4, straight_through=True)
        model = Shelgon("kvq-bert-tiny", gq, "kvq-bert-tiny", None, compute_dtype=torch.bfloat16).cuda().eval()
    elif kind == "bagon72":
        from models.bagon.Bagon import Bagon
        torch.manual_seed(0)
        model = Bagon("kvq-bert-tiny", "kvq-bert-tiny", True, compute_dtype=torch.bfloat16).cuda().eval()
    else:
        model = _shelgon(torch.bfloat16, "kvq-bert-tiny", K=32)
    eng = TrainEngine(model, lr=1e-3)
    from kvq import nnops
    ids, mask = _batch(6, 11 if kind == "vq66" else 12, seed=2, hi=2000)
    with monkeypatch.context() as mp:                     # (the judges read results back: not inside a capture)
        seen, worst, recorded = _judged_gemms(mp, tol=6e-3)
        _forbid_vendor_gemms(mp)
        before = dict(nnops.GEMM_ROUTES)
        out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
        torch.cuda.synchronize()
        routes = {k: v - before[k] for k, v in nnops.GEMM_ROUTES.items()}
    assert np.isfinite(out["loss_recon"].item()), out
    if kind in ("vq72", "bagon72"):
        assert routes["any"] == 0 and seen["grouped"] >= 8, (routes, seen)
    elif kind == "vq66":
        assert routes["row_split"] >= 8 and routes["any"] == routes["row_split"], routes
    else:
        assert routes["any"] >= 4, routes
    _forbid_vendor_gemms(monkeypatch)
    monkeypatch.setenv("KVQ_GRAPH_STRICT", "1")
    model.train()
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(5)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and eng._graphs, losses


@pytest.mark.parametrize("B", [100, 37])
def test_engine_bf16_token_counts_that_miss_the_k_tile_stay_on_the_mfma_kernels(B, monkeypatch, recwarn):
    """B = 100 sentences x 12 tokens = 1200 tokens (1200 % 64 = 48: the last batch of an epoch, any odd batch size): the
    weight gradients contract over a token count that is not a multiple of the MFMA k-tile -- their operands are zero-padded to
    1216 rows and stay on the MFMA kernels (single launches and the grouped queue alike).  B = 37 -> 444 tokens (444 % 8 = 4): the
    forward / input-gradient products run 440 rows on the MFMA kernel and 4 on the any-shape kernel.  No product above 64 MFLOP
    reaches the any-shape kernel (it would warn), no vendor GEMM, every launch judged against f32."""
    import warnings
    from kvq import nnops
    from kvq.engine import TrainEngine
    ids, mask = _batch(B, 12, seed=B)
    m32 = _shelgon(torch.float32)
    m32.backend = "hf"
    with torch.no_grad():
        _, _, _, logits = m32(ids, mask)
        ref_loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), ids.reshape(-1)).item()
    del m32, logits
    model = _shelgon(torch.bfloat16)
    eng = TrainEngine(model, lr=1e-4)
    with monkeypatch.context() as mp:                     # (the judges read results back: not inside a capture)
        seen, worst, recorded = _judged_gemms(mp)
        _forbid_vendor_gemms(mp)
        before = dict(nnops.GEMM_ROUTES)
        out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
        torch.cuda.synchronize()
        routes = {k: v - before[k] for k, v in nnops.GEMM_ROUTES.items()}
    _forbid_vendor_gemms(monkeypatch)
    monkeypatch.setenv("KVQ_GRAPH_STRICT", "1")
    print(f"B={B}: routes {routes}; judged", {k: v for k, v in seen.items() if k != "tiles"}, "worst", worst)
    assert not recorded and np.isfinite(out["loss_recon"].item())
    np.testing.assert_allclose(out["loss_recon"].item(), ref_loss, rtol=2e-2)
    assert not [w for w in recwarn.list if "any-shape kernel" in str(w.message)], [str(w.message) for w in recwarn.list]
    assert seen["grouped"] >= 20                                   # the layers' weight gradients: grouped MFMA launches, padded
    if B == 100:
        assert routes["any"] == 0 and routes["tn_padded"] + routes["mfma"] >= 30, routes
    else:
        assert routes["row_split"] >= 20 and routes["any"] == routes["row_split"], routes     # only the 4-row remainders
    model.train()
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(5)]                # eager, capture, replay
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and eng._graphs, losses
