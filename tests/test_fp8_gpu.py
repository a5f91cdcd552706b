"""fp8 (OCP e4m3fn) forward GEMMs -- the extension named by BASELINE.json configs[4] (the reference is f32 throughout):
quantisation kernels against torch's float8_e4m3fn conversion, the fp8 MFMA GEMM against an f32 matmul of the dequantised
operands, and the TrainEngine with fp8 forward GEMMs against the bf16 engine (loss parity; backward stays bf16)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_quant(x):
    amax = x.float().abs().max()
    scale = 448.0 / amax
    q = (x.float() * scale).clamp(-448, 448).cpu().to(torch.float8_e4m3fn)
    return q, scale.item()


def test_quantize_matches_torch_float8_conversion():
    from kvq import nnops
    torch.manual_seed(0)
    big = (torch.randn(300, 520, device="cuda") * 3).to(torch.bfloat16)
    x = big[:, 8:8 + 504]                                   # row stride 520, 16-byte aligned start
    q, scale = nnops.fp8_quantize(x)
    want, s = _ref_quant(x)
    np.testing.assert_allclose(scale.item(), s, rtol=1e-6)
    got = q.cpu().view(torch.float8_e4m3fn).float()
    assert torch.equal(got, want.float())
    assert got.abs().max().item() == 448.0                  # the largest element lands exactly on the largest e4m3 value


@pytest.mark.parametrize("shape", [(512, 768, 256), (1000, 776, 384), (8192, 768, 768), (2048, 3072, 768), (40, 24, 128)])
def test_fp8_gemm_equals_f32_matmul_of_the_dequantised_operands(shape):
    from kvq import nnops
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
    x8, sx = nnops.fp8_quantize(x)
    w8, sw = nnops.fp8_quantize(w)
    out = nnops.gemm_fp8_nt(x8, w8, sx, sw, bias=bias)
    xf = x8.cpu().view(torch.float8_e4m3fn).float().cuda()
    wf = w8.cpu().view(torch.float8_e4m3fn).float().cuda()
    ref = (xf @ wf.t()) / (sx * sw) + bias.float()
    err = (out.float() - ref).abs().max().item()
    assert err <= 2.0 ** -8 * ref.abs().max().item() + 1e-3, err        # only the bf16 rounding of the result
    # and the quantisation itself stays within e4m3's resolution of the bf16 product
    true = x.float() @ w.float().t() + bias.float()
    rel = (out.float() - true).norm().item() / true.norm().item()
    assert rel < 5e-2, rel


def _build(seed=0):
    from models.shelgon3.Shelgon import Shelgon
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    torch.manual_seed(seed)
    vq = VectorQuantizer(512, 768, 0.25, vq_codebook_init_values=torch.randn(512, 768))
    vq.materialize_min_encodings = False
    return Shelgon("kvq-bert-base-2l", vq, "kvq-bert-base-2l", None, compute_dtype=torch.bfloat16).cuda().eval()


@pytest.mark.parametrize("scope", ["all", True])
def test_engine_fp8_forward_against_the_bf16_engine(scope):
    """bert-base widths, 2 layers, 2048 tokens: forward GEMMs on the fp8 matrix cores -- every one ("all") or, the default of
    fp8_forward=True, the two that pay for their quantisation pass (LM head, all-layer cross-K/V) --, backward in bf16.
    Stated tolerance: reconstruction loss within 2e-2 relative of the bf16 engine, VQ loss within 5e-2; gradients point the
    same way (cosine > 0.9 per tensor, > 0.98 on average)."""
    from dsentences.synthetic import random_token_batch
    from kvq.engine import TrainEngine
    ids, mask = (t.cuda() for t in random_token_batch(64, 32, torch.Generator().manual_seed(4)))
    runs = {}
    for fp8 in (False, True):
        model = _build()
        eng = TrainEngine(model, lr=1e-4, fp8_forward=scope if fp8 else False)
        out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
        grads = {n: eng.flat.g(n).float().clone() for n, p in eng.param_of.items() if p.requires_grad}
        runs[fp8] = (out["loss_recon"].item(), out["loss_vq"].item(), out["indices"].clone(), grads, eng)
    (l0, v0, i0, g0, _), (l1, v1, i1, g1, e8) = runs[False], runs[True]
    np.testing.assert_allclose(l1, l0, rtol=2e-2)
    np.testing.assert_allclose(v1, v0, rtol=5e-2)
    assert (i0 == i1).float().mean().item() > 0.9
    cos = [F.cosine_similarity(g1[n].reshape(-1), g0[n].reshape(-1), dim=0).item() for n in g0 if g0[n].norm() > 0 and not n.endswith("k.b")]
    assert min(cos) > 0.9 and np.mean(cos) > 0.98, (min(cos), np.mean(cos))
    # the weight mirror: every segment is the torch conversion of the bf16 shadow at that segment's own scale
    # (two decoder layers: the all-layer cross-K/V block is 3072 rows, below the width from which fp8 pays; at 12 layers it is 18432)
    assert ("enc.0.f1.w" in e8._w8_index) == (scope == "all") and "dec.emb.word" in e8._w8_index
    assert (e8._cakv_w[0] in e8._w8_index) == (scope == "all")
    key = "enc.0.f1.w" if scope == "all" else "dec.emb.word"
    si, (o, n, shape) = e8._w8_index[key], e8.flat.seg[key]
    n = e8._w8_n[si].item()
    want, s = _ref_quant(e8.flat.shadow[o:o + n])
    np.testing.assert_allclose(e8._w8_scale[si].item(), s, rtol=1e-6)
    assert torch.equal(e8._w8[o:o + n].cpu().view(torch.float8_e4m3fn).float(), want.float())


def test_engine_fp8_trains_through_graph_replay():
    from dsentences.synthetic import random_token_batch
    from kvq.engine import TrainEngine
    model = _build(1).train()
    eng = TrainEngine(model, lr=2e-4, fp8_forward=True)
    ids, mask = (t.cuda() for t in random_token_batch(64, 32, torch.Generator().manual_seed(5)))
    losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and eng._graphs, losses


def test_fp8_forward_with_the_nine_factor_quantiser_at_bert_base_widths():
    """BASELINE.json configs[4] on one GPU: fp8 forward GEMMs AND nine codebooks over ragged slices of the 768 columns (86 / 85 wide),
    2 layers, 2048 tokens, against the bf16 engine on the same weights and batch.  Stated tolerance: reconstruction loss within
    2e-2 relative, VQ loss within 5e-2, nine tenths of the [B, S, 9] code indices equal, codebook-gradient cosine > 0.98; then
    eight steps (eager, capture, replay) that lower the loss."""
    from dsentences.synthetic import random_token_batch
    from kvq.engine import TrainEngine
    from models.shelgon3.MultiVectorQuantizer import MultiVectorQuantizer
    from models.shelgon3.Shelgon import Shelgon
    ids, mask = (t.cuda() for t in random_token_batch(64, 32, torch.Generator().manual_seed(6)))

    def build():
        torch.manual_seed(2)
        mq = MultiVectorQuantizer(n_factors=9, n_e=512, e_dim=768, beta=0.25)
        W = mq.embedding.weight.data
        W.copy_(torch.randn_like(W) * (W != 0 if mq.ragged else 1))
        return Shelgon("kvq-bert-base-2l", mq, "kvq-bert-base-2l", None, compute_dtype=torch.bfloat16).cuda().eval()

    runs = {}
    for fp8 in (False, True):
        eng = TrainEngine(build(), lr=2e-4, fp8_forward=fp8)
        assert eng.G == 9
        out = eng.forward_backward(ids, mask, training=False, compute_grads=True)
        runs[fp8] = (out["loss_recon"].item(), out["loss_vq"].item(), out["indices"].clone(), eng.gE.float().clone(), eng)
    (l0, v0, i0, g0, _), (l1, v1, i1, g1, e8) = runs[False], runs[True]
    assert i0.shape == (64, 32, 9)
    np.testing.assert_allclose(l1, l0, rtol=2e-2)
    np.testing.assert_allclose(v1, v0, rtol=5e-2)
    assert (i0 == i1).float().mean().item() > 0.9
    assert F.cosine_similarity(g1.reshape(-1), g0.reshape(-1), dim=0).item() > 0.98
    e8.model.train()
    losses = [float(e8.train_step(ids, mask)["loss_recon"]) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] and e8._graphs, losses


def test_weight_quantisation_replayed_from_a_graph_equals_eager():
    """kvq_fp8_quantize_segments inside a captured hipGraph, replayed on changing weights: scales and bytes equal the eager call every
    time.  Regression test of round 4's finding: the amax accumulators used to be cleared by hipMemsetAsync, and a memset node of a
    replayed graph did not keep its place ahead of the amax kernel -- with 123 segments (every forward GEMM on fp8) two identically
    seeded training runs parted ways after 8 - 14 steps (tools/fp8_flake.py); a kernel clears them now."""
    from kvq._ffi import check, lib, stream_ptr
    nseg, seg_n = 128, 768 * 768
    g = torch.Generator(device="cuda").manual_seed(0)
    src = torch.randn(nseg * seg_n, device="cuda", generator=g).to(torch.bfloat16)
    off = (torch.arange(nseg, device="cuda", dtype=torch.int64) * seg_n)
    n = torch.full((nseg,), seg_n, device="cuda", dtype=torch.int64)
    dst = torch.zeros(nseg * seg_n, dtype=torch.uint8, device="cuda")
    amax = torch.zeros(nseg, device="cuda")
    scale = torch.ones(nseg, device="cuda")

    def quantise():
        check(lib().kvq_fp8_quantize_segments(src.data_ptr(), off.data_ptr(), n.data_ptr(), nseg, seg_n, dst.data_ptr(), amax.data_ptr(),
                                              scale.data_ptr(), stream_ptr()), "kvq_fp8_quantize_segments")
    quantise()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        quantise()
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    base = src.clone()
    for it in range(40):
        src.copy_(base * (0.01 if it % 2 else 3.0) * (1.0 + 0.1 * (it % 5)))      # every replay sees other maxima than the one before
        graph.replay()
        got_scale, got = scale.clone(), dst.clone()
        quantise()                                                                   # eager, same stream: the truth
        assert torch.equal(scale, got_scale), f"replay {it}: {(scale != got_scale).sum().item()} of {nseg} scales differ from the eager call"
        assert torch.equal(dst, got), f"replay {it}: quantised bytes differ"
