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


@pytest.mark.parametrize("scope", ["all", True, "wide"])
def test_engine_fp8_forward_against_the_bf16_engine(scope):
    """bert-base widths, 2 layers, 2048 tokens: forward GEMMs on the fp8 matrix cores -- every one, the inputs quantised by the kernels
    that produce them (fp8_forward=True, round 5) or by a pass each ("all"), or only the two widest ("wide": LM head, all-layer
    cross-K/V) --, backward in bf16.
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
    assert ("enc.0.f1.w" in e8._w8_index) == (scope != "wide") and "dec.emb.word" in e8._w8_index
    assert (e8._cakv_w[0] in e8._w8_index) == (scope != "wide") and e8._fp8_fused == (scope is True)
    key = "enc.0.f1.w" if scope != "wide" else "dec.emb.word"
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


# ---- round 5: the fp8 copy of an activation written by the kernel that produces it ---------------------------------------------------
def _state(scale):
    from kvq._ffi import lib
    st = torch.zeros(lib().kvq_fp8_state_floats(), dtype=torch.float32, device="cuda")
    st[0] = scale
    return st


def _pass(x, scale):
    """kvq_fp8_quantize_delayed(x) with the given scale: (bytes, the scale kvq_fp8_update_scales derives from its amax)."""
    from kvq._ffi import check, lib
    from kvq.functional import _workspace  # noqa: F401  (loads the library)
    st = _state(scale)
    x8 = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    check(lib().kvq_fp8_quantize_delayed(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), x8.data_ptr(), st.data_ptr(), s), "q")
    check(lib().kvq_fp8_update_scales(st.data_ptr(), 1, 4.0, s), "u")
    return x8, st[0].item()


def _next_scale(st):
    from kvq._ffi import check, lib
    check(lib().kvq_fp8_update_scales(st.data_ptr(), 1, 4.0, torch.cuda.current_stream().cuda_stream), "u")
    assert float(st[8:].abs().max()) == 0.0                     # partials cleared
    return st[0].item()


@pytest.mark.parametrize("N,H,resid,p_drop", [(8192, 768, True, 0.1), (520, 768, False, 0.0), (64, 128, True, 0.0)])
def test_layernorm_forward_writes_the_bytes_of_the_quantisation_pass(N, H, resid, p_drop):
    from kvq import nnops
    g = torch.Generator(device="cuda").manual_seed(N)
    y = torch.randn((N, H), generator=g, device="cuda").to(torch.bfloat16)
    r = torch.randn((N, H), generator=g, device="cuda").to(torch.bfloat16) if resid else None
    gamma, beta = torch.randn(H, generator=g, device="cuda"), torch.randn(H, generator=g, device="cuda")
    st = _state(37.5)
    out, pre, mean, rstd, out8 = nnops.ln_fwd_fp8(y, r, gamma, beta, 1e-12, p_drop, 99, 3, st)
    out0, pre0, mean0, rstd0 = nnops.ln_fwd(y, r, gamma, beta, 1e-12, p_drop, 99, 3)
    assert torch.equal(out.view(torch.int16), out0.view(torch.int16)) and torch.equal(pre.view(torch.int16), pre0.view(torch.int16))
    assert torch.equal(mean, mean0) and torch.equal(rstd, rstd0)
    want8, want_scale = _pass(out0, 37.5)
    assert torch.equal(out8, want8)
    assert _next_scale(st) == want_scale
    # ... and what it means: the torch conversion of out * scale
    ref = (out0.float() * 37.5).clamp(-448, 448).cpu().to(torch.float8_e4m3fn).float()
    assert torch.equal(out8.cpu().view(torch.float8_e4m3fn).float(), ref)


@pytest.mark.parametrize("B,S,causal,p_drop", [(256, 32, False, 0.1), (16, 12, True, 0.0), (5, 32, True, 0.1)])
def test_attention_forward_writes_the_bytes_of_the_quantisation_pass(B, S, causal, p_drop):
    from kvq import nnops
    nh, H = 12, 768
    g = torch.Generator(device="cuda").manual_seed(B + S)
    qkv = torch.randn((B * S, 3 * H), generator=g, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    mask = (torch.rand((B, S), generator=g, device="cuda") < 0.8).long()
    mask[:, 0] = 1
    assert nnops.attn_fwd_fp8_ok(S, S)
    st = _state(90.0)
    ctx, lse, ctx8 = nnops.attn_fwd_fp8(q, k, v, mask, B, nh, S, S, causal, p_drop, 5, 11, st)
    ctx0, lse0 = nnops.attn_fwd(q, k, v, mask, B, nh, S, S, causal, p_drop, 5, 11)
    assert torch.equal(ctx.view(torch.int16), ctx0.view(torch.int16)) and torch.equal(lse, lse0)
    want8, want_scale = _pass(ctx0, 90.0)
    assert torch.equal(ctx8, want8) and _next_scale(st) == want_scale


@pytest.mark.parametrize("M,N,K", [(8192, 3072, 768), (520, 776, 256), (256, 256, 128)])
def test_fp8_gemm_with_the_gelu_epilogue_and_the_fp8_copy_of_its_activation(M, N, K):
    from kvq import nnops
    from kvq._ffi import check, lib
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
    amax, sx, sw = (torch.zeros(1, device="cuda") for _ in range(3))
    x8, w8 = torch.empty(x.shape, dtype=torch.uint8, device="cuda"), torch.empty(w.shape, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    check(lib().kvq_fp8_quantize(x.data_ptr(), M, K, K, x8.data_ptr(), amax.data_ptr(), sx.data_ptr(), s), "qx")
    check(lib().kvq_fp8_quantize(w.data_ptr(), N, K, K, w8.data_ptr(), amax.data_ptr(), sw.data_ptr(), s), "qw")
    h0 = nnops.gemm_fp8_nt(x8, w8, sx, sw, bias=bias)
    st = _state(12.0)
    h, a, a8 = nnops.gemm_fp8_nt_gelu(x8, w8, sx, sw, bias, state_out=st)
    assert torch.equal(h.view(torch.int16), h0.view(torch.int16))                         # the same GEMM, the same rounding
    want_a = F.gelu(h0.float())
    assert (a.float() - want_a).abs().max().item() <= 2.0 ** -8 * want_a.abs().max().item() + 1e-3
    want8, want_scale = _pass(a, 12.0)                                                    # the copy is the pass over what was stored
    assert torch.equal(a8, want8) and _next_scale(st) == want_scale
    h2, a2 = nnops.gemm_fp8_nt_gelu(x8, w8, sx, sw, bias)                                 # without the copy
    assert torch.equal(a2.view(torch.int16), a.view(torch.int16))


def test_fused_scope_equals_the_scope_with_quantisation_passes():
    """Scope "fused" changes WHO writes the fp8 copies, not the bytes: three training steps (eager, capture, replay) give the
    same losses as scope "all" up to the one kernel that differs -- BertIntermediate's GELU runs in the fp8 GEMM's epilogue
    instead of a separate kernel (same formula, same input bits) -- and the step launches no quantisation pass for the layers."""
    from dsentences.synthetic import random_token_batch
    from kvq.engine import TrainEngine
    ids, mask = (t.cuda() for t in random_token_batch(64, 32, torch.Generator().manual_seed(6)))
    runs = {}
    for scope in ("all", "fused"):
        eng = TrainEngine(_build(1).train(), lr=2e-4, fp8_forward=scope)
        losses = [float(eng.train_step(ids, mask)["loss_recon"]) for _ in range(4)]
        runs[scope] = (losses, eng)
    (la, ea), (lf, ef) = runs["all"], runs["fused"]
    assert np.isfinite(lf).all() and lf[-1] < lf[0]
    np.testing.assert_allclose(lf, la, rtol=2e-3)
    # the activation scales the two engines arrived at (one record per GEMM weight): equal wherever the producer wrote the copy
    sa, sf = ea._a8_state[:, 0].cpu().numpy(), ef._a8_state[:, 0].cpu().numpy()
    assert ea._w8_index == ef._w8_index
    np.testing.assert_allclose(sf, sa, rtol=5e-2)


def test_periodic_weight_scales_follow_the_device_step_count():
    """kvq_fp8_quantize_segments_periodic (round 5): inside the training step the per-weight amax pass runs only when the DEVICE step
    count is a multiple of the period; between refreshes the bytes are the conversion with the scale of the last refresh
    (saturating).  Captured once and replayed while the counter moves: the graph follows the counter."""
    from kvq._ffi import check, lib, stream_ptr
    nseg, seg_n, period = 6, 4096, 4
    g = torch.Generator(device="cuda").manual_seed(1)
    base = torch.randn(nseg * seg_n, device="cuda", generator=g)
    src = base.to(torch.bfloat16)
    off = torch.arange(nseg, device="cuda", dtype=torch.int64) * seg_n
    n = torch.full((nseg,), seg_n, device="cuda", dtype=torch.int64)
    dst = torch.zeros(nseg * seg_n, dtype=torch.uint8, device="cuda")
    amax, scale = torch.zeros(nseg, device="cuda"), torch.ones(nseg, device="cuda")
    step = torch.zeros(4, dtype=torch.int64, device="cuda")               # (a StepState: the count is its first 8 bytes)

    def quantise():
        check(lib().kvq_fp8_quantize_segments_periodic(src.data_ptr(), off.data_ptr(), n.data_ptr(), nseg, seg_n, dst.data_ptr(),
                                                       amax.data_ptr(), scale.data_ptr(), step.data_ptr(), period, stream_ptr()), "periodic")
    quantise()                                                            # step 0: a refresh
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        quantise()
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    held = None
    for t in range(10):
        step[0] = t
        src.copy_((base * (1.0 + 0.5 * t)).to(torch.bfloat16))            # the weights grow: every refresh sees a larger amax
        graph.replay()
        torch.cuda.synchronize()
        seg = src.view(nseg, seg_n).float()
        if t % period == 0:
            held = 448.0 / seg.abs().amax(dim=1)
            torch.testing.assert_close(scale, held, rtol=1e-6, atol=0)
        else:
            torch.testing.assert_close(scale, held, rtol=1e-6, atol=0)   # unchanged since the last refresh
        want = (seg * scale[:, None]).clamp(-448, 448).cpu().to(torch.float8_e4m3fn).float()
        assert torch.equal(dst.view(nseg, seg_n).cpu().view(torch.float8_e4m3fn).float(), want), t


def test_adam_kernel_writes_the_fp8_mirror_of_the_weights_it_updates():
    """kvq_adam_step_dev_fp8 (round 5): the optimiser kernel writes the e4m3 mirror of the GEMM weights from the bf16 value it stores in
    the shadow, with the scales in force -- the bytes of a conversion pass over the new shadow; elements outside every segment are
    left alone; the master / moments / shadow equal the plain kernel's."""
    from kvq import nnops
    from kvq._ffi import check, lib, stream_ptr
    n = 6 * 2048 + 512
    segs = [(16, 2048), (2064 + 496, 4096 + 1024), (11 * 1024, 1024)]            # starts / lengths in multiples of 16, spans cut anywhere
    g = torch.Generator(device="cuda").manual_seed(3)
    p0 = torch.randn(n, device="cuda", generator=g) * 0.05
    grad = (torch.randn(n, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    off = torch.tensor([o for o, _ in segs], dtype=torch.int64, device="cuda")
    cnt = torch.tensor([c for _, c in segs], dtype=torch.int64, device="cuda")
    scale = torch.tensor([300.0, 2000.0, 900.0], device="cuda")
    span = torch.full(((n + 2047) // 2048,), -1, dtype=torch.int32)
    for si, (o, c) in enumerate(segs):
        for k in range(o // 2048, (o + c - 1) // 2048 + 1):
            lo, hi = 2048 * k, min(2048 * k + 2048, n)
            span[k] = si if (o <= lo and o + c >= hi and span[k] == -1) else -2
    span = span.cuda()
    state = nnops.new_step_state("cuda")
    nnops.step_state_advance(state, 1e-3, 0.1, [], 0.9, 0.999)
    runs = []
    for fused in (False, True):
        p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        shadow = p.to(torch.bfloat16)
        w8 = torch.full((n,), 0xAB, dtype=torch.uint8, device="cuda")
        if fused:
            lo, hi = 8, n - 8                                   # a range that starts inside the first span (as the engine's chunks do)
            nnops.adam_step_dev(p[:lo], grad[:lo], m[:lo], v[:lo], state, shadow=shadow[:lo])
            check(lib().kvq_adam_step_dev_fp8(p[lo:hi].data_ptr(), grad[lo:hi].data_ptr(), m[lo:hi].data_ptr(), v[lo:hi].data_ptr(), None,
                                              shadow[lo:hi].data_ptr(), hi - lo, 1, state.data_ptr(), 0.9, 0.999, 1e-8, 0.0, 1.0,
                                              w8.data_ptr(), span.data_ptr(), scale.data_ptr(), off.data_ptr(), cnt.data_ptr(), len(segs), lo,
                                              stream_ptr()), "kvq_adam_step_dev_fp8")
            nnops.adam_step_dev(p[hi:], grad[hi:], m[hi:], v[hi:], state, shadow=shadow[hi:])
        else:
            nnops.adam_step_dev(p, grad, m, v, state, shadow=shadow)
        torch.cuda.synchronize()
        runs.append((p, m, v, shadow, w8))
    (p_a, m_a, v_a, s_a, _), (p_b, m_b, v_b, s_b, w8) = runs
    assert torch.equal(p_a, p_b) and torch.equal(m_a, m_b) and torch.equal(v_a, v_b) and torch.equal(s_a.view(torch.int16), s_b.view(torch.int16))
    inside = torch.zeros(n, dtype=torch.bool)
    for si, (o, c) in enumerate(segs):
        want = (s_b[o:o + c].float() * scale[si]).clamp(-448, 448).cpu().to(torch.float8_e4m3fn).float()
        assert torch.equal(w8[o:o + c].cpu().view(torch.float8_e4m3fn).float(), want), si
        inside[o:o + c] = True
    assert bool((w8.cpu()[~inside] == 0xAB).all())                               # nothing written outside the segments
