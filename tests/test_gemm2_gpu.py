"""csrc/kvq_gemm2.hip (kvq_gemm_bf16 / kvq_gemm_grouped_bf16) against torch f32 matmuls of the same bf16 operands: all three
layouts of an nn.Linear's forward / input gradient / weight gradient (modeling_bert.py:139-352 behind Bagon.py:46-53), every
workgroup tile, ragged edges, strided operands, bias, accumulate, grouped launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = ["128x192", "128x256", "256x192", "256x256", "64x128", "128x192h"]


def _ops(layout, M, N, K, seed=0, lda_pad=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    def rnd(r, c, pad=0):
        t = torch.randn((r, c + pad), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
        return t[:, :c] if pad else t
    if layout == "nt":
        a, b = rnd(M, K, lda_pad), rnd(N, K)
        ref = a.float() @ b.float().t()
    elif layout == "nn":
        a, b = rnd(M, K, lda_pad), rnd(K, N)
        ref = a.float() @ b.float()
    else:
        a, b = rnd(K, M, lda_pad), rnd(K, N)
        ref = a.float().t() @ b.float()
    return a, b, ref


def _check(out, ref, K):
    # bf16 output rounding (2^-9 relative) on values of magnitude ~sqrt(K)
    err = (out.float() - ref).abs().max().item()
    tol = 2.0 ** -8 * ref.abs().max().item() + 1e-3
    assert err <= tol, (err, tol)
    rel = (out.float() - ref).norm().item() / ref.norm().item()
    assert rel < 3e-3, rel


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("shape", [(512, 768, 256), (1000, 776, 192), (264, 200, 64), (8, 8, 128), (1024, 3072, 768)])
def test_gemm_layouts_tiles_and_edges(layout, tile, shape):
    from kvq import nnops
    M, N, K = shape
    a, b, ref = _ops(layout, M, N, K, seed=M + N)
    out = nnops.gemm(a, b, layout, tile=tile)
    _check(out, ref, K)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_bias_accumulate_strides(layout):
    from kvq import nnops
    M, N, K = 640, 392, 320
    a, b, ref = _ops(layout, M, N, K, seed=5, lda_pad=24)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    big = torch.randn((M, N + 40), device="cuda").to(torch.bfloat16)
    out = big[:, 8:8 + N]                                      # row stride N + 40, offset 16 bytes
    before = out.float().clone()
    nnops.gemm(a, b, layout, bias=bias, out=out, accumulate=True)
    want = (ref + bias.float()).to(torch.bfloat16).float() + before          # kernel rounds the product, then adds C in f32
    err = (out.float() - want).abs().max().item()
    assert err <= 2.0 ** -7 * want.abs().max().item(), err
    assert torch.equal(big[:, :8], big[:, :8]) and torch.isfinite(big.float()).all()


def test_gemm_does_not_touch_memory_outside_the_output():
    from kvq import nnops
    M, N, K = 200, 136, 128
    a, b, ref = _ops("nt", M, N, K, seed=9)
    big = torch.full((M + 2, N + 16), 7.0, device="cuda", dtype=torch.bfloat16)
    out = big[1:M + 1, 8:8 + N]
    nnops.gemm(a, b, "nt", out=out)
    _check(out, ref, K)
    assert (big[0] == 7).all() and (big[-1] == 7).all() and (big[:, :8] == 7).all() and (big[:, 8 + N:] == 7).all()


@pytest.mark.parametrize("tile", ["128x256", "128x192"])
def test_grouped_weight_gradients_of_one_layer(tile):
    """The engine's per-layer launch: four TN problems (QKV, attention output, FFN1, FFN2 weight gradients) in one grid."""
    from kvq import nnops
    T = 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    probs, outs, refs = [], [], []
    for (m, n) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        gy = torch.randn((T, m), generator=g, device="cuda").to(torch.bfloat16)
        x = torch.randn((T, n), generator=g, device="cuda").to(torch.bfloat16)
        out = torch.empty((m, n), device="cuda", dtype=torch.bfloat16)
        probs.append(nnops.gemm_problem(gy, x, out, "tn"))
        outs.append(out); refs.append(gy.float().t() @ x.float())
        outs.append(gy); outs.append(x)            # keep alive
    nnops.gemm_grouped(probs, "tn", tile)
    for i, r in enumerate(refs):
        _check(outs[3 * i], r, T)


def test_grouped_weight_gradients_of_two_decoder_layers():
    """The engine's default launch: twelve TN problems (two decoder layers) as 252 tiles of 256 x 256; 17 problems are refused."""
    from kvq import nnops
    from kvq._ffi import KvqError
    T = 512
    g = torch.Generator(device="cuda").manual_seed(2)
    probs, outs, refs, keep = [], [], [], []
    for (m, n) in [(2304, 768), (768, 768), (768, 768), (768, 768), (3072, 768), (768, 3072)] * 2:
        gy = torch.randn((T, m), generator=g, device="cuda").to(torch.bfloat16)
        x = torch.randn((T, n), generator=g, device="cuda").to(torch.bfloat16)
        out = torch.empty((m, n), device="cuda", dtype=torch.bfloat16)
        probs.append(nnops.gemm_problem(gy, x, out, "tn"))
        outs.append(out); refs.append(gy.float().t() @ x.float()); keep += [gy, x]
    nnops.gemm_grouped(probs, "tn", "256x256")
    for o, r in zip(outs, refs):
        _check(o, r, T)
    with pytest.raises(KvqError):
        nnops.gemm_grouped((probs * 2)[:17], "tn", "256x256")


def test_gemm_rejects_bad_arguments():
    """The MFMA entry point refuses what it cannot take (the C ABI's contract); nnops.gemm routes such a product to the any-shape
    kernel instead, and refuses only what neither takes (f32 operands)."""
    from kvq import nnops
    from kvq._ffi import KvqError, check, lib, stream_ptr
    a = torch.zeros((64, 100), device="cuda", dtype=torch.bfloat16)
    b = torch.zeros((64, 100), device="cuda", dtype=torch.bfloat16)
    out = torch.empty((64, 64), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(KvqError):                      # K = 100 is not a multiple of 64
        check(lib().kvq_gemm_bf16(a.data_ptr(), b.data_ptr(), None, out.data_ptr(), 64, 64, 100, 100, 100, 64, 0, 0, 0, stream_ptr()), "kvq_gemm_bf16")
    assert torch.equal(nnops.gemm(a, b, "nt"), torch.zeros_like(out))
    with pytest.raises(KvqError):
        nnops.gemm(a.float(), b.float(), "nt")


def _gelu(x):
    return torch.nn.functional.gelu(x)


@pytest.mark.parametrize("tile", ["256x192", "128x256"])
def test_gemm_gelu_epilogue(tile):
    """BertIntermediate (modeling_bert.py:325-337): h = x W^T + b and gelu(h) from one kernel."""
    from kvq import nnops
    M, N, K = 1032, 776, 256
    a, b, ref = _ops("nt", M, N, K, seed=3)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    h, g = nnops.gemm_gelu(a, b, bias, tile=tile)
    want_h = ref + bias.float()
    _check(h, want_h, K)
    want_g = _gelu(h.float())                                   # the activation of the bf16 value that was stored
    assert (g.float() - want_g).abs().max().item() <= 2.0 ** -8 * want_g.abs().max().item() + 1e-3


@pytest.mark.parametrize("tile", ["256x192", "128x256"])
def test_gemm_dgelu_epilogue_and_bias_partials(tile):
    """Autograd of BertOutput.dense + BertIntermediate's activation: g_h = (g W2) * gelu'(h); partial rows sum to colsum(g_h)."""
    from kvq import nnops
    M, N, K = 1032, 776, 256
    gy, w, ref = _ops("nn", M, N, K, seed=4)
    h = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    g_h, part = nnops.gemm_dgelu(gy, w, h, tile=tile)
    hf = h.float().requires_grad_(True)
    _gelu(hf).backward(ref.to(torch.bfloat16).float())          # the kernel rounds the product to bf16 before the derivative
    want = hf.grad
    assert (g_h.float() - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item() + 1e-3
    bm = int(tile.split("x")[0])
    assert part.shape == (-(-M // bm), N)
    torch.testing.assert_close(part.sum(0), g_h.float().sum(0), rtol=1e-4, atol=1e-2)
    for t in range(part.shape[0]):                                # each partial row is exactly its row tile's column sum
        torch.testing.assert_close(part[t], g_h[t * bm:(t + 1) * bm].float().sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("tile", ["256x192", "256x256"])
@pytest.mark.parametrize("shape", [(8192, 3072, 768), (8192, 2312, 256), (4104, 30528, 128)])
def test_multi_round_launches(layout, tile, shape):
    """More tiles than CUs of one problem (several rounds of workgroups per CU): ragged edges in both dimensions, bias."""
    from kvq import nnops
    M, N, K = shape
    a, b, ref = _ops(layout, M, N, K, seed=M + N + K)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    big = torch.full((M + 1, N + 24), 3.0, device="cuda", dtype=torch.bfloat16)
    out = big[:M, 8:8 + N]
    nnops.gemm(a, b, layout, bias=bias, out=out, tile=tile)
    _check(out, ref + bias.float(), K)
    assert (big[M] == 3).all() and (big[:, :8] == 3).all() and (big[:, 8 + N:] == 3).all()
    out2 = nnops.gemm(a, b, layout, bias=bias, tile=tile)          # same launch again: bitwise reproducible
    assert torch.equal(out2, out)


@pytest.mark.parametrize("tile", ["128x192", "128x256", "256x192", "256x256"])
@pytest.mark.parametrize("shape", [(8192, 3072, 768), (8192, 2304, 768), (2048, 30528, 768), (520, 1000, 192), (256, 256, 192),
                                   (4104, 18432, 256)])
def test_persistent_tile_loop_equals_the_one_tile_kernel(tile, shape):
    """KVQ_GEMM_PERSISTENT (one workgroup per CU walks its tiles; the k-tiles of successive tiles are one LDS-DMA stream; the
    accumulators go to memory from registers; edge tiles are pulled back inside the matrix): same bits as the one-tile-per-
    workgroup kernel -- same summation order per element -- on 1 .. 30 tiles per CU, ragged M and N, strided output with a
    canary frame, with and without bias; and both against an f32 matmul."""
    from kvq import nnops
    M, N, K = shape
    a, b, ref = _ops("nt", M, N, K, seed=M + N + K + 1)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    for bv in (bias, None):
        big = torch.full((M + 1, N + 24), 3.0, device="cuda", dtype=torch.bfloat16)
        out = big[:M, 8:8 + N]
        nnops.gemm(a, b, "nt", bias=bv, out=out, tile=tile + "p")
        _check(out, ref + (bv.float() if bv is not None else 0.0), K)
        assert (big[M] == 3).all() and (big[:, :8] == 3).all() and (big[:, 8 + N:] == 3).all()
        one = nnops.gemm(a, b, "nt", bias=bv, tile=tile)
        assert torch.equal(out, one)
        again = nnops.gemm(a, b, "nt", bias=bv, tile=tile + "p")
        assert torch.equal(again, one)


@pytest.mark.parametrize("tile", ["256x192", "128x256"])
def test_persistent_gemm_gelu_epilogue(tile):
    from kvq import nnops
    for (M, N, K) in [(8192, 3072, 768), (1032, 776, 256)]:
        a, b, ref = _ops("nt", M, N, K, seed=7)
        bias = torch.randn(N, device="cuda").to(torch.bfloat16)
        h0, g0 = nnops.gemm_gelu(a, b, bias, tile=tile)
        h1, g1 = nnops.gemm_gelu(a, b, bias, tile=tile + "p")
        _check(h1, ref + bias.float(), K)
        assert torch.equal(h1, h0) and torch.equal(g1, g0)


def test_persistent_gemm_rejects_what_it_does_not_cover():
    from kvq import nnops
    from kvq._ffi import KvqError
    a, b, _ = _ops("nn", 512, 512, 256)
    with pytest.raises(KvqError):
        nnops.gemm(a, b, "nn", tile="256x256p")                         # layout
    a, b, _ = _ops("nt", 512, 512, 256)
    with pytest.raises(KvqError):
        nnops.gemm(a, b, "nt", out=torch.zeros((512, 512), device="cuda", dtype=torch.bfloat16), accumulate=True, tile="256x256p")
    a, b, _ = _ops("nt", 128, 512, 256)
    with pytest.raises(KvqError):
        nnops.gemm(a, b, "nt", tile="256x256p")                         # less than one tile of rows
    a, b, _ = _ops("nt", 512, 512, 64)
    with pytest.raises(KvqError):
        nnops.gemm(a, b, "nt", tile="128x256p")                         # fewer k-tiles than ring slots


@pytest.mark.parametrize("M,N,V,K", [(296, 776, 770, 128), (512, 256, 256, 64), (2048, 30528, 30522, 768), (8192, 30528, 30522, 768)])
def test_lm_head_gemm_with_loss_statistics(M, N, V, K):
    """kvq_gemm_bf16_ce + kvq_ce_forward_stats against the plain GEMM + kvq_ce_forward: the logits bit
    for bit, arg-max and accuracy exactly, lse / loss to f32 summation-order noise; padding columns >= V take no part."""
    from kvq import nnops
    from kvq._ffi import lib, check
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.2).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
    b[V:] = 50.0                                                # a padding column that would win every arg-max if it were counted
    tgt = torch.randint(0, V, (M,), generator=g, device="cuda")
    ref_logits = nnops.gemm(x, w, "nt", bias=b, tile="256x256")
    f = lambda: (torch.empty(M, device="cuda"), torch.empty(M, device="cuda"), torch.empty(M, dtype=torch.int64, device="cuda"),
                 torch.empty(2, device="cuda"))
    rl0, lse0, pred0, out0 = f()
    check(lib().kvq_ce_forward(ref_logits.data_ptr(), tgt.data_ptr(), M, V, N, 1, rl0.data_ptr(), lse0.data_ptr(), pred0.data_ptr(),
                               out0[0:].data_ptr(), out0[1:].data_ptr(), None), "kvq_ce_forward")
    logits, stats = nnops.gemm_ce(x, w, b, V)
    rl1, lse1, pred1, out1 = f()
    nnops.ce_forward_stats(logits, tgt, stats, rl1, lse1, pred1, out1[0:], out1[1:])
    assert torch.equal(logits, ref_logits)
    assert torch.equal(pred1, pred0) and int(pred1.max()) < V
    torch.testing.assert_close(lse1, lse0, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(rl1, rl0, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(out1, out0, rtol=1e-5, atol=1e-6)
    want = torch.logsumexp(ref_logits[:, :V].float(), dim=1)
    torch.testing.assert_close(lse1, want, rtol=2e-6, atol=2e-6)
    # the reference's own expression (models/shelgon3/Trainer.py:94-101): KL(one-hot || softmax), "batchmean", arg-max of the softmax
    lg = ref_logits[:, :V].float()
    kl = torch.nn.functional.kl_div(torch.log_softmax(lg, dim=-1), torch.nn.functional.one_hot(tgt, V).float(), reduction="batchmean")
    torch.testing.assert_close(out1[0], kl, rtol=2e-5, atol=1e-6)
    assert torch.equal(pred1, torch.softmax(lg, dim=-1).argmax(-1)) or (pred1 != lg.argmax(-1)).sum().item() == 0
    torch.testing.assert_close(out1[1], (pred1 == tgt).float().mean(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("shape", [(72, 128, 128), (72, 9, 128), (100, 2048, 72), (7, 5, 3), (333, 130, 129), (768, 768, 768)])
def test_gemm_any_shape_kernel(layout, shape):
    """csrc/kvq_gemm_any.hip: products the MFMA kernel refuses (K not a multiple of 64, odd M / N / leading dimensions, 2-byte
    aligned views) -- what nnops.gemm routes there -- plus one it would take, called directly; bias and accumulate."""
    from kvq import nnops
    from kvq._ffi import check, lib, stream_ptr
    M, N, K = shape
    a, b, ref = _ops(layout, M, N, K, seed=M + N + K)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    check(lib().kvq_gemm_any_bf16(a.data_ptr(), b.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), b.stride(0), N,
                                  {"nt": 0, "nn": 1, "tn": 2}[layout], 0, stream_ptr()), "kvq_gemm_any_bf16")
    want = ref + bias.float()
    assert (out.float() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() + 1e-3
    if not nnops.gemm_mfma_ok(a, b, out, layout, bias):
        routed = nnops.gemm(a, b, layout, bias=bias)                       # the router sends it to the same kernel: same bits
        assert torch.equal(routed, out)
    before = out.float().clone()
    check(lib().kvq_gemm_any_bf16(a.data_ptr(), b.data_ptr(), None, out.data_ptr(), M, N, K, a.stride(0), b.stride(0), N,
                                  {"nt": 0, "nn": 1, "tn": 2}[layout], 1, stream_ptr()), "kvq_gemm_any_bf16")
    want2 = ref.to(torch.bfloat16).float() + before
    assert (out.float() - want2).abs().max().item() <= 2.0 ** -7 * want2.abs().max().item() + 1e-3


def test_gemm_router_takes_misaligned_views():
    """nnops.gemm on views whose base is only 2-byte aligned / whose row stride is odd: the any-shape kernel, right answer."""
    from kvq import nnops
    g = torch.Generator(device="cuda").manual_seed(3)
    big_a = torch.randn((130, 200), generator=g, device="cuda").to(torch.bfloat16)
    big_b = torch.randn((96, 203), generator=g, device="cuda").to(torch.bfloat16)
    a, b = big_a[:, 3:131], big_b[:, 1:129]                    # [130, 128] and [96, 128], misaligned
    assert not nnops.gemm_mfma_ok(a, b, None, "nt")
    out = nnops.gemm(a, b, "nt")
    ref = a.float() @ b.float().t()
    assert (out.float() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-3


def test_pick_tile_rule():
    """The rule that replaced the engine's exact-shape tables: at the benchmarked 8192 rows it reproduces the measured choices;
    at the reference's own row counts (12 tokens x 64 / 128 sentences) it turns to the small tile."""
    from kvq import nnops
    name = lambda M, N: nnops.TILE_NAMES[nnops.pick_tile(M, N)]
    assert name(8192, 768) == "128x192" and name(8192, 3072) == "256x192" and name(8192, 18432) == "256x256" and name(8192, 30528) == "256x256"
    assert name(768, 768) == "64x128" and name(1536, 768) == "64x128"


@pytest.mark.parametrize("tile", ["128x192", "128x256", "64x128", None])
@pytest.mark.parametrize("shape,p_drop", [((1032, 776, 256), 0.1), ((8192, 768, 768), 0.1), ((520, 768, 3072), 0.0), ((264, 200, 64), 0.5)])
def test_gemm_dropout_residual_epilogue_equals_the_two_kernel_form(tile, shape, p_drop):
    """kvq_gemm_bf16_dropres (round 5): the dense layer of a BertSelfOutput / BertOutput block (modeling_bert.py:282-296, 339-352)
    with dropout and the residual add in its epilogue stores, BIT FOR BIT, the `pre` that kvq_dropout_residual_ln_fwd stores behind
    the plain GEMM (same roundings, same Philox masks), and LayerNorm alone behind it gives the two-kernel form's output."""
    from kvq import nnops
    M, N, K = shape
    a, b, _ = _ops("nt", M, N, K, seed=M + K)
    g = torch.Generator(device="cuda").manual_seed(77)
    bias = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
    resid = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16)
    gamma, beta = torch.randn(N, generator=g, device="cuda"), torch.randn(N, generator=g, device="cuda")
    seed, site = 123456789, 7
    y = nnops.gemm(a, b, "nt", bias=bias)
    out0, pre0, mean0, rstd0 = nnops.ln_fwd(y, resid, gamma, beta, 1e-12, p_drop, seed, site)
    pre1 = nnops.gemm_dropres(a, b, bias, resid, p_drop, seed, site, tile=tile)
    assert torch.equal(pre1.view(torch.int16), pre0.view(torch.int16))
    if p_drop > 0:                                   # the mask is there: about p_drop of the elements are the bare residual
        dropped = (pre1 == resid).float().mean().item()
        assert abs(dropped - p_drop) < 0.02, dropped
    out1, none, mean1, rstd1 = nnops.ln_fwd(pre1, None, gamma, beta, 1e-12, 0.0, 0, 0, save_pre=False)
    assert none is None and torch.equal(out1.view(torch.int16), out0.view(torch.int16))
    assert torch.equal(mean1, mean0) and torch.equal(rstd1, rstd0)
