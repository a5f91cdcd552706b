"""GPU numerics of the fused block kernels (csrc/kvq_nn.hip) against plain PyTorch f32 references of the same ops
(the ops HuggingFace's BertLayer runs: modeling_bert.py:111-352) and torch.optim.Adam."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from kvq import _ffi, nnops
    _ffi.lib()
    assert torch.cuda.is_available()
    return nnops


def _tol(dtype):
    return dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("N,H", [(37, 768), (8, 64), (130, 3072), (5, 2048)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ln_fwd_bwd_no_dropout(ops, N, H, dtype):
    torch.manual_seed(N + H)
    y = torch.randn(N, H, device="cuda").to(dtype)
    r = torch.randn(N, H, device="cuda").to(dtype)
    gamma = torch.randn(H, device="cuda"); beta = torch.randn(H, device="cuda")
    g = torch.randn(N, H, device="cuda").to(dtype)
    out, pre, mean, rstd = ops.ln_fwd(y, r, gamma, beta, 1e-12)
    yr = y.float().requires_grad_(True); rr = r.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    pre_ref = (yr + rr).to(dtype).float() if dtype != torch.float32 else yr + rr
    ref = F.layer_norm((yr + rr), (H,), gr, br, 1e-12)
    torch.testing.assert_close(out.float(), F.layer_norm(pre.float(), (H,), gamma, beta, 1e-12), **_tol(dtype))
    torch.testing.assert_close(pre.float(), pre_ref.detach(), **_tol(dtype))
    ref.backward(g.float())
    flat = torch.zeros(3 * H, device="cuda")                       # [dense bias | LN weight | LN bias] adjacent, as in the engine
    gbias, gg, gb = flat[:H], flat[H:2 * H], flat[2 * H:]
    g_y, g_r = ops.ln_bwd(g, pre, mean, rstd, gamma, g_gamma=gg, g_beta=gb, g_bias_prev=gbias)
    torch.testing.assert_close(gbias, g_y.float().sum(0), rtol=1e-4, atol=1e-3 * math.sqrt(N))
    gbias2 = torch.zeros(H, device="cuda")                         # non-adjacent destinations take the separate passes
    ops.ln_bwd(g, pre, mean, rstd, gamma, g_gamma=torch.zeros(H, device="cuda"), g_bias_prev=gbias2)
    torch.testing.assert_close(gbias2, gbias)
    t = _tol(dtype)
    torch.testing.assert_close(g_y.float(), yr.grad, **t)
    torch.testing.assert_close(g_r.float(), rr.grad, **t)
    scale = math.sqrt(N)
    torch.testing.assert_close(gg, gr.grad, rtol=t["rtol"], atol=t["atol"] * scale)
    torch.testing.assert_close(gb, br.grad, rtol=t["rtol"], atol=t["atol"] * scale)
    gg2 = torch.ones(H, device="cuda", dtype=torch.bfloat16)
    ops.ln_bwd(g, pre, mean, rstd, gamma, g_gamma=gg2, accumulate=True, need_g_y=False, need_g_resid=False)
    torch.testing.assert_close(gg2.float(), gr.grad + 1, rtol=2e-2, atol=2e-2 * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,p", [(768, 0.1), (128, 0.0), (1024, 0.25)])
def test_embedding_block_kernels_equal_the_separate_kernels(ops, dtype, H, p):
    """BertEmbeddings (modeling_bert.py:53-110): kvq_embed_ln_fwd == gather + (pos + type) + kvq_dropout_residual_ln_fwd +
    kvq_dropout bit for bit, and kvq_ln_dropout_bwd_partial == kvq_dropout on the gradient + the LayerNorm backward; the
    composition itself is checked against torch (embedding, layer_norm, autograd) with dropout off."""
    B, S, V = 9, 13, 500
    torch.manual_seed(H + int(100 * p))
    ids = torch.randint(0, V, (B, S), device="cuda")
    word = torch.randn(V, H, device="cuda").to(dtype); pos = torch.randn(40, H, device="cuda").to(dtype)
    typ = torch.randn(2, H, device="cuda").to(dtype)
    gamma = torch.randn(H, device="cuda"); beta = torch.randn(H, device="cuda")
    seed, site = 1234, 7
    out, pre, mean, rstd = ops.embed_ln_fwd(ids.reshape(-1), word, pos, typ[0], gamma, beta, 1e-12, S, p, seed, site)
    y = F.embedding(ids.reshape(-1), word)
    pt = (pos[:S] + typ[0]).repeat(B, 1)
    o2, pre2, mean2, rstd2 = ops.ln_fwd(y, pt, gamma, beta, 1e-12)
    if p > 0:
        o2 = ops.dropout(o2, p, seed, site)
    assert torch.equal(pre, pre2) and torch.equal(out, o2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    g = torch.randn(B * S, H, device="cuda").to(dtype)
    g_x, part = ops.ln_dropout_bwd_partial(g, pre, mean, rstd, gamma, p, seed, site)
    gd = ops.dropout(g, p, seed, site) if p > 0 else g
    g_y2, g_r2, part2 = ops.ln_bwd_partial(gd, pre, mean, rstd, gamma, 0.0, 0, 0)
    assert torch.equal(g_x, g_y2) and torch.equal(part[:, H:], part2[:, H:])
    if p == 0:
        wr = word.float().requires_grad_(True)
        ref = F.layer_norm((F.embedding(ids.reshape(-1), wr) + pt.float()).to(dtype).float(), (H,), gamma, beta, 1e-12)
        torch.testing.assert_close(out.float(), ref, **_tol(dtype))
    if p > 0:                                                  # the keep rate is the requested one
        kept = (out != 0).float().mean().item()
        assert abs(kept - (1 - p)) < 0.02


def test_ln_dropout_mask_consistency(ops):
    """The mask is regenerated, not stored: forward and backward must see the same one, at the requested rate."""
    N, H, p = 512, 768, 0.1
    ones = torch.ones(N, H, device="cuda")
    gamma = torch.randn(H, device="cuda"); beta = torch.zeros(H, device="cuda")
    _, pre, mean, rstd = ops.ln_fwd(ones, None, gamma, beta, 1e-12, p, seed=123, site=7)
    mask = pre > 0
    assert torch.all((pre == 0) | torch.isclose(pre, torch.full_like(pre, 1 / (1 - p))))
    rate = 1 - mask.float().mean().item()
    assert abs(rate - p) < 0.005
    _, pre2, _, _ = ops.ln_fwd(ones, None, gamma, beta, 1e-12, p, seed=123, site=8)
    assert not torch.equal(pre, pre2)                       # another site -> another mask
    _, pre3, _, _ = ops.ln_fwd(ones, None, gamma, beta, 1e-12, p, seed=123, site=7)
    assert torch.equal(pre, pre3)                           # same (seed, site) -> same mask
    y = torch.randn(N, H, device="cuda"); r = torch.randn(N, H, device="cuda"); g = torch.randn(N, H, device="cuda")
    out, pre, mean, rstd = ops.ln_fwd(y, r, gamma, beta, 1e-12, p, seed=123, site=7)
    yr = y.clone().requires_grad_(True); rr = r.clone().requires_grad_(True)
    ref = F.layer_norm(yr * mask / (1 - p) + rr, (H,), gamma, beta, 1e-12)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    ref.backward(g)
    g_y, g_r = ops.ln_bwd(g, pre, mean, rstd, gamma, p, 123, 7)
    torch.testing.assert_close(g_y, yr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g_r, rr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,C,ld", [(8192, 768, 768), (300, 30522, 30528), (1000, 3072, 3072), (77, 13, 16), (5, 2304, 2304)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(ops, N, C, ld, dtype):
    torch.manual_seed(C)
    buf = torch.randn(N, ld, device="cuda").to(dtype)
    x = buf[:, :C]
    out = torch.empty(C, device="cuda")
    ops.colsum(buf, out, cols=C)
    ref = x.float().sum(0)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-3)
    out_b = torch.ones(C, device="cuda", dtype=torch.bfloat16)
    ops.colsum(buf, out_b, scale=0.5, accumulate=True, cols=C)
    torch.testing.assert_close(out_b.float(), 1 + 0.5 * ref, rtol=1e-2, atol=0.2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sum_slabs(ops, dtype):
    torch.manual_seed(4)
    part = torch.randn(16, 768, 768, device="cuda").to(dtype)
    out = torch.empty(768, 768, device="cuda", dtype=dtype)
    ops.sum_slabs(part, out)
    torch.testing.assert_close(out.float(), part.float().sum(0).to(dtype).float(), rtol=1e-5, atol=1e-5)


def test_reduce_batch_mixed_items(ops):
    """One launch: split-K slab sums (bf16 and f32), LayerNorm partial finals with column offsets, a colsum final, accumulate."""
    torch.manual_seed(6)
    dev = "cuda"
    slabs_b = torch.randn(16, 768, 768, device=dev).bfloat16(); out_b = torch.empty(768, 768, device=dev, dtype=torch.bfloat16)
    slabs_f = torch.randn(3, 100, 44, device=dev); out_f = torch.ones(100, 44, device=dev)
    x = torch.randn(8192, 2304, device=dev).bfloat16()
    part_c = ops.colsum_partial(x); out_c = torch.empty(2304, device=dev, dtype=torch.bfloat16)
    H = 768
    part_ln = torch.randn(1024, 3 * H, device=dev); tgt = torch.zeros(3 * H + 8, device=dev, dtype=torch.bfloat16)
    lone = torch.empty(H, device=dev)                       # f32 target, unaligned source column offset is fine in tree mode
    odd = torch.randn(5, 1001, device=dev); out_odd = torch.empty(1001, device=dev)      # cols % 4 != 0 -> tree path
    items = [ops.reduce_item(slabs_b, out_b, 16, 768 * 768, 768 * 768),
             ops.reduce_item(slabs_f, out_f, 3, 4400, 4400, scale=0.5, accumulate=True),
             ops.reduce_item(part_c, out_c, part_c.shape[0], 2304, 2304),
             ops.reduce_item(part_ln, tgt[8:], 1024, 2 * H, 3 * H, src_offset=H),
             ops.reduce_item(part_ln, lone, 1024, H, 3 * H),
             ops.reduce_item(odd, out_odd, 5, 1001, 1001)]
    ops.reduce_batch(items)
    torch.testing.assert_close(out_b.float(), slabs_b.float().sum(0).bfloat16().float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out_f, 1 + 0.5 * slabs_f.sum(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out_c.float(), x.float().sum(0), rtol=1e-2, atol=0.5)
    torch.testing.assert_close(tgt[8:8 + 2 * H].float(), part_ln[:, H:].sum(0), rtol=1e-2, atol=0.3)
    assert torch.all(tgt[:8] == 0) and torch.all(tgt[8 + 2 * H:] == 0)
    torch.testing.assert_close(lone, part_ln[:, :H].sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(out_odd, odd.sum(0), rtol=1e-5, atol=1e-5)
    many = [ops.reduce_item(slabs_f, torch.empty(100, 44, device=dev), 3, 4400, 4400) for _ in range(37)]   # > KVQ_REDUCE_MAX_ITEMS (32): several launches
    ops.reduce_batch(many)
    from kvq import _ffi
    arr = (_ffi.ReduceItem * 17)()
    assert _ffi.lib().kvq_reduce_batch(arr, 17, None) != 0 and _ffi.lib().kvq_reduce_batch(arr, 1, None) != 0     # too many / null item


def test_ln_bwd_partial_plus_reduce_equals_ln_bwd(ops):
    torch.manual_seed(8)
    N, H = 2048, 768
    g = torch.randn(N, H, device="cuda").bfloat16(); y = torch.randn(N, H, device="cuda").bfloat16(); r = torch.randn(N, H, device="cuda").bfloat16()
    gamma = torch.randn(H, device="cuda"); beta = torch.randn(H, device="cuda")
    out, pre, mean, rstd = ops.ln_fwd(y, r, gamma, beta, 1e-12, 0.1, 3, 4)
    flat = torch.zeros(3 * H, device="cuda", dtype=torch.bfloat16)
    gy0, gr0 = ops.ln_bwd(g, pre, mean, rstd, gamma, 0.1, 3, 4, g_gamma=flat[H:2 * H], g_beta=flat[2 * H:], g_bias_prev=flat[:H])
    gy1, gr1, part = ops.ln_bwd_partial(g, pre, mean, rstd, gamma, 0.1, 3, 4, want_dbias=True)
    flat1 = torch.zeros_like(flat)
    ops.reduce_batch([ops.reduce_item(part, flat1, part.shape[0], 3 * H, 3 * H)])
    assert torch.equal(gy0, gy1) and torch.equal(gr0, gr1) and torch.equal(flat, flat1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gelu(ops, dtype):
    torch.manual_seed(0)
    h = (3 * torch.randn(1000, 3072, device="cuda")).to(dtype)
    g = torch.randn_like(h)
    hr = h.float().requires_grad_(True)
    ref = F.gelu(hr)
    ref.backward(g.float())
    torch.testing.assert_close(ops.gelu_fwd(h).float(), ref.detach(), **_tol(dtype))
    torch.testing.assert_close(ops.gelu_bwd(h, g).float(), hr.grad, **_tol(dtype))
    g_h, part = ops.gelu_bwd_bias(h, g)                      # same values + column-sum partials of what was stored
    assert torch.equal(g_h, ops.gelu_bwd(h, g)) and part.shape == (125, 3072)
    torch.testing.assert_close(part.sum(0), g_h.float().sum(0), rtol=1e-4, atol=1e-2)


def _ref_attention(q, k, v, mask, causal, keep=None, p=0.0):
    """BertSelfAttention math (modeling_bert.py:111-136) in f32 on [B,nh,S,64] tensors."""
    B, nh, Sq, _ = q.shape
    Sk = k.shape[2]
    s = q @ k.transpose(-1, -2) / 8.0
    allow = torch.ones(B, 1, Sq, Sk, dtype=torch.bool, device=q.device)
    if mask is not None:
        allow = allow & mask.bool()[:, None, None, :]
    if causal:
        allow = allow & torch.ones(Sq, Sk, dtype=torch.bool, device=q.device).tril()[None, None]
    s = s.masked_fill(~allow, float("-inf"))
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    return pr @ v


@pytest.mark.parametrize("B,nh,Sq,Sk,causal,masked", [(3, 12, 32, 32, False, True), (2, 4, 32, 32, True, True), (2, 3, 12, 12, True, True),
                                                      (2, 2, 7, 32, False, False), (1, 1, 32, 5, False, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, "bf16-dot2", "bf16-fma"])
def test_attention_fwd_bwd(ops, B, nh, Sq, Sk, causal, masked, dtype):
    from kvq import _ffi
    # bf16 flavours: MFMA kernels (default, 2), packed-dot kernels (1), convert+fma kernels (0)
    assert _ffi.lib().kvq_attn_set_variant({"bf16-fma": 0, "bf16-dot2": 1}.get(dtype, 2)) == 0
    dtype = torch.bfloat16 if isinstance(dtype, str) else dtype
    torch.manual_seed(B * 100 + Sq)
    H = nh * 64
    qkv = torch.randn(B * Sq, 3 * H, device="cuda").to(dtype)        # fused QKV layout: strided q/k/v views
    kv = torch.randn(B * Sk, 2 * H, device="cuda").to(dtype)
    if Sq == Sk:
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    else:
        q, k, v = qkv[:, :H], kv[:, :H], kv[:, H:]
    mask = None
    if masked:
        lens = torch.randint(1, Sk + 1, (B,), device="cuda")
        mask = (torch.arange(Sk, device="cuda")[None] < lens[:, None]).long()
    ctx, lse = ops.attn_fwd(q, k, v, mask, B, nh, Sq, Sk, causal)
    def heads(t, S):
        return t.float().reshape(B, S, nh, 64).transpose(1, 2).detach().clone().requires_grad_(True)
    qr, kr, vr = heads(q, Sq), heads(k, Sk), heads(v, Sk)
    ref = _ref_attention(qr, kr, vr, mask, causal)
    torch.testing.assert_close(ctx.float(), ref.transpose(1, 2).reshape(B * Sq, H), **_tol(dtype))
    g = torch.randn(B * Sq, H, device="cuda").to(dtype)
    ref.backward(g.float().reshape(B, Sq, nh, 64).transpose(1, 2))
    g_q = torch.empty_like(qkv)[:, :H]; 
    gbuf_q = torch.zeros_like(qkv); gbuf_kv = torch.zeros_like(kv)
    if Sq == Sk:
        gq, gk, gv = gbuf_q[:, :H], gbuf_q[:, H:2 * H], gbuf_q[:, 2 * H:]
    else:
        gq, gk, gv = gbuf_q[:, :H], gbuf_kv[:, :H], gbuf_kv[:, H:]
    pbq = torch.full((B, H + 8), 7.0, device="cuda"); pbkv = torch.full((B, 2 * H), 7.0, device="cuda")
    ops.attn_bwd(q, k, v, mask, g, B, nh, Sq, Sk, causal, 0.0, 0, 0, gq, gk, gv, pbq[:, :H], pbkv[:, :H], pbkv[:, H:])
    # per-sentence column sums of the stored gradients (the q/k/v bias-gradient partial rows)
    bt = dict(rtol=1e-2, atol=5e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(pbq[:, :H], gq.float().reshape(B, Sq, H).sum(1), **bt)
    torch.testing.assert_close(pbkv[:, :H], gk.float().reshape(B, Sk, H).sum(1), **bt)
    torch.testing.assert_close(pbkv[:, H:], gv.float().reshape(B, Sk, H).sum(1), **bt)
    assert torch.all(pbq[:, H:] == 7.0)
    unh = lambda t, S: t.transpose(1, 2).reshape(B * S, H)
    t = _tol(dtype)
    torch.testing.assert_close(gq.float(), unh(qr.grad, Sq), **t)
    torch.testing.assert_close(gk.float(), unh(kr.grad, Sk), **t)
    torch.testing.assert_close(gv.float(), unh(vr.grad, Sk), **t)
    _ffi.lib().kvq_attn_set_variant(2)


def test_attention_bf16_dropout_same_mask_in_both_flavours(ops):
    """The MFMA, packed-dot and convert+fma bf16 kernels draw the same Philox mask: outputs agree to bf16 rounding."""
    from kvq import _ffi
    torch.manual_seed(11)
    B, nh, S, H = 4, 12, 32, 768
    qkv = torch.randn(B * S, 3 * H, device="cuda").bfloat16()
    g = torch.randn(B * S, H, device="cuda").bfloat16()
    mask = (torch.arange(S, device="cuda")[None] < torch.tensor([32, 9, 20, 4], device="cuda")[:, None]).long()
    outs = []
    for flavour in (2, 1, 0):
        _ffi.lib().kvq_attn_set_variant(flavour)
        ctx, _ = ops.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask, B, nh, S, S, True, 0.1, seed=5, site=2)
        gq = torch.zeros_like(qkv)
        ops.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], mask, g, B, nh, S, S, True, 0.1, 5, 2, gq[:, :H], gq[:, H:2 * H], gq[:, 2 * H:])
        outs.append((ctx.float(), gq.float()))
    _ffi.lib().kvq_attn_set_variant(2)
    assert _ffi.lib().kvq_attn_set_variant(3) != 0          # unknown flavour is refused, the selection stays
    for other in (1, 2):
        torch.testing.assert_close(outs[0][0], outs[other][0], rtol=3e-2, atol=3e-2)
        torch.testing.assert_close(outs[0][1], outs[other][1], rtol=5e-2, atol=5e-2)


def test_attention_dropout_mask_consistency(ops):
    """Reveal the Philox mask with q = 0, V = one-hot rows; then fwd/bwd must match a reference using that mask."""
    B, nh, S, p = 2, 3, 32, 0.1
    H = nh * 64
    q0 = torch.zeros(B * S, H, device="cuda"); k0 = torch.zeros(B * S, H, device="cuda")
    eye = torch.zeros(S, 64, device="cuda"); eye[torch.arange(S), torch.arange(S)] = 1
    v1 = eye[None, :, None, :].expand(B, S, nh, 64).reshape(B * S, H).contiguous()
    ctx, _ = ops.attn_fwd(q0, k0, v1, None, B, nh, S, S, False, p, seed=99, site=3)
    pt = ctx.reshape(B, S, nh, 64)[..., :S].permute(0, 2, 1, 3)            # P~[b,h,i,j] = keep/(S(1-p))
    keep = (pt > 0).float()
    assert abs((1 - keep.mean().item()) - p) < 0.02
    torch.manual_seed(5)
    q = torch.randn(B * S, H, device="cuda"); k = torch.randn(B * S, H, device="cuda"); v = torch.randn(B * S, H, device="cuda")
    g = torch.randn(B * S, H, device="cuda")
    heads = lambda t: t.reshape(B, S, nh, 64).transpose(1, 2).detach().clone().requires_grad_(True)
    qr, kr, vr = heads(q), heads(k), heads(v)
    ref = _ref_attention(qr, kr, vr, None, True, keep, p)
    ctx, _ = ops.attn_fwd(q, k, v, None, B, nh, S, S, True, p, seed=99, site=3)
    torch.testing.assert_close(ctx, ref.transpose(1, 2).reshape(B * S, H), rtol=1e-4, atol=1e-5)
    ref.backward(g.reshape(B, S, nh, 64).transpose(1, 2))
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attn_bwd(q, k, v, None, g, B, nh, S, S, True, p, 99, 3, gq, gk, gv)
    unh = lambda t: t.transpose(1, 2).reshape(B * S, H)
    torch.testing.assert_close(gq, unh(qr.grad), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gk, unh(kr.grad), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gv, unh(vr.grad), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,nh,Sq,Sk,causal,masked", [(3, 12, 64, 64, False, True), (2, 4, 64, 64, True, True), (2, 3, 48, 48, True, True),
                                                      (2, 2, 33, 33, True, False), (2, 2, 100, 100, True, True), (2, 2, 128, 128, False, True),
                                                      (2, 2, 40, 96, False, True), (1, 3, 70, 20, False, False), (2, 2, 20, 70, False, True)])
def test_attention_above_32_tokens(ops, B, nh, Sq, Sk, causal, masked):
    """The blocked bf16 kernels (33 .. 128 tokens, 32-token blocks, running softmax forward; dQ and dK/dV kernels backward from the
    saved output and log-sum-exp) against the f32 BertSelfAttention math: full and ragged last blocks, key-padding masks, causal
    self-attention, cross-attention with different lengths on either side.  bf16 tolerances of the 32-token kernels."""
    dtype = torch.bfloat16
    torch.manual_seed(B * 100 + Sq + Sk)
    H = nh * 64
    qkv = torch.randn(B * Sq, 3 * H, device="cuda").to(dtype)
    kv = torch.randn(B * Sk, 2 * H, device="cuda").to(dtype)
    if Sq == Sk:
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    else:
        q, k, v = qkv[:, :H], kv[:, :H], kv[:, H:]
    mask = None
    if masked:
        lens = torch.randint(1, Sk + 1, (B,), device="cuda")
        lens[0] = Sk
        mask = (torch.arange(Sk, device="cuda")[None] < lens[:, None]).long()
    ctx, lse = ops.attn_fwd(q, k, v, mask, B, nh, Sq, Sk, causal)
    heads = lambda t, S: t.float().reshape(B, S, nh, 64).transpose(1, 2).detach().clone().requires_grad_(True)
    qr, kr, vr = heads(q, Sq), heads(k, Sk), heads(v, Sk)
    ref = _ref_attention(qr, kr, vr, mask, causal)
    torch.testing.assert_close(ctx.float(), ref.transpose(1, 2).reshape(B * Sq, H), **_tol(dtype))
    s = (qr @ kr.transpose(-1, -2) / 8.0).detach()
    allow = torch.ones(B, 1, Sq, Sk, dtype=torch.bool, device="cuda")
    if mask is not None:
        allow = allow & mask.bool()[:, None, None, :]
    if causal:
        allow = allow & torch.ones(Sq, Sk, dtype=torch.bool, device="cuda").tril()[None, None]
    torch.testing.assert_close(lse, torch.logsumexp(s.masked_fill(~allow, float("-inf")), -1), rtol=1e-3, atol=2e-2)
    g = torch.randn(B * Sq, H, device="cuda").to(dtype)
    ref.backward(g.float().reshape(B, Sq, nh, 64).transpose(1, 2))
    gbuf_q = torch.zeros_like(qkv); gbuf_kv = torch.zeros_like(kv)
    if Sq == Sk:
        gq, gk, gv = gbuf_q[:, :H], gbuf_q[:, H:2 * H], gbuf_q[:, 2 * H:]
    else:
        gq, gk, gv = gbuf_q[:, :H], gbuf_kv[:, :H], gbuf_kv[:, H:]
    pbq = torch.full((B, H), 7.0, device="cuda"); pbkv = torch.full((B, 2 * H), 7.0, device="cuda")
    from kvq._ffi import KvqError
    with pytest.raises(KvqError):                       # without the forward's output / lse the long kernels cannot run
        ops.attn_bwd(q, k, v, mask, g, B, nh, Sq, Sk, causal, 0.0, 0, 0, gq, gk, gv)
    ops.attn_bwd(q, k, v, mask, g, B, nh, Sq, Sk, causal, 0.0, 0, 0, gq, gk, gv, pbq, pbkv[:, :H], pbkv[:, H:], ctx=ctx, lse=lse)
    bt = dict(rtol=1e-2, atol=1e-1)
    torch.testing.assert_close(pbq, gq.float().reshape(B, Sq, H).sum(1), **bt)
    torch.testing.assert_close(pbkv[:, :H], gk.float().reshape(B, Sk, H).sum(1), **bt)
    torch.testing.assert_close(pbkv[:, H:], gv.float().reshape(B, Sk, H).sum(1), **bt)
    unh = lambda t, S: t.transpose(1, 2).reshape(B * S, H)
    t = _tol(dtype)
    torch.testing.assert_close(gq.float(), unh(qr.grad, Sq), **t)
    torch.testing.assert_close(gk.float(), unh(kr.grad, Sk), **t)
    torch.testing.assert_close(gv.float(), unh(vr.grad, Sk), **t)


def test_attention_above_32_tokens_dropout_mask_consistency(ops):
    """64 tokens with dropout: the mask is revealed through q = 0, V = one-hot rows (bf16 holds 1 / (64 * 0.9) to 3 digits: a kept
    entry is simply non-zero); forward and both backward kernels must then agree with the f32 math under THAT mask."""
    B, nh, S, p = 2, 3, 64, 0.1
    H = nh * 64
    bf = torch.bfloat16
    z = torch.zeros(B * S, H, device="cuda", dtype=bf)
    eye = torch.zeros(S, 64, device="cuda"); eye[torch.arange(S), torch.arange(S)] = 1
    v1 = eye[None, :, None, :].expand(B, S, nh, 64).reshape(B * S, H).contiguous().to(bf)
    ctx, _ = ops.attn_fwd(z, z, v1, None, B, nh, S, S, False, p, seed=99, site=3)
    keep = (ctx.float().reshape(B, S, nh, 64).permute(0, 2, 1, 3) > 0).float()          # [b, h, i, j]
    assert abs((1 - keep.mean().item()) - p) < 0.02
    torch.manual_seed(5)
    q, k, v, g = (torch.randn(B * S, H, device="cuda").to(bf) for _ in range(4))
    heads = lambda t: t.float().reshape(B, S, nh, 64).transpose(1, 2).detach().clone().requires_grad_(True)
    qr, kr, vr = heads(q), heads(k), heads(v)
    ref = _ref_attention(qr, kr, vr, None, True, keep, p)
    ctx, lse = ops.attn_fwd(q, k, v, None, B, nh, S, S, True, p, seed=99, site=3)
    torch.testing.assert_close(ctx.float(), ref.transpose(1, 2).reshape(B * S, H), **_tol(bf))
    ref.backward(g.float().reshape(B, S, nh, 64).transpose(1, 2))
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attn_bwd(q, k, v, None, g, B, nh, S, S, True, p, 99, 3, gq, gk, gv, ctx=ctx, lse=lse)
    unh = lambda t: t.transpose(1, 2).reshape(B * S, H)
    for got, want in ((gq, qr.grad), (gk, kr.grad), (gv, vr.grad)):
        torch.testing.assert_close(got.float(), unh(want), **_tol(bf))


@pytest.mark.parametrize("amsgrad,wd", [(False, 0.0), (True, 0.01)])
@pytest.mark.parametrize("gdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n", [4096 * 3, 9, 4 * 5 + 3, 2])
def test_adam_matches_torch(ops, amsgrad, wd, gdtype, n):
    """(n = 9: the bias of the reference analysis' 9-code Gumbel projection; sizes that are not multiples of 4 take the scalar tail)"""
    torch.manual_seed(1)
    p = torch.randn(n, device="cuda")
    ref_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3, weight_decay=wd, amsgrad=amsgrad)
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    vmax = torch.zeros(n, device="cuda") if amsgrad else None
    shadow = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for step in range(1, 6):
        g = torch.randn(n, device="cuda").to(gdtype)
        ref_p.grad = g.float().clone()
        opt.step()
        ops.adam_step(p, g, m, v, step, 1e-3, weight_decay=wd, vmax=vmax, shadow=shadow)
        torch.testing.assert_close(p, ref_p.detach(), rtol=2e-6, atol=2e-7)
    assert torch.equal(shadow, p.bfloat16())


def test_dropout_kernel_and_seed_offset(ops):
    """kvq_dropout draws the documented mask (call i/4, 32 bits per element); a device-side seed offset equals adding it on the host."""
    torch.manual_seed(2)
    x = torch.randn(1000, 64, device="cuda").bfloat16()
    y = ops.dropout(x, 0.25, seed=77, site=5)
    kept = y != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    torch.testing.assert_close(y[kept].float(), (x[kept].float() / 0.75).bfloat16().float(), rtol=1e-2, atol=1e-2)
    assert torch.equal(ops.dropout(x, 0.25, seed=77, site=5), y) and not torch.equal(ops.dropout(x, 0.25, seed=78, site=5), y)
    # same mask as the LayerNorm kernel's dropout of (seed, site): LN(dropout(x)) with gamma=1, beta=0 vs LN of y
    st = ops.new_step_state("cuda")
    st[0] = 7
    ops.set_seed_offset(st)
    try:
        y_off = ops.dropout(x, 0.25, seed=70, site=5)
    finally:
        ops.set_seed_offset(None)
    assert torch.equal(y_off, y)                         # 70 + 7 == 77
    assert torch.equal(ops.dropout(x, 0.0, seed=1, site=1), x)


def test_step_state_and_adam_dev(ops):
    torch.manual_seed(3)
    n = 4096
    p = torch.randn(n, device="cuda"); p2 = p.clone()
    g = torch.randn(n, device="cuda")
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); m2 = m.clone(); v2 = v.clone()
    st = ops.new_step_state("cuda")
    for step in range(1, 6):
        ops.step_state_advance(st, 1e-2, 0.1, [2, 4], 0.9, 0.999)
        s, lr, bc1, bc2s = ops.read_step_state(st)
        want_lr = 1e-2 * (0.1 ** sum(1 for ms in (2, 4) if step - 1 >= ms))
        assert s == step
        np.testing.assert_allclose([lr, bc1, bc2s], [want_lr, 1 - 0.9 ** step, (1 - 0.999 ** step) ** 0.5], rtol=2e-5)   # betas are f32
        ops.adam_step_dev(p, g, m, v, st)
        ops.adam_step(p2, g, m2, v2, step, want_lr)
        torch.testing.assert_close(p, p2, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("H,gdt,wdt", [(768, torch.bfloat16, torch.bfloat16), (128, torch.float32, torch.float32), (1024, torch.bfloat16, torch.float32), (100, torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("pattern", ["padded", "one-id", "unique", "blocks"])
def test_embed_grad_matches_index_add(ops, H, gdt, wdt, pattern):
    """Sorted-run embedding gradient vs index_add_ in f64: runs cut by block boundaries (hot pad row), single runs, all-unique
    ids, runs that end exactly on a block boundary; '=' and '+=' forms; out-of-range ids ignored."""
    torch.manual_seed(H)
    V, N = 1000, 2048 + 17
    if pattern == "padded":
        ids = torch.randint(1, V, (N,), device="cuda")
        ids[torch.rand(N, device="cuda") < 0.6] = 0
    elif pattern == "one-id":
        ids = torch.full((N,), 7, device="cuda")
    elif pattern == "unique":
        ids = torch.randperm(N, device="cuda") % V
        ids = torch.arange(N, device="cuda") % V if V >= N else torch.randperm(V, device="cuda").repeat((N + V - 1) // V)[:N]
    else:
        ids = (torch.arange(N, device="cuda") // 32) % V            # every run is exactly one block of 32
        ids[5] = V + 3                                               # ignored
    g = torch.randn(N, H, device="cuda").to(gdt)
    sid, perm = torch.sort(ids.reshape(-1), stable=True)
    ok = ids < V
    ref = torch.zeros(V, H, device="cuda", dtype=torch.float64).index_add_(0, ids[ok], g[ok].double())
    tol = dict(rtol=1e-2, atol=3e-2) if wdt == torch.bfloat16 else dict(rtol=1e-5, atol=1e-4)
    gw = torch.zeros(V, H, device="cuda", dtype=wdt)
    ops.embed_grad(g, perm, sid, gw)
    torch.testing.assert_close(gw.double(), ref, **tol)
    base = torch.randn(V, H, device="cuda").to(wdt)
    gw2 = base.clone()
    ops.embed_grad(g, perm, sid, gw2, accumulate=True)
    torch.testing.assert_close(gw2.double(), base.double() + ref, **(dict(rtol=2e-2, atol=6e-2) if wdt == torch.bfloat16 else tol))
    gw3 = torch.zeros(V, H, device="cuda", dtype=wdt)
    ops.embed_grad(g, perm, sid, gw3)
    assert torch.equal(gw3, gw)                                       # deterministic


def test_embed_grad_full_size_properties(ops):
    """BASELINE configs[1] size (N = 8192 tokens, V = 30522, H = 768, ~60 % pad tokens): linearity (the rows of gW add up to the
    column sums of g), untouched rows stay zero, and equality with an f64 scatter-add."""
    torch.manual_seed(0)
    N, V, H = 8192, 30522, 768
    ids = torch.randint(1000, 30000, (N,), device="cuda")
    ids[torch.rand(N, device="cuda") < 0.6] = 0
    g = torch.randn(N, H, device="cuda").bfloat16()
    sid, perm = torch.sort(ids, stable=True)
    gw = torch.zeros(V, H, device="cuda", dtype=torch.float32)
    ops.embed_grad(g, perm, sid, gw)
    torch.testing.assert_close(gw.sum(0), g.float().sum(0), rtol=1e-4, atol=2e-3)
    used = torch.zeros(V, dtype=torch.bool, device="cuda"); used[ids] = True
    assert torch.all(gw[~used] == 0)
    ref = torch.zeros(V, H, device="cuda", dtype=torch.float64).index_add_(0, ids, g.double())
    torch.testing.assert_close(gw.double(), ref, rtol=1e-5, atol=1e-4)
