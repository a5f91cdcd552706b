"""GPU parity of the fused reconstruction loss (Trainer.py:94-101) against the CPU oracle and torch's own ops."""
import numpy as np
import pytest
import torch

from oracle import vq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kvq():
    import kvq as _k
    from kvq import _ffi
    _ffi.lib()
    O.build()
    return _k


@pytest.mark.parametrize("N,V", [(7, 13), (40, 1000), (64, 30522), (3, 30523), (5, 8), (2, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ce_forward_backward_vs_oracle(kvq, N, V, dtype):
    rng = np.random.default_rng(N * 1000 + V)
    x = (4.0 * rng.standard_normal((N, V))).astype(np.float32)
    t = rng.integers(0, V, N)
    xt = torch.from_numpy(x).to(dtype)
    x_up = xt.float().numpy()                       # what the kernel actually sees
    ora = O.ce_forward(x_up, t)
    lg = xt.cuda().requires_grad_(True)
    loss, acc, pred = kvq.fused_cross_entropy(lg, torch.from_numpy(t).cuda())
    np.testing.assert_allclose(loss.item(), ora["loss"], rtol=2e-6, atol=1e-6)
    assert np.array_equal(pred.cpu().numpy(), ora["pred"])            # first maximum, like torch.argmax
    np.testing.assert_allclose(acc.item(), ora["acc"], rtol=1e-6)
    (loss * 1.7).backward()
    g = O.ce_backward(x_up, t, 1.7)
    tol = dict(rtol=2e-5, atol=1e-8) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-6)
    np.testing.assert_allclose(lg.grad.float().cpu().numpy(), g, **tol)


def test_ce_equals_reference_expression(kvq):
    """The exact expression of Trainer.py:94-101 on torch ops (one-hot KL, argmax of softmax, seq_acc)."""
    torch.manual_seed(1)
    B, S, V = 6, 12, 30522
    logits = (3 * torch.randn(B, S, V)).cuda()
    ids = torch.randint(0, V, (B, S)).cuda()
    ids[:, 8:] = 0                                                       # padding tokens are scored too
    ref = torch.nn.functional.kl_div(
        input=torch.log_softmax(logits.reshape(-1, V), dim=-1),
        target=torch.nn.functional.one_hot(ids, V).reshape(-1, V).float(), reduction="batchmean")
    recon = torch.argmax(torch.softmax(logits, dim=-1), dim=-1)
    loss, acc, pred = kvq.fused_cross_entropy(logits, ids)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
    assert torch.equal(pred, recon)
    np.testing.assert_allclose(acc.item(), (recon == ids).float().mean().item(), rtol=1e-6)


def test_ce_inplace_backward(kvq):
    torch.manual_seed(2)
    N, V = 33, 30522
    x = torch.randn(N, V, device="cuda", dtype=torch.bfloat16)
    t = torch.randint(0, V, (N,), device="cuda")
    a = x.clone().requires_grad_(True)
    la, _, _ = kvq.fused_cross_entropy(a, t)
    la.backward()
    lin = torch.nn.Linear(8, V).cuda().bfloat16()
    h = torch.randn(N, 8, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    with torch.no_grad():
        lin.weight.zero_(); lin.bias.zero_()
    y = lin(h) + x                                                       # a non-leaf logits tensor
    lb, _, _ = kvq.fused_cross_entropy(y, t, inplace_backward=True)
    lb.backward()
    assert torch.equal(lin.bias.grad, a.grad.float().sum(0).bfloat16()) or \
        torch.allclose(lin.bias.grad.float(), a.grad.float().sum(0), rtol=2e-2, atol=1e-4)


@pytest.mark.parametrize("N,V,ld", [(100, 1000, 1000), (77, 30522, 30528), (33, 13, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ce_backward_with_bias_partials_equals_plain_backward(kvq, N, V, ld, dtype):
    """The row-block-tiled backward (kvq_ce_backward_bias): same gradient as the per-row kernel, zero padding columns, and
    partial rows that sum to the column sums of the stored gradient (the LM-head bias gradient)."""
    from kvq._ffi import check, lib, stream_ptr
    torch.manual_seed(N + V)
    io = 0 if dtype == torch.float32 else 1
    logits = (3 * torch.randn(N, ld, device="cuda")).to(dtype)
    tgt = torch.randint(0, V, (N,), device="cuda")
    rl = torch.empty(N, device="cuda"); lse = torch.empty(N, device="cuda"); pred = torch.empty(N, dtype=torch.int64, device="cuda")
    check(lib().kvq_ce_forward(logits.data_ptr(), tgt.data_ptr(), N, V, ld, io, rl.data_ptr(), lse.data_ptr(), pred.data_ptr(), None, None, stream_ptr()), "fwd")
    gs = torch.full((1,), 1.3, device="cuda")
    g0 = torch.empty_like(logits)
    check(lib().kvq_ce_backward(logits.data_ptr(), tgt.data_ptr(), lse.data_ptr(), gs.data_ptr(), N, V, ld, io, g0.data_ptr(), stream_ptr()), "bwd")
    P = lib().kvq_ce_bwd_partial_rows(N)
    part = torch.full((P, ld), 7.0, device="cuda")
    g1 = logits.clone()                                              # in place, as the engine runs it
    check(lib().kvq_ce_backward_bias(g1.data_ptr(), tgt.data_ptr(), lse.data_ptr(), gs.data_ptr(), N, V, ld, io, g1.data_ptr(),
                                     part.data_ptr(), part.numel() * 4, stream_ptr()), "bwd_bias")
    torch.testing.assert_close(g1.float(), g0.float(), rtol=1e-5, atol=1e-9)
    assert torch.all(g1[:, V:] == 0) and torch.all(part[:, V:] == 0)
    torch.testing.assert_close(part.sum(0), g1.float().sum(0), rtol=1e-4, atol=1e-6)
    assert lib().kvq_ce_backward_bias(g1.data_ptr(), tgt.data_ptr(), lse.data_ptr(), gs.data_ptr(), N, V, ld, io, g1.data_ptr(),
                                      part.data_ptr(), 16, stream_ptr()) != 0          # partial buffer too small
