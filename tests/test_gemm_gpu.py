"""The hand-written bf16 MFMA GEMM (csrc/kvq_gemm.hip) against an f32 torch reference, all epilogue forms, ragged sizes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from kvq import _ffi, nnops
    _ffi.lib()
    return nnops


@pytest.mark.parametrize("M,N,K", [(8192, 768, 768), (256, 2304, 768), (1000, 3072, 64), (384, 30528, 768), (130, 72, 192), (128, 128, 3072)])
def test_gemm_nt_matches_f32_reference(ops, M, N, K):
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    ref = torch.addmm(bias.float(), a.float(), w.float().t())
    out = ops.gemm_nt(a, w, bias)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)           # bf16 output rounding
    # same accumulation quality as the library GEMM: compare with torch's bf16 result too
    torch.testing.assert_close(out.float(), torch.addmm(bias, a, w.t()).float(), rtol=2e-2, atol=2e-2)
    c0 = torch.randn(M, N, device="cuda").bfloat16()
    c1 = c0.clone()
    ops.gemm_nt(a, w, None, out=c1, accumulate=True)
    torch.testing.assert_close(c1.float(), c0.float() + a.float() @ w.float().t(), rtol=2e-2, atol=3e-2)


def test_gemm_strided_operands(ops):
    """Row strides larger than the logical width (views into fused [N,3H] buffers)."""
    torch.manual_seed(1)
    big = torch.randn(512, 2304, device="cuda").bfloat16()
    a = big[:, 768:1536]                                     # ld = 2304
    w = torch.randn(768, 768, device="cuda").bfloat16() * 0.05
    outbuf = torch.zeros(512, 1536, device="cuda").bfloat16()
    ops.gemm_nt(a, w, None, out=outbuf[:, :768])
    torch.testing.assert_close(outbuf[:, :768].float(), a.float() @ w.float().t(), rtol=2e-2, atol=2e-2)
    assert not outbuf[:, 768:].any()


def test_gemm_fused_gelu_epilogues(ops):
    torch.manual_seed(2)
    M = 640
    x = torch.randn(M, 768, device="cuda").bfloat16()
    w1 = (torch.randn(3072, 768, device="cuda") * 0.04).bfloat16()
    b1 = torch.randn(3072, device="cuda").bfloat16()
    h, a = ops.gemm_nt_gelu(x, w1, b1)
    href = torch.addmm(b1.float(), x.float(), w1.float().t())
    torch.testing.assert_close(h.float(), href, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(a.float(), F.gelu(h.float()), rtol=1e-2, atol=1e-2)      # gelu of the STORED pre-activation
    gf = torch.randn(M, 768, device="cuda").bfloat16()
    w2t = (torch.randn(3072, 768, device="cuda") * 0.04).bfloat16()
    g = ops.gemm_nt_dgelu(gf, w2t, h)
    hr = h.float().requires_grad_(True)
    F.gelu(hr).backward(gf.float() @ w2t.float().t())
    torch.testing.assert_close(g.float(), hr.grad, rtol=3e-2, atol=3e-2)


def test_gemm_rejects_unsupported_shapes(ops):
    from kvq._ffi import KvqError
    a = torch.randn(64, 100, device="cuda").bfloat16()
    w = torch.randn(64, 100, device="cuda").bfloat16()
    with pytest.raises(KvqError):
        ops.gemm_nt(a, w)                                    # K % 64 != 0


def test_transpose_batch():
    from kvq import nnops
    torch.manual_seed(0)
    srcs = [torch.randn(768, 768, device="cuda").bfloat16() for _ in range(70)]          # > KVQ_TRANSPOSE_MAX: two launches
    dsts = [torch.empty(768, 768, device="cuda", dtype=torch.bfloat16) for _ in srcs]
    op = nnops.TransposeBatch(srcs, dsts)
    op.run()
    assert all(torch.equal(d, s.t()) for s, d in zip(srcs, dsts))
    a = torch.randn(100, 36, device="cuda").bfloat16(); b = torch.empty(36, 100, device="cuda", dtype=torch.bfloat16)
    nnops.TransposeBatch([a], [b]).run()
    assert torch.equal(b, a.t())
