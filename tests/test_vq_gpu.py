"""GPU parity tests of the fused VQ step: libkvq.so (HIP, through the C ABI) against the CPU oracle and the
golden vectors of the reference.  Bar: indices and z_q bit-exact; loss/perplexity/gradients to stated tolerance."""
import numpy as np
import pytest
import torch

from _golden_util import case_names, check_flip_budget, check_indices, load_case
from oracle import vq_oracle as O

pytestmark = pytest.mark.gpu

CASES = case_names()


@pytest.fixture(scope="module")
def kvq():
    import kvq as _k
    from kvq import _ffi
    _ffi.lib()   # fails loudly if libkvq.so is missing
    assert torch.cuda.is_available()
    O.build()
    return _k


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def _run(kvq, z, E, beta, g=None, c=1.0):
    zt = _dev(z).reshape(-1, z.shape[-1]).requires_grad_(True)
    Et = _dev(E).requires_grad_(True)
    loss, z_q, perp, idx, counts = kvq.vector_quantize(zt, Et, beta)
    out = dict(loss=loss.item(), z_q=z_q.detach().cpu().numpy(), perplexity=perp.item(), idx=idx.cpu().numpy(),
               counts=counts.cpu().numpy())
    if g is not None:
        (float(c) * loss + (z_q * _dev(g).reshape(z_q.shape)).sum()).backward()
        out["grad_z"] = zt.grad.cpu().numpy()
        out["grad_E"] = Et.grad.cpu().numpy()
    return out


@pytest.mark.parametrize("name", CASES)
def test_forward_bit_exact_vs_oracle_and_reference(kvq, name):
    c = load_case(name)
    D = c["D"]
    got = _run(kvq, c["z"], c["E"], float(c["beta"]))
    ora = O.vq_forward(c["z"], c["E"], float(c["beta"]))
    # HIP == oracle: indices and z_q bit for bit (same summation order by construction)
    assert np.array_equal(got["idx"], ora["idx"]), f"{name}: {(got['idx'] != ora['idx']).sum()} indices differ from the oracle"
    assert np.array_equal(got["z_q"], ora["z_q"].reshape(-1, D))
    assert np.array_equal(got["counts"], ora["counts"])
    np.testing.assert_allclose(got["loss"], ora["loss"], rtol=1e-6)         # f64-accumulated on both sides
    np.testing.assert_allclose(got["perplexity"], ora["perplexity"], rtol=1e-5)
    # HIP == reference golden: exact where the reference itself is stable, minimiser-within-ulps on near ties
    ndiff = check_indices(c, got["idx"])
    check_flip_budget(c, ndiff)
    if c["regime"] != "default_init":
        assert ndiff == 0
    np.testing.assert_allclose(got["loss"], c["loss"], rtol=2e-6)
    if ndiff == 0:
        np.testing.assert_allclose(got["perplexity"], c["perplexity"], rtol=2e-5)
        if c["full"]:
            assert np.array_equal(got["z_q"], c["z_q"].reshape(-1, D))


@pytest.mark.parametrize("name", CASES)
def test_backward_vs_oracle_and_reference(kvq, name):
    c = load_case(name)
    D = c["D"]
    got = _run(kvq, c["z"], c["E"], float(c["beta"]), c["g"], c["c"])
    gz, gE = O.vq_backward(c["z"], c["E"], got["idx"], c["g"], float(c["c"]), float(c["beta"]))
    np.testing.assert_allclose(got["grad_z"], gz.reshape(-1, D), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got["grad_E"], gE, rtol=2e-5, atol=1e-7)
    if np.array_equal(got["idx"], c["idx"]):
        if c["full"]:
            np.testing.assert_allclose(got["grad_z"], c["grad_z"].reshape(-1, D), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(got["grad_E"], c["grad_E"], rtol=2e-5, atol=1e-7)
        else:
            np.testing.assert_allclose(got["grad_z"][c["tok_rows"]], c["grad_z_rows"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(got["grad_E"][c["code_rows"]], c["grad_E_rows"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose((got["grad_E"].astype(np.float64) ** 2).sum(), c["grad_E_sq"], rtol=1e-4, atol=1e-12)
    unused = np.setdiff1d(np.arange(c["K"]), got["idx"])
    assert not got["grad_E"][unused].any()


@pytest.fixture(params=["per_call_pack", "prepacked"])
def variant(request, kvq):
    """The two entry points of the MFMA forward (kvq_vq_forward packs the codebook itself, kvq_vq_forward_packed takes the copy
    kvq_vq_pack_codebook made) must give identical bits."""
    return request.param


def _run_fwd(kvq, z, E, beta, variant):
    if variant == "per_call_pack":
        return _run(kvq, z, E, beta)
    import torch
    from kvq._ffi import check
    from kvq.functional import _workspace
    lib = kvq._ffi.lib()
    zt, Et = _dev(z.reshape(-1, z.shape[-1])), _dev(E)
    N, D = zt.shape
    K = Et.shape[0]
    pk = torch.empty(lib.kvq_vq_packed_bytes(K, D, 1), dtype=torch.uint8, device="cuda")
    check(lib.kvq_vq_pack_codebook(Et.data_ptr(), K, D, 1, pk.data_ptr(), None), "pack")
    z_q = torch.empty_like(zt); idx = torch.empty(N, dtype=torch.int64, device="cuda")
    out = torch.empty(2, dtype=torch.float32, device="cuda"); cnt = torch.empty(K, dtype=torch.float32, device="cuda")
    ws = _workspace(zt.device, lib.kvq_vq_workspace_bytes(N, K, D, 1))
    check(lib.kvq_vq_forward_packed(zt.data_ptr(), Et.data_ptr(), pk.data_ptr(), N, K, D, 1, 0, float(beta), z_q.data_ptr(), idx.data_ptr(),
                                    out[0:].data_ptr(), out[1:].data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(), None), "fwd")
    torch.cuda.synchronize()
    return dict(idx=idx.cpu().numpy(), z_q=z_q.cpu().numpy(), counts=cnt.cpu().numpy(), loss=out[0].item(), perplexity=out[1].item())


@pytest.mark.parametrize("name", ["c1_sep", "c1_default", "k8192_sep", "demo_default"])
def test_forward_both_variants_vs_oracle(kvq, variant, name):
    c = load_case(name)
    got = _run_fwd(kvq, c["z"], c["E"], float(c["beta"]), variant)
    ora = O.vq_forward(c["z"], c["E"], float(c["beta"]))
    assert np.array_equal(got["idx"], ora["idx"]) and np.array_equal(got["z_q"], ora["z_q"].reshape(-1, c["D"]))
    assert np.array_equal(got["counts"], ora["counts"])
    np.testing.assert_allclose(got["loss"], ora["loss"], rtol=1e-6)


@pytest.mark.parametrize("shape", [(64, 512, 768), (100, 300, 128), (33, 37, 64), (256, 8192, 64)])
@pytest.mark.parametrize("regime", ["sep", "near_tie"])
def test_mfma_distances_bitwise_equal_oracle(kvq, shape, regime):
    """The f32 MFMA contraction must be the documented fmaf chain: distance matrices equal bit for bit."""
    N, K, D = shape
    rng = np.random.default_rng(N + K + D)
    z = rng.standard_normal((N, D), dtype=np.float32)
    E = rng.standard_normal((K, D), dtype=np.float32) if regime == "sep" else \
        rng.uniform(-1.0 / K, 1.0 / K, (K, D)).astype(np.float32)
    assert kvq._ffi.lib().kvq_vq_uses_mfma(N, K, D) == 1
    d_ref = O.distances(z, E)
    d_mfma = kvq.vq_debug_distances(_dev(z), _dev(E), use_mfma=True).cpu().numpy()
    d_gen = kvq.vq_debug_distances(_dev(z), _dev(E), use_mfma=False).cpu().numpy()
    assert np.array_equal(d_gen.view(np.uint32), d_ref.view(np.uint32)), "generic kernel != oracle"
    assert np.array_equal(d_mfma.view(np.uint32), d_ref.view(np.uint32)), "MFMA kernel != oracle"


def test_ties_first_index_and_nan_rule(kvq):
    c = load_case("tiny_ties")
    got = _run(kvq, c["z"], c["E"], 0.25)
    assert (got["idx"] < c["K"] - c["K"] // 2).all()
    # big duplicated codebook on the MFMA path: duplicates sit in other waves / other passes
    rng = np.random.default_rng(3)
    E = rng.standard_normal((300, 64), dtype=np.float32)
    E = np.concatenate([E, E, E])                    # K = 900: 4 passes of 256, copies 300 and 600 apart
    z = rng.standard_normal((70, 64), dtype=np.float32)
    got = _run(kvq, z, E, 0.25)
    assert (got["idx"] < 300).all()
    assert np.array_equal(got["idx"], O.vq_forward(z, E, 0.25)["idx"])
    z[5, 3] = np.nan                                  # NaN distance row: torch.argmin returns the first index
    got = _run(kvq, z, E, 0.25)
    assert got["idx"][5] == 0


def test_bf16_io_matches_oracle_on_upcast_inputs(kvq):
    """bf16 activations are upcast exactly, all arithmetic stays f32: indices equal the oracle on the upcast z."""
    rng = np.random.default_rng(8)
    N, K, D = 256, 512, 768
    z = torch.from_numpy(rng.standard_normal((N, D), dtype=np.float32)).bfloat16()
    E = rng.standard_normal((K, D), dtype=np.float32)
    zt = z.cuda().requires_grad_(True)
    Et = _dev(E).requires_grad_(True)
    loss, z_q, perp, idx, counts = kvq.vector_quantize(zt, Et, 0.25)
    ora = O.vq_forward(z.float().numpy(), E, 0.25)
    assert np.array_equal(idx.cpu().numpy(), ora["idx"])
    assert z_q.dtype == torch.bfloat16
    assert torch.equal(z_q.cpu(), torch.from_numpy(ora["z_q"]).bfloat16())
    np.testing.assert_allclose(loss.item(), ora["loss"], rtol=1e-4)   # north_star: bf16 losses within 1e-4 relative
    g = torch.from_numpy(rng.standard_normal((N, D), dtype=np.float32)).bfloat16()
    (loss + (z_q.float() * g.cuda().float()).sum()).backward()
    gz, gE = O.vq_backward(z.float().numpy(), E, ora["idx"], g.float().numpy(), 1.0, 0.25)
    np.testing.assert_allclose(zt.grad.float().cpu().numpy(), gz, rtol=1e-2, atol=1e-6)
    np.testing.assert_allclose(Et.grad.cpu().numpy(), gE, rtol=2e-5, atol=1e-7)


def test_grouped_codebooks_equal_independent_calls(kvq):
    """G codebooks in one launch == G separate calls."""
    rng = np.random.default_rng(21)
    G, N, K, D = 3, 96, 128, 64
    z = rng.standard_normal((G, N, D), dtype=np.float32)
    E = rng.standard_normal((G, K, D), dtype=np.float32)
    loss, z_q, perp, idx, counts = kvq.vector_quantize(_dev(z), _dev(E), 0.3)
    for g in range(G):
        ora = O.vq_forward(z[g], E[g], 0.3)
        assert np.array_equal(idx[g].cpu().numpy(), ora["idx"])
        assert np.array_equal(z_q[g].cpu().numpy(), ora["z_q"])
        np.testing.assert_allclose(loss[g].item(), ora["loss"], rtol=1e-6)
        np.testing.assert_allclose(perp[g].item(), ora["perplexity"], rtol=1e-5)


def test_full_size_properties_c2(kvq):
    """BASELINE config 2 shape (N=8192, K=512, D=768): size-independent properties instead of a CPU recompute."""
    torch.manual_seed(0)
    N, K, D = 8192, 512, 768
    z = torch.randn(N, D, device="cuda")
    E = torch.randn(K, D, device="cuda")
    loss, z_q, perp, idx, counts = kvq.vector_quantize(z, E, 0.25)
    e = E[idx]
    assert torch.equal(z_q, z + (e - z))                                     # straight-through value identity
    np.testing.assert_allclose(loss.item(), 1.25 * ((e - z).double() ** 2).mean().item(), rtol=1e-6)
    assert counts.sum().item() == N and torch.equal(counts, torch.bincount(idx, minlength=K).float())
    # idempotence: quantising the codes themselves returns each code's own (first) index with zero loss
    l2, zq2, _, idx2, _ = kvq.vector_quantize(E.clone(), E, 0.25)
    assert torch.equal(idx2, torch.arange(K, device="cuda")) and l2.item() == 0.0 and torch.equal(zq2, E)
    # optimality: no code is closer (fp64 check on a token sample)
    sel = torch.randperm(N, device="cuda")[:256]
    d = torch.cdist(z[sel].double(), E.double()) ** 2
    chosen = d.gather(1, idx[sel, None]).squeeze(1)
    assert (chosen - d.min(1).values <= 1e-3).all()
    # permutation equivariance over tokens
    perm = torch.randperm(N, device="cuda")
    _, _, _, idx_p, _ = kvq.vector_quantize(z[perm].contiguous(), E, 0.25)
    assert torch.equal(idx_p, idx[perm])
    # determinism (no float atomics): two runs are bitwise equal, backward included
    zr = z.clone().requires_grad_(True); Er = E.clone().requires_grad_(True)
    la, zqa, *_ = kvq.vector_quantize(zr, Er, 0.25); (la + zqa.sum()).backward()
    zr2 = z.clone().requires_grad_(True); Er2 = E.clone().requires_grad_(True)
    lb, zqb, *_ = kvq.vector_quantize(zr2, Er2, 0.25); (lb + zqb.sum()).backward()
    assert la.item() == lb.item() and torch.equal(Er.grad, Er2.grad) and torch.equal(zr.grad, zr2.grad)


def test_large_codebook_c4(kvq):
    """BASELINE config 4 (K=8192): 32 passes over the codebook; compare a token sample with the oracle."""
    rng = np.random.default_rng(4)
    N, K, D = 2048, 8192, 768
    z = rng.standard_normal((N, D), dtype=np.float32)
    E = rng.standard_normal((K, D), dtype=np.float32)
    loss, z_q, perp, idx, counts = kvq.vector_quantize(_dev(z), _dev(E), 0.25)
    O.set_threads(8)
    ora = O.vq_forward(z[:128], E, 0.25)
    O.set_threads(1)
    assert np.array_equal(idx[:128].cpu().numpy(), ora["idx"])
    assert counts.sum().item() == N


def test_k8192_default_indices_equal_the_reference(kvq):
    """BASELINE.json configs[3] in the near-tie regime (golden case k8192_default: N = 8192, K = 8192, default-init codebook,
    produced by the reference module): ALL 8192 indices of the HIP kernel against the reference's, flip budget asserted;
    bit for bit against the oracle on every token (the oracle runs on all host threads: 51 GFLOP of scalar f32)."""
    import os
    c = load_case("k8192_default")
    D = c["D"]
    got = _run(kvq, c["z"], c["E"], float(c["beta"]), c["g"], c["c"])
    ndiff = check_indices(c, got["idx"])
    check_flip_budget(c, ndiff)
    np.testing.assert_allclose(got["loss"], c["loss"], rtol=2e-6)
    O.set_threads(min(os.cpu_count() or 1, 16))
    try:
        ora = O.vq_forward(c["z"], c["E"], float(c["beta"]))
    finally:
        O.set_threads(1)
    assert np.array_equal(got["idx"], ora["idx"]), f"{(got['idx'] != ora['idx']).sum()} indices differ from the oracle"
    assert np.array_equal(got["z_q"], ora["z_q"].reshape(-1, D)) and np.array_equal(got["counts"], ora["counts"])
    np.testing.assert_allclose(got["perplexity"], ora["perplexity"], rtol=1e-6)
    if ndiff == 0:
        np.testing.assert_allclose(got["perplexity"], c["perplexity"], rtol=2e-5)
        np.testing.assert_allclose(got["grad_z"][c["tok_rows"]], c["grad_z_rows"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(got["grad_E"][c["code_rows"]], c["grad_E_rows"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose((got["grad_E"].astype(np.float64) ** 2).sum(), c["grad_E_sq"], rtol=1e-4)


def test_module_surface_matches_reference(kvq):
    """nn.Module drop-in: ctor, parameter name, 5-tuple, shapes and dtypes (VectorQuantizer.py:19-29,:93)."""
    from models.shelgon3.VectorQuantizer import VectorQuantizer
    torch.manual_seed(0)
    vq = VectorQuantizer(n_e=10, e_dim=768, beta=0.69).cuda()
    assert list(vq.state_dict().keys()) == ["embedding.weight"] and len(list(vq.buffers())) == 0
    assert vq.embedding.weight.abs().max().item() <= 0.1
    z = torch.rand(16, 7, 768, device="cuda", requires_grad=True)          # the reference's demo shape (:100-110)
    loss, z_q, perp, enc, idx = vq.forward(z, "cuda")
    assert loss.dim() == 0 and perp.dim() == 0
    assert z_q.shape == z.shape and enc.shape == (112, 10) and enc.dtype == torch.float32
    assert idx.shape == (16, 7, 1) and idx.dtype == torch.int64
    assert torch.equal(enc.argmax(1), idx.reshape(-1)) and torch.equal(enc.sum(1), torch.ones(112, device="cuda"))
    (loss + z_q.sum()).backward()
    assert z.grad is not None and vq.embedding.weight.grad is not None
    init = torch.randn(10, 768)
    vq2 = VectorQuantizer(10, 768, 0.25, vq_codebook_init_values=init)
    assert torch.equal(vq2.embedding.weight.data, init)
    with pytest.raises(RuntimeError):
        vq.forward(torch.rand(4, 768, 7, device="cuda").transpose(1, 2), "cuda")   # non-contiguous, like .view
    with pytest.raises(RuntimeError):
        vq.forward(torch.rand(4, 7, 100, device="cuda"), "cuda")
    from kvq._ffi import KvqError
    with pytest.raises(KvqError):
        vq.cpu().forward(torch.rand(2, 3, 768), "cpu")                               # no CPU fallback, by design


def test_ema_update_matches_textbook_oracle(kvq):
    rng = np.random.default_rng(2)
    N, K, D = 500, 32, 64
    z = rng.standard_normal((N, D), dtype=np.float32)
    E = rng.standard_normal((K, D), dtype=np.float32)
    idx = O.vq_forward(z, E, 0.25)["idx"]
    ema_n = np.ones(K, np.float32); ema_m = E.copy()
    n2, m2, E2 = O.vq_ema_update(z, idx, 0.99, 1e-5, ema_n, ema_m, E)
    tn, tm, tE = _dev(ema_n), _dev(ema_m), _dev(E)
    kvq.vq_ema_update(_dev(z), _dev(idx), tn, tm, tE, 0.99, 1e-5)
    np.testing.assert_allclose(tn.cpu().numpy(), n2, rtol=1e-6)
    np.testing.assert_allclose(tm.cpu().numpy(), m2, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tE.cpu().numpy(), E2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,K,D", [(8192, 512, 768), (5000, 10, 768), (700, 37, 40)])
def test_collapsed_codebook(kvq, N, K, D):
    """Codebook collapse (every token on one or two codes) is the common failure mode of VQ training and the
    worst case for histogram / scatter contention: results must not change, only (at most) speed."""
    rng = np.random.default_rng(N + K)
    E = rng.standard_normal((K, D), dtype=np.float32)
    hot = np.where(rng.random(N) < 0.97, 3, K - 1)
    z = (E[hot] + 0.05 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
    g = rng.standard_normal((N, D), dtype=np.float32)
    got = _run(kvq, z, E, 0.25, g, 1.3)
    assert np.array_equal(got["idx"], hot)
    assert got["counts"][3] == (hot == 3).sum() and got["counts"].sum() == N
    O.set_threads(8)
    ora = O.vq_forward(z, E, 0.25)
    O.set_threads(1)
    assert np.array_equal(got["idx"], ora["idx"]) and np.array_equal(got["z_q"], ora["z_q"])
    np.testing.assert_allclose(got["loss"], ora["loss"], rtol=1e-6)
    gz, gE = O.vq_backward(z, E, got["idx"], g, 1.3, 0.25)
    np.testing.assert_allclose(got["grad_z"], gz, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got["grad_E"], gE, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,K,D", [(4096, 9, 768), (3000, 64, 128), (513, 5, 40)])
def test_kmeans2_points_matches_scipy(N, K, D, dtype):
    """GPU k-means (VQ arg-min kernel + kvq_kmeans_update) vs scipy.cluster.vq.kmeans2 -- the routine the reference calls
    (vq_codebook_init_weights.py:91) -- started from the same points: same labels, same centroids, empty clusters kept."""
    from scipy.cluster.vq import kmeans2
    from kvq import functional as KF
    rng = np.random.default_rng(N + K)
    centers = rng.normal(size=(K, D)).astype(np.float32) * 3
    data = (centers[rng.integers(0, K, size=N)] + rng.normal(size=(N, D)).astype(np.float32))
    z = torch.from_numpy(data).cuda().to(dtype)
    data = z.float().cpu().numpy()                       # cluster exactly what the GPU sees
    init = rng.choice(N, size=K, replace=False)
    cb_ref, lab_ref = kmeans2(data.astype(np.float64), data[init].astype(np.float64), iter=10, minit="matrix", missing="warn")
    cb, lab = KF.kmeans2_points(z, K, iters=10, init_indices=torch.from_numpy(init))
    assert cb.dtype == torch.float32 and lab.dtype == torch.int64 and lab.shape == (N,)
    agree = (lab.cpu().numpy() == lab_ref).mean()
    assert agree > 0.999, agree                          # a point on a bisector may round either way
    np.testing.assert_allclose(cb.cpu().numpy(), cb_ref, rtol=2e-3, atol=2e-3)


def test_kmeans_update_keeps_empty_clusters_and_counts():
    from kvq import functional as KF
    torch.manual_seed(0)
    z = torch.randn(1000, 64, device="cuda")
    E = torch.randn(4, 64, device="cuda")
    E0 = E.clone()
    idx = torch.randint(0, 3, (1000,), device="cuda")     # cluster 3 never used
    counts = KF.kmeans_update(z, idx, E)
    assert counts.tolist() == [int((idx == k).sum()) for k in range(4)] and counts[3] == 0
    assert torch.equal(E[3], E0[3])
    for k in range(3):
        torch.testing.assert_close(E[k], z[idx == k].double().mean(0).float(), rtol=1e-5, atol=1e-6)
    from kvq._ffi import KvqError
    with pytest.raises(KvqError):
        KF.kmeans2_points(z, 2000)
    with pytest.raises(KvqError):
        KF.kmeans2_points(z.cpu(), 4)


def test_nine_codebooks_full_size_equals_separate_launches(kvq):
    """BASELINE configs[4] shape of the quantiser: 9 factor codebooks x K = 512, D = 768, N = 8192 bf16 tokens each.
    One grouped launch must equal nine single-codebook launches bit for bit (indices, z_q, losses, histograms, gradients)."""
    torch.manual_seed(9)
    G, N, K, D = 9, 8192, 512, 768
    z = torch.randn(G, N, D, device="cuda").bfloat16().requires_grad_(True)
    E = torch.randn(G, K, D, device="cuda").requires_grad_(True)
    loss, z_q, perp, idx, counts = kvq.vector_quantize(z, E, 0.25)
    assert idx.shape == (G, N) and z_q.shape == (G, N, D) and counts.shape == (G, K)
    gq = torch.randn(G, N, D, device="cuda").bfloat16()
    (loss.sum() + (z_q.float() * gq.float()).sum()).backward()
    for g in range(G):
        zg = z[g].detach().clone().requires_grad_(True); Eg = E[g].detach().clone().requires_grad_(True)
        l1, zq1, p1, i1, c1 = kvq.vector_quantize(zg, Eg, 0.25)
        assert torch.equal(i1, idx[g]) and torch.equal(zq1, z_q[g]) and torch.equal(c1, counts[g])
        assert l1.item() == loss[g].item() and p1.item() == perp[g].item()
        (l1 + (zq1.float() * gq[g].float()).sum()).backward()
        assert torch.equal(zg.grad, z.grad[g]) and torch.equal(Eg.grad, E.grad[g])
    assert counts.sum().item() == G * N


@pytest.mark.parametrize("G,N,K,D,dtype", [(1, 8192, 512, 768, torch.bfloat16), (1, 8192, 8192, 768, torch.bfloat16),
                                           (1, 5000, 300, 128, torch.float32), (9, 1000, 512, 96, torch.bfloat16),
                                           (1, 33, 37, 64, torch.float32), (3, 8192, 64, 768, torch.bfloat16)])
def test_fused_tail_equals_the_three_kernel_forward(kvq, G, N, K, D, dtype):
    """Round 5: the distance kernel's last-arriving workgroups run the epilogue and the final sums (one launch).  The three-kernel
    forward of rounds 1 - 4 (kvq_vq_set_variant(0): distance / epilogue / finalize) is its checker: indices, z_q and the
    histogram bit for bit, loss and perplexity to f64-summation-order rounding, on full-size, ragged (N % 64 != 0, K % 128 != 0),
    grouped and collapsed-histogram inputs -- twice, so that a stale arrival ticket of the first call would show in the second."""
    from kvq import _ffi
    lib = _ffi.lib()
    torch.manual_seed(N + K)
    z = torch.randn(G, N, D, device="cuda").to(dtype)
    E = torch.randn(G, K, D, device="cuda")
    if K == 64:
        E[:, 1:] += 50.0                                      # (nearly) every token on code 0: one histogram address takes all adds
    runs = {}
    try:
        for fused in (1, 0, 1):
            _ffi.check(lib.kvq_vq_set_variant(fused), "kvq_vq_set_variant")
            loss, z_q, perp, idx, counts = kvq.vector_quantize(z, E, 0.25)
            torch.cuda.synchronize()
            got = (idx.clone(), z_q.clone(), counts.clone(), loss.clone(), perp.clone())
            if fused in runs:                                 # the second fused call against the first: bitwise, scalars included
                assert all(torch.equal(a, b) for a, b in zip(got, runs[fused]))
            runs[fused] = got
    finally:
        _ffi.check(lib.kvq_vq_set_variant(1), "kvq_vq_set_variant")
    (i1, q1, c1, l1, p1), (i0, q0, c0, l0, p0) = runs[1], runs[0]
    assert torch.equal(i1, i0) and torch.equal(q1, q0) and torch.equal(c1, c0)
    assert c1.sum().item() == G * N
    torch.testing.assert_close(l1, l0, rtol=1e-6, atol=0)
    torch.testing.assert_close(p1, p0, rtol=1e-6, atol=0)
