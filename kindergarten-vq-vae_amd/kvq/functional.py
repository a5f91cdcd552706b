"""Autograd glue between PyTorch tensors and the C ABI (include/kvq.h).

Boundary mirrored: the tensor-level contract of the reference's VectorQuantizer.forward
(models/shelgon3/VectorQuantizer.py:31-93) and of the loss block of step() (models/shelgon3/Trainer.py:94-101).
All work is enqueued on the current HIP stream; nothing here synchronises with the host.
"""
from __future__ import annotations

import torch

from . import _ffi
from ._ffi import check, io_dtype_of, lib, require_gpu, stream_ptr


def vq_forward_backward_available() -> bool:
    """True when libkvq.so loads and a HIP device is visible (used only for diagnostics, never to pick a fallback)."""
    try:
        lib()
    except _ffi.KvqError:
        return False
    return torch.cuda.is_available()


_WS = {}


def _workspace(dev, nbytes: int, tag=None) -> torch.Tensor:
    """Per-device (and per-stream: `tag`) scratch, grown on demand and kept: size everything once, reuse every step
    (graph-capture safe after the first call at a given shape).  Kernels that run concurrently on different streams must
    not share a scratch buffer, hence the stream tag."""
    if tag is None:
        tag = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, tag)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[key] = ws
    return ws


def _ptr(t):
    return None if t is None else t.data_ptr()


class _VectorQuantize(torch.autograd.Function):
    """z[G,N,D], E[G,K,D] -> loss[G], z_q[G,N,D], perplexity[G], idx[G,N], counts[G,K]."""

    @staticmethod
    def forward(ctx, z, E, beta):
        require_gpu(z, E)
        if not (z.is_contiguous() and E.is_contiguous()):
            raise _ffi.KvqError("vector_quantize: z and E must be contiguous (the reference's .view has the same demand)")
        if E.dtype != torch.float32:
            raise _ffi.KvqError("vector_quantize: the codebook is always float32")
        G, N, D = z.shape
        K = E.shape[1]
        dt = io_dtype_of(z)
        l = lib()
        nbytes = l.kvq_vq_workspace_bytes(N, K, D, G)
        ws = _workspace(z.device, nbytes)
        z_q = torch.empty_like(z)
        idx = torch.empty((G, N), dtype=torch.int64, device=z.device)
        loss = torch.empty((G,), dtype=torch.float32, device=z.device)
        perp = torch.empty((G,), dtype=torch.float32, device=z.device)
        counts = torch.empty((G, K), dtype=torch.float32, device=z.device)
        check(l.kvq_vq_forward(z.data_ptr(), E.data_ptr(), N, K, D, G, dt, float(beta), z_q.data_ptr(), idx.data_ptr(),
                               loss.data_ptr(), perp.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(),
                               stream_ptr()), "kvq_vq_forward")
        ctx.save_for_backward(z, E, idx)
        ctx.beta = float(beta)
        ctx.mark_non_differentiable(perp, idx, counts)
        return loss, z_q, perp, idx, counts

    @staticmethod
    def backward(ctx, g_loss, g_zq, _gp, _gi, _gc):
        z, E, idx = ctx.saved_tensors
        G, N, D = z.shape
        K = E.shape[1]
        l = lib()
        need_z, need_E = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_z = torch.empty_like(z) if need_z else None
        g_E = torch.empty_like(E) if need_E else None
        if g_zq is not None:
            g_zq = g_zq.contiguous()
            if g_zq.dtype != z.dtype:
                g_zq = g_zq.to(z.dtype)
        if g_loss is not None:
            g_loss = g_loss.contiguous().float()
        ws = _workspace(z.device, l.kvq_vq_workspace_bytes(N, K, D, G))
        check(l.kvq_vq_backward(z.data_ptr(), E.data_ptr(), idx.data_ptr(), _ptr(g_zq), _ptr(g_loss), N, K, D, G,
                                io_dtype_of(z), ctx.beta, _ptr(g_z), _ptr(g_E), ws.data_ptr(), ws.numel(), stream_ptr()),
              "kvq_vq_backward")
        return g_z, g_E, None


def vector_quantize(z: torch.Tensor, E: torch.Tensor, beta: float):
    """Fused VQ step.  z: [N,D] or [G,N,D]; E: [K,D] or [G,K,D] (float32).

    Returns (loss, z_q, perplexity, idx, counts) with a leading G axis only if z had one."""
    grouped = z.dim() == 3 and E.dim() == 3
    if not grouped:
        if z.dim() != 2 or E.dim() != 2:
            raise _ffi.KvqError(f"vector_quantize: expected z[N,D], E[K,D] (or grouped [G,..]); got {tuple(z.shape)}, {tuple(E.shape)}")
        z, E = z.unsqueeze(0), E.unsqueeze(0)
    if z.shape[-1] != E.shape[-1] or z.shape[0] != E.shape[0]:
        raise _ffi.KvqError(f"vector_quantize: shape mismatch z{tuple(z.shape)} vs E{tuple(E.shape)}")
    loss, z_q, perp, idx, counts = _VectorQuantize.apply(z, E, beta)
    if not grouped:
        return loss[0], z_q[0], perp[0], idx[0], counts[0]
    return loss, z_q, perp, idx, counts


def vq_one_hot(idx: torch.Tensor, K: int) -> torch.Tensor:
    """min_encodings of the reference: float32 one-hot [N,K] (VectorQuantizer.py:67-68)."""
    require_gpu(idx)
    idx = idx.reshape(-1).contiguous()
    enc = torch.empty((idx.numel(), K), dtype=torch.float32, device=idx.device)
    check(lib().kvq_vq_one_hot(idx.data_ptr(), idx.numel(), K, enc.data_ptr(), stream_ptr()), "kvq_vq_one_hot")
    return enc


def vq_debug_distances(z: torch.Tensor, E: torch.Tensor, use_mfma: bool) -> torch.Tensor:
    """Test hook: the distance matrix as the fused kernel computes it."""
    require_gpu(z, E)
    z = z.reshape(-1, z.shape[-1]).contiguous()
    N, D = z.shape
    K = E.shape[0]
    d = torch.empty((N, K), dtype=torch.float32, device=z.device)
    check(lib().kvq_vq_debug_distances(z.data_ptr(), E.contiguous().data_ptr(), N, K, D, io_dtype_of(z), int(use_mfma),
                                       d.data_ptr(), stream_ptr()), "kvq_vq_debug_distances")
    return d


def vq_ema_update(z, idx, ema_n, ema_m, E, decay: float, eps: float = 1e-5):
    """In-place EMA codebook update (extension, off by default; see include/kvq.h)."""
    require_gpu(z, idx, ema_n, ema_m, E)
    grouped = z.dim() == 3
    G = z.shape[0] if grouped else 1
    N, D = z.shape[-2], z.shape[-1]
    K = E.shape[-2]
    l = lib()
    ws = _workspace(z.device, l.kvq_vq_workspace_bytes(N, K, D, G))
    check(l.kvq_vq_ema_update(z.contiguous().data_ptr(), idx.contiguous().data_ptr(), N, K, D, G, io_dtype_of(z),
                              float(decay), float(eps), ema_n.data_ptr(), ema_m.data_ptr(), E.data_ptr(),
                              ws.data_ptr(), ws.numel(), stream_ptr()), "kvq_vq_ema_update")


def kmeans_update(z: torch.Tensor, idx: torch.Tensor, E: torch.Tensor):
    """In place: E[k] <- mean of z[idx == k] (clusters without points keep their centroid).  Returns the cluster sizes [K] int64."""
    require_gpu(z, idx, E)
    N, D = z.shape
    K = E.shape[0]
    l = lib()
    counts = torch.empty(K, dtype=torch.int64, device=z.device)
    ws = _workspace(z.device, l.kvq_vq_workspace_bytes(N, K, D, 1))
    check(l.kvq_kmeans_update(z.data_ptr(), idx.data_ptr(), N, K, D, io_dtype_of(z), E.data_ptr(), counts.data_ptr(),
                              ws.data_ptr(), ws.numel(), stream_ptr()), "kvq_kmeans_update")
    return counts


def kmeans2_points(z: torch.Tensor, k: int, iters: int = 10, init_indices=None, generator=None):
    """Data-driven codebook initialisation on the GPU: the algorithm of `scipy.cluster.vq.kmeans2(z, k, iter=iters,
    minit='points', missing='warn')`, which the reference runs on the CPU (models/shelgon3/vq_codebook_init_weights.py:85-101).

    z [N, D] (f32 or bf16, on the GPU).  Initial centroids = k distinct rows of z (`init_indices`, or drawn with `generator`);
    every iteration assigns each row to its nearest centroid with the VQ arg-min kernel (kvq_vq_forward) and moves every
    non-empty cluster to the mean of its rows.  Returns (codebook [k, D] f32, labels [N] int64 of the LAST assignment), like
    kmeans2.  Raises KvqError for bad arguments; nothing runs on the CPU."""
    require_gpu(z)
    if z.dim() != 2 or k < 1 or k > z.shape[0] or iters < 1:
        raise _ffi.KvqError(f"kmeans2_points: need z[N,D], 1 <= k <= N, iters >= 1 (got {tuple(z.shape)}, k={k}, iters={iters})")
    z = z.contiguous()
    N = z.shape[0]
    if init_indices is None:
        init_indices = torch.randperm(N, generator=generator)[:k]
    init_indices = torch.as_tensor(init_indices, dtype=torch.int64, device=z.device)
    if init_indices.numel() != k or init_indices.unique().numel() != k:
        raise _ffi.KvqError("kmeans2_points: init_indices must be k distinct row numbers")
    E = z[init_indices].float().contiguous()
    labels = None
    with torch.no_grad():
        for _ in range(iters):
            _loss, _zq, _perp, labels, _cnt = vector_quantize(z, E, 0.0)
            kmeans_update(z, labels, E)
    return E, labels


class _GumbelQuantize(torch.autograd.Function):
    """logits [N,K] -> (y [N,K], diff (0-d), ind [N]); row-wise part of GumbelQuantizer.forward (kvq_gumbel_forward / _backward)."""

    @staticmethod
    def forward(ctx, logits, tau, hard, kld_scale, noise, seed, site):
        require_gpu(logits, noise)
        logits = logits.contiguous()
        N, K = logits.shape
        dev = logits.device
        y = torch.empty_like(logits)
        y_soft = torch.empty((N, K), dtype=torch.float32, device=dev)
        ind = torch.empty(N, dtype=torch.int64, device=dev)
        kl_row = torch.empty(N, dtype=torch.float32, device=dev)
        if noise is not None:
            if noise.shape != logits.shape or noise.dtype != torch.float32:
                raise _ffi.KvqError("gumbel_quantize: noise must be float32 with the shape of logits")
            noise = noise.contiguous()
        check(lib().kvq_gumbel_forward(logits.data_ptr(), _ptr(noise), N, K, float(tau), int(bool(hard)), int(seed), int(site),
                                       io_dtype_of(logits), y.data_ptr(), y_soft.data_ptr(), ind.data_ptr(), kl_row.data_ptr(),
                                       stream_ptr()), "kvq_gumbel_forward")
        diff = kl_row.double().mean().float() * float(kld_scale)          # GumbelQuantizer.py:73 (.mean() over the B*S rows)
        ctx.save_for_backward(logits, y_soft)
        ctx.tau, ctx.kld_scale = float(tau), float(kld_scale)
        ctx.mark_non_differentiable(ind)
        return y, diff, ind

    @staticmethod
    def backward(ctx, g_y, g_diff, _g_ind):
        logits, y_soft = ctx.saved_tensors
        N, K = logits.shape
        g_logits = torch.empty_like(logits)
        gd = g_diff.reshape(1).float().contiguous() if g_diff is not None else torch.zeros(1, device=logits.device)
        gy = g_y.contiguous().to(logits.dtype) if g_y is not None else None
        check(lib().kvq_gumbel_backward(logits.data_ptr(), y_soft.data_ptr(), _ptr(gy), gd.data_ptr(), N, K, ctx.tau, ctx.kld_scale,
                                        io_dtype_of(logits), g_logits.data_ptr(), stream_ptr()), "kvq_gumbel_backward")
        return g_logits, None, None, None, None, None, None


def gumbel_quantize(logits: torch.Tensor, tau: float, hard: bool, kld_scale: float, noise=None, seed: int = 0, site: int = 0):
    """Gumbel-softmax sample over the last dim of logits [N,K] plus the KL-to-uniform term (GumbelQuantizer.py:56-76).

    Returns (y, diff, ind): y the soft sample or, with `hard`, its straight-through one-hot; diff = kld_scale * mean_n
    sum_k q log(q K + 1e-10); ind = argmax.  `noise` (float32 [N,K] Gumbel(0,1) samples) replaces the Philox stream."""
    if logits.dim() != 2:
        raise _ffi.KvqError(f"gumbel_quantize: logits must be [N, K], got {tuple(logits.shape)}")
    return _GumbelQuantize.apply(logits, tau, hard, kld_scale, noise, seed, site)


class _FusedCE(torch.autograd.Function):
    """logits[N,V], target[N] -> loss (mean CE), acc (token accuracy), pred[N].  Backward overwrites `logits`
    with its gradient in place when `inplace_backward` (saves N*V elements; the logits are dead by then)."""

    @staticmethod
    def forward(ctx, logits, target, inplace_backward):
        require_gpu(logits, target)
        if not logits.is_contiguous():
            raise _ffi.KvqError("fused_cross_entropy: logits must be contiguous")
        N, V = logits.shape
        target = target.reshape(-1).contiguous()
        dev = logits.device
        row_loss = torch.empty(N, dtype=torch.float32, device=dev)
        row_lse = torch.empty(N, dtype=torch.float32, device=dev)
        pred = torch.empty(N, dtype=torch.int64, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        check(lib().kvq_ce_forward(logits.data_ptr(), target.data_ptr(), N, V, V, io_dtype_of(logits), row_loss.data_ptr(),
                                   row_lse.data_ptr(), pred.data_ptr(), out[0:].data_ptr(), out[1:].data_ptr(),
                                   stream_ptr()), "kvq_ce_forward")
        ctx.save_for_backward(logits, target, row_lse)
        ctx.inplace = bool(inplace_backward)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc, pred)
        return loss, acc, pred

    @staticmethod
    def backward(ctx, g_loss, _ga, _gp):
        logits, target, row_lse = ctx.saved_tensors
        N, V = logits.shape
        g = logits if ctx.inplace else torch.empty_like(logits)
        g_loss = g_loss.contiguous().float()
        check(lib().kvq_ce_backward(logits.data_ptr(), target.data_ptr(), row_lse.data_ptr(), g_loss.data_ptr(), N, V, V,
                                    io_dtype_of(logits), g.data_ptr(), stream_ptr()), "kvq_ce_backward")
        return g, None, None


def fused_cross_entropy(logits: torch.Tensor, target: torch.Tensor, inplace_backward: bool = False):
    """Trainer.py:94-101 in one pass: (loss_recon, acc_per_batch, recon_ids).  logits [..., V], target [...]."""
    V = logits.shape[-1]
    shape = target.shape
    loss, acc, pred = _FusedCE.apply(logits.reshape(-1, V), target, inplace_backward)
    return loss, acc, pred.reshape(shape)
