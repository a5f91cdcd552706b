"""Tensor-level wrappers of the fused block kernels in libkvq.so (csrc/kvq_nn.hip).  Plain functions, no autograd:
the training engine (kvq/engine.py) calls forward and backward kernels explicitly.  All work goes to the current stream."""
from __future__ import annotations

import torch

from ._ffi import KvqError, check, io_dtype_of, lib, require_gpu, stream_ptr
from .functional import _workspace


def _p(t):
    return None if t is None else t.data_ptr()


def ln_fwd(y, resid, gamma, beta, eps, p_drop=0.0, seed=0, site=0, save_pre=True):
    """out = LayerNorm(dropout(y) + resid).  Returns (out, pre, mean, rstd); pre = the LN input as stored."""
    require_gpu(y, gamma, beta)
    N, H = y.shape
    out = torch.empty_like(y)
    pre = torch.empty_like(y) if save_pre else None
    mean = torch.empty(N, dtype=torch.float32, device=y.device)
    rstd = torch.empty(N, dtype=torch.float32, device=y.device)
    check(lib().kvq_dropout_residual_ln_fwd(y.data_ptr(), _p(resid), gamma.data_ptr(), beta.data_ptr(), N, H, float(eps),
                                            float(p_drop), int(seed), int(site), io_dtype_of(y), out.data_ptr(), _p(pre),
                                            mean.data_ptr(), rstd.data_ptr(), stream_ptr()), "kvq_dropout_residual_ln_fwd")
    return out, pre, mean, rstd


def ln_fwd_fp8(y, resid, gamma, beta, eps, p_drop, seed, site, state, save_pre=True):
    """ln_fwd plus the fp8 (e4m3) copy of `out` for the fp8 GEMM that reads it next: (out, pre, mean, rstd, out8).  out8 holds the
    bytes kvq_fp8_quantize_delayed(out, state) would write; the call's amax is noted in `state` (one delayed-scaling record)."""
    require_gpu(y, gamma, beta)
    N, H = y.shape
    assert y.dtype == torch.bfloat16
    out = torch.empty_like(y)
    out8 = torch.empty((N, H), dtype=torch.uint8, device=y.device)
    pre = torch.empty_like(y) if save_pre else None
    mean = torch.empty(N, dtype=torch.float32, device=y.device)
    rstd = torch.empty(N, dtype=torch.float32, device=y.device)
    check(lib().kvq_dropout_residual_ln_fwd_fp8(y.data_ptr(), _p(resid), gamma.data_ptr(), beta.data_ptr(), N, H, float(eps),
                                                float(p_drop), int(seed), int(site), out.data_ptr(), _p(pre), mean.data_ptr(),
                                                rstd.data_ptr(), out8.data_ptr(), state.data_ptr(), stream_ptr()),
          "kvq_dropout_residual_ln_fwd_fp8")
    return out, pre, mean, rstd, out8


def ln_bwd(g_out, pre, mean, rstd, gamma, p_drop=0.0, seed=0, site=0, g_gamma=None, g_beta=None, accumulate=False,
           need_g_y=True, need_g_resid=True, g_bias_prev=None):
    """Returns (g_y, g_resid).  g_gamma / g_beta (f32 or bf16 [H]) are written (or accumulated into) when given;
    g_bias_prev receives colsum(g_y) = the bias gradient of the dense layer in front of this block."""
    N, H = g_out.shape
    g_y = torch.empty_like(g_out) if need_g_y else None
    g_resid = torch.empty_like(g_out) if need_g_resid else None
    l = lib()
    ws = _workspace(g_out.device, l.kvq_ln_bwd_workspace_bytes(N, H))
    ref = g_gamma if g_gamma is not None else (g_beta if g_beta is not None else g_bias_prev)
    pdt = io_dtype_of(ref) if ref is not None else 0
    check(l.kvq_dropout_residual_ln_bwd(g_out.data_ptr(), pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), N, H,
                                        float(p_drop), int(seed), int(site), io_dtype_of(g_out), _p(g_y), _p(g_resid), _p(g_gamma),
                                        _p(g_beta), _p(g_bias_prev), pdt, int(accumulate), ws.data_ptr(), ws.numel(), stream_ptr()),
          "kvq_dropout_residual_ln_bwd")
    return g_y, g_resid


def colsum(x, out, scale=1.0, accumulate=False, cols=None):
    """out[c] (= | +=) scale * sum_n x[n, c] for the first `cols` columns of a row-major 2-D tensor (row stride = x.stride(0))."""
    N = x.shape[0]
    C = x.shape[1] if cols is None else cols
    l = lib()
    ws = _workspace(x.device, l.kvq_colsum_workspace_bytes(N, C))
    check(l.kvq_colsum(x.data_ptr(), N, C, x.stride(0), io_dtype_of(x), out.data_ptr(), io_dtype_of(out), float(scale),
                       int(accumulate), ws.data_ptr(), ws.numel(), stream_ptr()), "kvq_colsum")
    return out


def ln_bwd_partial(g_out, pre, mean, rstd, gamma, p_drop=0.0, seed=0, site=0, need_g_y=True, need_g_resid=True, want_dbias=False):
    """LayerNorm backward without the final parameter-gradient sums: returns (g_y, g_resid, part) with part [rows, 3H] f32 =
    per-workgroup partials [dbias_prev | dgamma | dbeta]; hand column ranges of `part` to reduce_batch()."""
    N, H = g_out.shape
    g_y = torch.empty_like(g_out) if need_g_y else None
    g_resid = torch.empty_like(g_out) if need_g_resid else None
    l = lib()
    part = torch.empty((l.kvq_ln_bwd_partial_rows(N), 3 * H), dtype=torch.float32, device=g_out.device)
    check(l.kvq_dropout_residual_ln_bwd_partial(g_out.data_ptr(), pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                                N, H, float(p_drop), int(seed), int(site), io_dtype_of(g_out), _p(g_y), _p(g_resid),
                                                int(want_dbias), part.data_ptr(), part.numel() * 4, stream_ptr()),
          "kvq_dropout_residual_ln_bwd_partial")
    return g_y, g_resid, part


def embed_ln_fwd(ids, word, pos, type_row, gamma, beta, eps, seq_len, p_drop=0.0, seed=0, site=0):
    """BertEmbeddings in one kernel: out = dropout(LayerNorm(word[ids] + pos[n % seq_len] + type_row)).
    Returns (out, pre, mean, rstd) like ln_fwd; ids flat int64 [N]."""
    require_gpu(ids, word, pos)
    N, H = ids.numel(), word.shape[1]
    if pos.shape[0] < seq_len or N % seq_len != 0:      # the kernel reads pos[(n % seq_len)]: rows that must exist
        raise KvqError(f"kvq.nnops.embed_ln_fwd: {N} ids in sentences of {seq_len} tokens need {seq_len} position rows "
                       f"(the table has {pos.shape[0]}) and a whole number of sentences")
    out = torch.empty((N, H), dtype=word.dtype, device=word.device)
    pre = torch.empty_like(out)
    mean = torch.empty(N, dtype=torch.float32, device=word.device)
    rstd = torch.empty(N, dtype=torch.float32, device=word.device)
    check(lib().kvq_embed_ln_fwd(ids.data_ptr(), word.data_ptr(), pos.data_ptr(), type_row.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                 N, int(seq_len), H, word.shape[0], float(eps), float(p_drop), int(seed), int(site), io_dtype_of(word),
                                 out.data_ptr(), pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(), stream_ptr()), "kvq_embed_ln_fwd")
    return out, pre, mean, rstd


def ln_dropout_bwd_partial(g_out, pre, mean, rstd, gamma, p_drop, seed, site):
    """Backward of dropout(LayerNorm(x)) (the embedding block): returns (g_x, part) with part [rows, 3H] f32 = [- | dgamma | dbeta]."""
    N, H = g_out.shape
    g_x = torch.empty_like(g_out)
    l = lib()
    part = torch.empty((l.kvq_ln_bwd_partial_rows(N), 3 * H), dtype=torch.float32, device=g_out.device)
    check(l.kvq_ln_dropout_bwd_partial(g_out.data_ptr(), pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), N, H,
                                       float(p_drop), int(seed), int(site), io_dtype_of(g_out), g_x.data_ptr(), part.data_ptr(),
                                       part.numel() * 4, stream_ptr()), "kvq_ln_dropout_bwd_partial")
    return g_x, part


def colsum_partial(x, cols=None):
    """Per-workgroup partial column sums of a row-major 2-D tensor: part [rows, C] f32 (finish with reduce_batch())."""
    N = x.shape[0]
    C = x.shape[1] if cols is None else cols
    l = lib()
    part = torch.empty((l.kvq_colsum_partial_rows(N), C), dtype=torch.float32, device=x.device)
    check(l.kvq_colsum_partial(x.data_ptr(), N, C, x.stride(0), io_dtype_of(x), part.data_ptr(), part.numel() * 4, stream_ptr()),
          "kvq_colsum_partial")
    return part


def reduce_item(src, dst, count, cols, ld, scale=1.0, accumulate=False, src_offset=0):
    """One entry of reduce_batch(): dst[c] (= | +=) scale * sum_{p<count} src.flat[src_offset + p*ld + c], c < cols."""
    from ._ffi import ReduceItem
    return ReduceItem(src.data_ptr() + src_offset * src.element_size(), dst.data_ptr(), int(count), int(cols), int(ld), float(scale),
                      io_dtype_of(src), io_dtype_of(dst), int(accumulate))


def reduce_batch(items):
    """Run a list of reduce_item()s, KVQ_REDUCE_MAX_ITEMS per launch.  The caller keeps the source tensors alive until here."""
    from ._ffi import ReduceItem, KVQ_REDUCE_MAX_ITEMS
    l = lib()
    st = stream_ptr()
    for i in range(0, len(items), KVQ_REDUCE_MAX_ITEMS):
        chunk = items[i:i + KVQ_REDUCE_MAX_ITEMS]
        arr = (ReduceItem * len(chunk))(*chunk)
        check(l.kvq_reduce_batch(arr, len(chunk), st), "kvq_reduce_batch")


def sum_slabs(part, out):
    """out = part.sum(0) for a [S, ...] stack of split-K partial results (f32 accumulate)."""
    S = part.shape[0]
    check(lib().kvq_sum_slabs(part.data_ptr(), S, out.numel(), io_dtype_of(part), out.data_ptr(), stream_ptr()), "kvq_sum_slabs")
    return out


def gelu_fwd(h):
    a = torch.empty_like(h)
    check(lib().kvq_gelu_fwd(h.data_ptr(), a.data_ptr(), h.numel(), io_dtype_of(h), stream_ptr()), "kvq_gelu_fwd")
    return a


def gelu_bwd(h, g_a, out=None):
    g_h = torch.empty_like(h) if out is None else out
    check(lib().kvq_gelu_bwd(h.data_ptr(), g_a.data_ptr(), g_h.data_ptr(), h.numel(), io_dtype_of(h), stream_ptr()), "kvq_gelu_bwd")
    return g_h


def gelu_bwd_bias(h, g_a, out=None):
    """(g_h, part): gelu_bwd plus part [rows, C] f32 partial column sums of g_h (bias gradient partials for reduce_batch())."""
    N, Cc = h.shape
    g_h = torch.empty_like(h) if out is None else out
    l = lib()
    part = torch.empty((l.kvq_gelu_bwd_partial_rows(N), Cc), dtype=torch.float32, device=h.device)
    check(l.kvq_gelu_bwd_bias(h.data_ptr(), g_a.data_ptr(), g_h.data_ptr(), N, Cc, io_dtype_of(h), part.data_ptr(), part.numel() * 4,
                              stream_ptr()), "kvq_gelu_bwd_bias")
    return g_h, part


def attn_fwd(q, k, v, mask, B, nh, Sq, Sk, causal, p_drop=0.0, seed=0, site=0, out=None):
    """q [B*Sq, >=nh*64] (any row stride), k, v [B*Sk, ...]; returns (ctx [B*Sq, nh*64], lse [B, nh, Sq])."""
    require_gpu(q, k, v)
    dh = 64
    ctx = torch.empty((B * Sq, nh * dh), dtype=q.dtype, device=q.device) if out is None else out
    lse = torch.empty((B, nh, Sq), dtype=torch.float32, device=q.device)
    check(lib().kvq_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(mask), B, nh, Sq, Sk, dh, q.stride(0), k.stride(0),
                             v.stride(0), ctx.stride(0), int(causal), 1.0 / 8.0, float(p_drop), int(seed), int(site),
                             io_dtype_of(q), ctx.data_ptr(), lse.data_ptr(), stream_ptr()), "kvq_attn_fwd")
    return ctx, lse


def attn_fwd_fp8_ok(Sq, Sk):
    return bool(lib().kvq_attn_fwd_fp8_ok(int(Sq), int(Sk)))


def attn_fwd_fp8(q, k, v, mask, B, nh, Sq, Sk, causal, p_drop, seed, site, state):
    """attn_fwd plus the fp8 copy of the context for the fp8 BertSelfOutput.dense: (ctx, lse, ctx8) -- see ln_fwd_fp8."""
    require_gpu(q, k, v)
    dh = 64
    assert q.dtype == torch.bfloat16
    ctx = torch.empty((B * Sq, nh * dh), dtype=q.dtype, device=q.device)
    ctx8 = torch.empty((B * Sq, nh * dh), dtype=torch.uint8, device=q.device)
    lse = torch.empty((B, nh, Sq), dtype=torch.float32, device=q.device)
    check(lib().kvq_attn_fwd_fp8(q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(mask), B, nh, Sq, Sk, dh, q.stride(0), k.stride(0),
                                 v.stride(0), ctx.stride(0), int(causal), 1.0 / 8.0, float(p_drop), int(seed), int(site),
                                 ctx.data_ptr(), lse.data_ptr(), ctx8.data_ptr(), ctx8.stride(0), state.data_ptr(), stream_ptr()),
          "kvq_attn_fwd_fp8")
    return ctx, lse, ctx8


def attn_bwd(q, k, v, mask, g_ctx, B, nh, Sq, Sk, causal, p_drop, seed, site, g_q, g_k, g_v, bias_part_q=None, bias_part_k=None,
             bias_part_v=None, ctx=None, lse=None):
    """Writes g_q / g_k / g_v (same layouts / row strides as q / k / v).  bias_part_* (f32 2-D views with B rows, row stride
    free, k and v sharing one) receive the per-batch column sums of g_q / g_k / g_v = partial rows of the projection-bias gradients."""
    dh = 64
    assert g_q.stride(0) == q.stride(0) and g_k.stride(0) == k.stride(0) and g_v.stride(0) == v.stride(0)
    ldp_q = bias_part_q.stride(0) if bias_part_q is not None else 0
    kv_ref = bias_part_k if bias_part_k is not None else bias_part_v
    ldp_kv = kv_ref.stride(0) if kv_ref is not None else 0
    for t in (bias_part_q, bias_part_k, bias_part_v):
        assert t is None or (t.dtype == torch.float32 and t.shape[0] == B and t.shape[1] >= nh * dh and t.stride(1) == 1)
    assert bias_part_k is None or bias_part_v is None or bias_part_k.stride(0) == bias_part_v.stride(0)
    if ctx is not None and lse is not None:       # the forward's output and log-sum-exp: what the kernels above 32 tokens work from
        assert ctx.stride(0) == g_ctx.stride(0) and lse.is_contiguous() and lse.dtype == torch.float32
        check(lib().kvq_attn_bwd_saved(q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(mask), ctx.data_ptr(), lse.data_ptr(), g_ctx.data_ptr(),
                                       B, nh, Sq, Sk, dh, q.stride(0), k.stride(0), v.stride(0), g_ctx.stride(0), int(causal), 1.0 / 8.0,
                                       float(p_drop), int(seed), int(site), io_dtype_of(q), g_q.data_ptr(), g_k.data_ptr(), g_v.data_ptr(),
                                       _p(bias_part_q), _p(bias_part_k), _p(bias_part_v), int(ldp_q), int(ldp_kv), stream_ptr()),
              "kvq_attn_bwd_saved")
        return
    check(lib().kvq_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(mask), g_ctx.data_ptr(), B, nh, Sq, Sk, dh, q.stride(0),
                             k.stride(0), v.stride(0), g_ctx.stride(0), int(causal), 1.0 / 8.0, float(p_drop), int(seed), int(site),
                             io_dtype_of(q), g_q.data_ptr(), g_k.data_ptr(), g_v.data_ptr(), _p(bias_part_q), _p(bias_part_k),
                             _p(bias_part_v), int(ldp_q), int(ldp_kv), stream_ptr()), "kvq_attn_bwd")


def adam_step(p32, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, vmax=None, shadow=None, grad_scale=1.0):
    check(lib().kvq_adam_step(p32.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(vmax), _p(shadow), p32.numel(),
                              io_dtype_of(g), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                              float(grad_scale), stream_ptr()), "kvq_adam_step")


def adam_step_dev(p32, g, m, v, step_state, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, vmax=None, shadow=None, grad_scale=1.0):
    """adam_step with lr and the bias corrections read on the device from `step_state` (see step_state_advance)."""
    check(lib().kvq_adam_step_dev(p32.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(vmax), _p(shadow), p32.numel(),
                                  io_dtype_of(g), step_state.data_ptr(), float(beta1), float(beta2), float(eps), float(weight_decay),
                                  float(grad_scale), stream_ptr()), "kvq_adam_step_dev")


def new_step_state(device):
    """24 zeroed bytes: struct { uint64 step; float lr, bc1, bc2s, pad; } of include/kvq.h."""
    return torch.zeros(3, dtype=torch.int64, device=device)


def step_state_advance(step_state, lr0, gamma, milestones, beta1, beta2, phase="advance"):
    """phase "advance": step += 1 and the lr / bias corrections of that step; "prepare": only the latter (the step counter,
    which the dropout kernels add to their seed, stays); see step_state_commit()."""
    import ctypes
    ms = [int(x) for x in milestones]
    arr = (ctypes.c_int64 * max(len(ms), 1))(*ms)
    fn = {"advance": lib().kvq_step_state_advance, "prepare": lib().kvq_step_state_prepare}[phase]
    check(fn(step_state.data_ptr(), float(lr0), float(gamma), arr, len(ms), float(beta1), float(beta2), stream_ptr()),
          "kvq_step_state_" + phase)


def step_state_commit(step_state):
    check(lib().kvq_step_state_commit(step_state.data_ptr(), stream_ptr()), "kvq_step_state_commit")


def read_step_state(step_state):
    """(step, lr, bc1, bc2s) -- synchronises; for tests and logging."""
    raw = step_state.cpu()
    f = raw[1:].view(torch.float32)
    return int(raw[0]), float(f[0]), float(f[1]), float(f[2])


def set_seed_offset(step_state):
    """Dropout-bearing kernels launched from now on use seed + step_state.step (read on the device); None switches it off."""
    check(lib().kvq_set_seed_offset(step_state.data_ptr() if step_state is not None else None), "kvq_set_seed_offset")


def dropout(x, p_drop, seed, site, out=None):
    """x * keep / (1 - p) with the Philox mask of (seed, site); applying it to a gradient is the backward."""
    require_gpu(x)
    xc = x.contiguous()
    o = torch.empty_like(xc) if out is None else out
    check(lib().kvq_dropout(xc.data_ptr(), xc.numel(), float(p_drop), int(seed), int(site), io_dtype_of(xc), o.data_ptr(), stream_ptr()),
          "kvq_dropout")
    return o


def zero_ranges(tensors):
    """Zero up to four contiguous tensors (16-byte aligned, sizes multiples of 16 bytes) in one launch."""
    import ctypes
    ts = [t for t in tensors if t is not None]
    if not ts:
        return
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    sizes = (ctypes.c_int64 * len(ts))(*[t.numel() * t.element_size() for t in ts])
    check(lib().kvq_zero_ranges(ptrs, sizes, len(ts), stream_ptr()), "kvq_zero_ranges")


def embed_grad(g, perm, sorted_ids, gW, accumulate=False):
    """gW[id] (= | +=) sum of g[n] over the tokens with that id, in token order (deterministic, no atomics).
    perm / sorted_ids: stable sort of the flattened ids (sorted_ids, perm = torch.sort(ids.reshape(-1), stable=True))."""
    require_gpu(g, gW)
    N, H = g.shape
    l = lib()
    ws = _workspace(g.device, l.kvq_embed_grad_workspace_bytes(N, H), tag="embed_grad")
    check(l.kvq_embed_grad(g.data_ptr(), perm.data_ptr(), sorted_ids.data_ptr(), N, H, gW.shape[0], io_dtype_of(g), gW.data_ptr(),
                           io_dtype_of(gW), int(accumulate), ws.data_ptr(), ws.numel(), stream_ptr()), "kvq_embed_grad")
    return gW


# ---- the GEMM family of csrc/kvq_gemm2.hip ---------------------------------------------------------------------------------
_LAYOUTS = {"nt": 0, "nn": 1, "tn": 2}
# "128x192h" (round 5): four waves, two ring slots, TWO workgroups per CU.  Faster than the persistent / 256 x 192 kernels on hot
# operands ([8192, 2304] x 768: 34.5 against 37.0 us; [8192, 3072]: 39.6 against 41.6) and slower inside the step, where the operands
# are cold and one k-tile of prefetch is not enough (QKV 35.6 against 31.9 us): kept as an explicit choice, not in pick_tile's set
# (profiles/r05_gemm_ceiling.md)
TILES = {"128x192": 0, "128x256": 1, "256x192": 2, "256x256": 3, "64x128": 4, "128x192h": 5,
         "128x192p": 0x100, "128x256p": 0x101, "256x192p": 0x102, "256x256p": 0x103}      # p: KVQ_GEMM_PERSISTENT (layout nt only)
TILE_NAMES = {v: k for k, v in TILES.items()}
_TILE_DIMS = {0: (128, 192), 1: (128, 256), 2: (256, 192), 3: (256, 256), 4: (64, 128)}
def _gemm_dims(a, b, layout):
    if layout == "nt":
        (M, K), N = a.shape, b.shape[0]
    elif layout == "nn":
        (M, K), N = a.shape, b.shape[1]
    else:
        (K, M), N = a.shape, b.shape[1]
    return M, N, K


# Cost model of one launch, microseconds, fitted to tools/gemm2_probe_small.py on MI355X (profiles/r04_gemm_small.md: 72 products at
# 768 .. 6144 rows x 5 tiles; mean regret of the model's pick against the best measured tile 0.7 %, worst 17 %) and consistent with
# the choices measured at 8192 rows in rounds 2 - 3 (tools/gemm2_probe.py):
#   per 64-deep k-tile of ONE workgroup: _TK_ALONE when few CUs are busy (operands come from L2 unopposed), _TK_FULL when all 256
#   are (the L2 -> LDS fill is shared); linear in the busy fraction between.  _T_TILE: start-up + epilogue of a tile.
#   The 64 x 128 tile (4 waves, 72 KiB of LDS) runs two workgroups per CU, which together take _TK_PAIR per k-tile.
_TK_ALONE = {0: 0.459, 1: 0.557, 2: 0.891, 3: 1.035, 4: 0.296}
_TK_FULL = {0: 0.825, 1: 0.98, 2: 1.5, 3: 1.83, 4: 0.40}
_TK_PAIR = (0.45, 0.52)
_T_TILE = {0: 1.5, 1: 2.0, 2: 2.5, 3: 3.0, 4: 1.0}


def tile_cost_us(t, M, N, K, n_cu=256):
    """Modelled duration of an [M, N] x K product on tile t (see the table above)."""
    bm, bn = _TILE_DIMS[t]
    n = -(-M // bm) * -(-N // bn)
    kt = max(K // 64, 1)
    f = min(n, n_cu) / n_cu
    lerp = lambda a, b: a + (b - a) * f
    w = -(-n // n_cu)                                    # workgroups the busiest CU runs
    if t == 4:
        return kt * ((w // 2) * lerp(*_TK_PAIR) + (w % 2) * lerp(_TK_ALONE[t], _TK_FULL[t])) + w * _T_TILE[t]
    return w * (kt * lerp(_TK_ALONE[t], _TK_FULL[t]) + _T_TILE[t])


def pick_tile(M, N, K=768, n_cu=256, candidates=None):
    """Workgroup tile for an [M, N] x K product: the cheapest under tile_cost_us() -- the rule over (M, N, K) that replaced the
    engine's exact-shape tables.  candidates: restrict the choice (the fused-activation epilogues exist for two tiles)."""
    cand = list(_TILE_DIMS) if candidates is None else [TILES[c] if isinstance(c, str) else c for c in candidates]
    return min(cand, key=lambda t: (tile_cost_us(t, M, N, K, n_cu), -_TILE_DIMS[t][0] * _TILE_DIMS[t][1]))


def persistent_pays(t, M, N, K, layout, accumulate=False, n_cu=256):
    """The persistent tile loop (KVQ_GEMM_PERSISTENT: NT only, whole tiles, no accumulate) instead of one tile per workgroup:
    when a CU owns 3 .. 12 tiles.  Measured at 8192 rows (tools/gemm2_probe_persist.py, profiles/NOTES_r01-r03_design_and_experiments.md section 2.3): QKV (3 tiles per CU)
    and the all-layer cross-K/V projection (9) gain 1 - 4 %, two tiles per CU (FFN1) lose, the LM head (15, column bands) loses 6 %."""
    bm, bn = _TILE_DIMS[t]
    if layout != "nt" or accumulate or t == 4 or M < bm or N < bn or K // 64 < 3:
        return False
    per_cu = -(-(-(-M // bm) * -(-N // bn)) // n_cu)
    return 3 <= per_cu <= 12


def gemm_mfma_ok(a, b, out, layout, bias=None):
    """True when kvq_gemm_bf16 (LDS-DMA + MFMA, csrc/kvq_gemm2.hip) takes this product: K % 64 == 0; M, N and the leading dimensions
    in multiples of 8; 16-byte aligned operands (bias: 8); unit column stride."""
    M, N, K = _gemm_dims(a, b, layout)
    ts = [a, b] + ([out] if out is not None else [])
    return K % 64 == 0 and M % 8 == 0 and N % 8 == 0 and M >= 8 and N >= 8 \
        and all(t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0 for t in ts) \
        and (bias is None or (bias.data_ptr() % 8 == 0 and bias.stride(0) == 1))


def gemm_problem(a, b, out, layout, bias=None, accumulate=False):
    from ._ffi import GemmProblem
    M, N, K = _gemm_dims(a, b, layout)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
    assert a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1 and tuple(out.shape) == (M, N)
    return GemmProblem(a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(bias), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                       int(accumulate))


def pad_rows(t, rows_padded):
    """[rows_padded, cols] copy of a row-major 2-D tensor (unit column stride) with zero rows behind its own -- one kernel."""
    rows, cols = t.shape
    es = t.element_size()
    out = torch.empty((rows_padded, cols), dtype=t.dtype, device=t.device)
    check(lib().kvq_pad_rows(t.data_ptr(), rows, cols * es, t.stride(0) * es, out.data_ptr(), rows_padded, stream_ptr()), "kvq_pad_rows")
    return out


def _pad_ok(t):
    return t.dim() == 2 and t.stride(1) == 1 and (t.shape[1] * t.element_size()) % 16 == 0 and (t.stride(0) * t.element_size()) % 16 == 0 \
        and t.data_ptr() % 16 == 0


def tn_operands_k64(a, b):
    """The operands of a "tn" product (both [K, .], K = tokens) with K padded to the MFMA kernel's 64-deep k-tile by zero rows
    (which add nothing to a^T . b), or None when K % 64 == 0 already / the rows cannot be copied in 16-byte pieces.  A batch of
    100 sentences x 12 tokens (the last batch of an epoch, any odd batch size) would otherwise put every weight gradient of the
    step -- the [30528, 768] LM head included -- on the any-shape kernel."""
    K = a.shape[0]
    if K % 64 == 0 or not (_pad_ok(a) and _pad_ok(b)):
        return None
    Kp = (K + 63) // 64 * 64
    return pad_rows(a, Kp), pad_rows(b, Kp)


_ANY_WARNED = set()
GEMM_ROUTES = {"mfma": 0, "any": 0, "tn_padded": 0, "row_split": 0}      # launches of gemm() by route (tests read it)


def _warn_any(M, N, K, layout):
    """Once per shape: a product of more than 64 MFLOP that no MFMA route takes (the any-shape kernel is a scalar-FMA kernel)."""
    if 2.0 * M * N * K > 64e6 and (M, N, K, layout) not in _ANY_WARNED:
        _ANY_WARNED.add((M, N, K, layout))
        import warnings
        warnings.warn(f"kvq.nnops.gemm: {layout} product M={M} N={N} K={K} ({2e-6 * M * N * K:.0f} MFLOP) runs on the any-shape kernel: "
                      f"the MFMA kernel needs K % 64 == 0, M / N / leading dimensions % 8 == 0 and 16-byte aligned operands", stacklevel=3)


def _gemm_any(a, b, layout, bias, out, accumulate, warn=True):
    M, N, K = _gemm_dims(a, b, layout)
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    if out.stride(1) != 1 or (bias is not None and bias.stride(0) != 1):
        raise KvqError("kvq.nnops.gemm: output rows / bias must be contiguous")
    if warn:
        _warn_any(M, N, K, layout)
    GEMM_ROUTES["any"] += 1
    check(lib().kvq_gemm_any_bf16(a.data_ptr(), b.data_ptr(), _p(bias), out.data_ptr(), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                                  _LAYOUTS[layout], int(accumulate), stream_ptr()), "kvq_gemm_any_bf16")
    return out


def gemm(a, b, layout="nt", bias=None, out=None, accumulate=False, tile=None):
    """out[M,N] (= | +=) op(a) @ op(b) (+ bias), bf16 with f32 accumulation.  layout "nt": a[M,K], b[N,K];  "nn": a[M,K], b[K,N];
    "tn": a[K,M], b[K,N].  The MFMA GEMM of csrc/kvq_gemm2.hip whenever the product meets its requirements (gemm_mfma_ok), tile
    from pick_tile() unless given.  Products that miss them only through their TOKEN count still reach it: a "tn" contraction
    over tokens is zero-padded to a multiple of 64 (tn_operands_k64), an "nt" / "nn" product over a token count that is not a
    multiple of 8 runs its first M - M % 8 rows there and the last rows on the any-shape kernel of csrc/kvq_gemm_any.hip, which
    also takes everything else (with a warning above 64 MFLOP).  Never a vendor library."""
    require_gpu(a, b)
    M, N, K = _gemm_dims(a, b, layout)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or (out is not None and out.dtype != torch.bfloat16):
        raise KvqError(f"kvq.nnops.gemm: bf16 operands only (got {a.dtype}, {b.dtype})")
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    if not gemm_mfma_ok(a, b, out, layout, bias):
        if layout == "tn" and K % 64 != 0 and K > 64:
            padded = tn_operands_k64(a, b)
            if padded is not None and gemm_mfma_ok(padded[0], padded[1], out, layout, bias):
                GEMM_ROUTES["tn_padded"] += 1
                return gemm(padded[0], padded[1], layout, bias=bias, out=out, accumulate=accumulate, tile=tile)
        if layout != "tn" and M % 8 != 0 and M > 8 and gemm_mfma_ok(a[:M - M % 8], b, out[:M - M % 8], layout, bias):
            M8 = M - M % 8
            GEMM_ROUTES["row_split"] += 1
            gemm(a[:M8], b, layout, bias=bias, out=out[:M8], accumulate=accumulate, tile=tile)
            _gemm_any(a[M8:], b, layout, bias, out[M8:], accumulate, warn=False)                 # (at most 7 rows)
            return out
        return _gemm_any(a, b, layout, bias, out, accumulate)
    if tile is None:
        t = pick_tile(M, N, K)
        if persistent_pays(t, M, N, K, layout, accumulate):
            t |= 0x100
    else:
        t = TILES[tile] if isinstance(tile, str) else tile
    GEMM_ROUTES["mfma"] += 1
    check(lib().kvq_gemm_bf16(a.data_ptr(), b.data_ptr(), _p(bias), out.data_ptr(), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                              _LAYOUTS[layout], t, int(accumulate), stream_ptr()), "kvq_gemm_bf16")
    return out


def gemm_grouped(problems, layout, tile):
    """One launch over a list of gemm_problem()s of one layout (at most 16)."""
    from ._ffi import GemmProblem
    arr = (GemmProblem * len(problems))(*problems)
    t = TILES[tile] if isinstance(tile, str) else tile
    check(lib().kvq_gemm_grouped_bf16(arr, len(problems), _LAYOUTS[layout], t, stream_ptr()), "kvq_gemm_grouped_bf16")


def gemm_gelu(x, w, bias, tile="256x192"):
    """(h, gelu(h)) with h = x @ w.T + bias in one kernel (BertIntermediate, modeling_bert.py:325-337)."""
    M, K = x.shape
    N = w.shape[0]
    h = torch.empty((M, N), dtype=x.dtype, device=x.device)
    a = torch.empty_like(h)
    check(lib().kvq_gemm_bf16_gelu(x.data_ptr(), w.data_ptr(), _p(bias), h.data_ptr(), a.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                                   h.stride(0), TILES[tile], stream_ptr()), "kvq_gemm_bf16_gelu")
    return h, a


DROPRES_TILES = ("128x192", "128x256", "64x128")


def gemm_dropres(x, w, bias, resid, p_drop, seed, site, tile=None):
    """pre = dropout(x @ w.T + bias, p_drop) + resid in one kernel: the LayerNorm input of BertSelfOutput / BertOutput
    (modeling_bert.py:282-296, 339-352), bit for bit what ln_fwd(x @ w.T + bias, resid, ..., p_drop, seed, site) stores as `pre`
    (same roundings, same Philox masks) -- so the block's forward is this + ln_fwd(pre, None, ..., p_drop=0) and its backward is
    unchanged.  The tile comes from pick_tile() over the tiles this epilogue is built for."""
    M, K = x.shape
    N = w.shape[0]
    assert tuple(resid.shape) == (M, N) and resid.dtype == torch.bfloat16 and resid.stride(1) == 1
    pre = torch.empty((M, N), dtype=x.dtype, device=x.device)
    assert resid.stride(0) == pre.stride(0)
    t = pick_tile(M, N, K, candidates=DROPRES_TILES) if tile is None else (TILES[tile] if isinstance(tile, str) else tile)
    GEMM_ROUTES["mfma"] += 1
    check(lib().kvq_gemm_bf16_dropres(x.data_ptr(), w.data_ptr(), _p(bias), resid.data_ptr(), pre.data_ptr(), M, N, K, x.stride(0),
                                      w.stride(0), pre.stride(0), t, float(p_drop), int(seed), int(site), stream_ptr()),
          "kvq_gemm_bf16_dropres")
    return pre


def gemm_dgelu(gy, w, h, tile="256x192"):
    """(g_h, part): g_h = (gy @ w) * gelu'(h) and part [rows, N] f32 partial column sums of g_h (bias-gradient partials for
    reduce_batch()) in one kernel -- the input gradient of BertOutput.dense pushed through BertIntermediate's activation."""
    M, K = gy.shape
    N = w.shape[1]
    assert tuple(h.shape) == (M, N) and h.is_contiguous()
    out = torch.empty((M, N), dtype=gy.dtype, device=gy.device)
    l = lib()
    t = TILES[tile]
    part = torch.empty((l.kvq_gemm_dgelu_partial_rows(M, t), N), dtype=torch.float32, device=gy.device)
    check(l.kvq_gemm_bf16_dgelu(gy.data_ptr(), w.data_ptr(), h.data_ptr(), out.data_ptr(), part.data_ptr(), part.numel() * 4, M, N, K,
                                gy.stride(0), w.stride(0), out.stride(0), t, stream_ptr()), "kvq_gemm_bf16_dgelu")
    return out, part


# ---- fp8 forward GEMMs (csrc/kvq_fp8.hip, kvq_gemm_fp8_nt) -----------------------------------------------------------------------
def gemm_ce(x, w, bias, V):
    """(logits, stats): logits = x @ w.T + bias [M, N] and the per-(row, 256-column tile) softmax statistics of the columns < V
    from the same kernel (LM head + forward half of the reconstruction loss; see ce_forward_stats)."""
    M, K = x.shape
    N = w.shape[0]
    logits = torch.empty((M, N), dtype=x.dtype, device=x.device)
    stats = torch.empty((M, (N + 255) // 256, 4), dtype=torch.float32, device=x.device)
    check(lib().kvq_gemm_bf16_ce(x.data_ptr(), w.data_ptr(), _p(bias), logits.data_ptr(), M, N, K, x.stride(0), w.stride(0),
                                 logits.stride(0), int(V), stats.data_ptr(), stats.numel() * 4, stream_ptr()), "kvq_gemm_bf16_ce")
    return logits, stats


def ce_forward_stats(logits, target, stats, row_loss, row_lse, pred, loss, acc):
    """kvq_ce_forward's outputs from gemm_ce()'s statistics (reads one logit per row)."""
    N = logits.shape[0]
    check(lib().kvq_ce_forward_stats(logits.data_ptr(), target.data_ptr(), N, logits.stride(0), io_dtype_of(logits), stats.data_ptr(),
                                     stats.shape[1], row_loss.data_ptr(), row_lse.data_ptr(), pred.data_ptr(), _p(loss), _p(acc),
                                     stream_ptr()), "kvq_ce_forward_stats")


def fp8_quantize(x, out=None):
    """bf16 [rows, cols] (unit column stride) -> (fp8 bytes [rows, cols] as uint8, scale [1] f32 on the device)."""
    require_gpu(x)
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    st = torch.empty(2, dtype=torch.float32, device=x.device)             # [amax, scale]
    check(lib().kvq_fp8_quantize(x.data_ptr(), rows, cols, x.stride(0), out.data_ptr(), st[0:].data_ptr(), st[1:].data_ptr(), stream_ptr()),
          "kvq_fp8_quantize")
    return out, st[1:]


def gemm_fp8_nt(a8, b8, scale_a, scale_b, bias=None, out=None):
    """out[M,N] bf16 = (a8[M,K] @ b8[N,K].T) / (scale_a * scale_b) + bias on the fp8 matrix cores."""
    M, K = a8.shape
    N = b8.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a8.device)
    check(lib().kvq_gemm_fp8_nt(a8.data_ptr(), b8.data_ptr(), scale_a.data_ptr(), scale_b.data_ptr(), _p(bias), out.data_ptr(), M, N, K,
                                a8.stride(0), b8.stride(0), out.stride(0), stream_ptr()), "kvq_gemm_fp8_nt")
    return out


def gemm_fp8_nt_gelu(a8, b8, scale_a, scale_b, bias, state_out=None):
    """(h, gelu(h)[, fp8 copy of gelu(h)]) with h = (a8 @ b8.T) / (scale_a * scale_b) + bias on the fp8 matrix cores: BertIntermediate
    with the GELU epilogue of gemm_gelu; state_out = the delayed-scaling record of the fp8 GEMM that reads gelu(h) next."""
    M, K = a8.shape
    N = b8.shape[0]
    h = torch.empty((M, N), dtype=torch.bfloat16, device=a8.device)
    a = torch.empty_like(h)
    a_8 = torch.empty((M, N), dtype=torch.uint8, device=a8.device) if state_out is not None else None
    check(lib().kvq_gemm_fp8_nt_gelu(a8.data_ptr(), b8.data_ptr(), scale_a.data_ptr(), scale_b.data_ptr(), _p(bias), h.data_ptr(),
                                     a.data_ptr(), _p(a_8), N if a_8 is not None else 0, _p(state_out), M, N, K, a8.stride(0), b8.stride(0),
                                     h.stride(0), stream_ptr()), "kvq_gemm_fp8_nt_gelu")
    return (h, a, a_8) if a_8 is not None else (h, a)


# ---- clock probe (bench.py) -----------------------------------------------------------------------------------------------------
def clock_probe():
    """Launch kvq_clock_probe on the current stream; returns its [rows, 4] int64 result buffer (read it after a synchronisation)."""
    l = lib()
    out = torch.zeros((l.kvq_clock_probe_rows(), 4), dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
    check(l.kvq_clock_probe(out.data_ptr(), out.numel() * 8, stream_ptr()), "kvq_clock_probe")
    return out


def clock_mhz(p0, p1):
    """Shader clock held between two clock_probe() results of one stream: per XCD (the cycle counter is the XCD's own) the first
    stamp of each probe, d(s_memtime) / d(s_memrealtime) x 100 MHz; returns (median over the XCDs seen by both, {xcd: MHz})."""
    a, b = p0.cpu().tolist(), p1.cpu().tolist()

    def first(rows):
        d = {}
        for xcc, t, r, _ in rows:
            if xcc not in d or r < d[xcc][1]:
                d[xcc] = (t, r)
        return d
    fa, fb = first(a), first(b)
    per = {}
    for x in sorted(set(fa) & set(fb)):
        dt, dr = fb[x][0] - fa[x][0], fb[x][1] - fa[x][1]
        if dr > 0 and dt > 0:
            per[int(x)] = dt / dr * 100.0
    if not per:
        return None, {}
    v = sorted(per.values())
    return v[len(v) // 2], per
