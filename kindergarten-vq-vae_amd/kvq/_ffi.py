"""ctypes binding of libkvq.so -- the exact symbol list of include/kvq.h.

The library is loaded lazily on first use and the load FAILS LOUDLY: there is no Python fallback for any
of these entry points (a silent fallback would void every parity claim made for the HIP path).
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# KVQ_LIB_PATH: another build of the same library (A/B runs of two builds on one GPU box, the diagnostic build of tools/); the load
# still fails loudly when the file is missing
LIB_PATH = os.environ.get("KVQ_LIB_PATH") or os.path.join(_PKG, "lib", "libkvq.so")

KVQ_F32, KVQ_BF16 = 0, 1

_vp, _i64, _int, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t

KVQ_REDUCE_MAX_ITEMS = 32


class ReduceItem(C.Structure):
    """struct kvq_reduce_item of include/kvq.h"""
    _fields_ = [("src", _vp), ("dst", _vp), ("count", _i64), ("cols", _i64), ("ld", _i64), ("scale", _f32),
                ("src_dtype", C.c_int32), ("dst_dtype", C.c_int32), ("accumulate", C.c_int32)]


class GemmProblem(C.Structure):
    """struct kvq_gemm_problem of include/kvq.h"""
    _fields_ = [("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("M", _int), ("N", _int), ("K", _int),
                ("lda", _int), ("ldb", _int), ("ldc", _int), ("accumulate", _int)]


GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
GEMM_TILE_128x192, GEMM_TILE_128x256, GEMM_TILE_256x192, GEMM_TILE_256x256, GEMM_TILE_64x128 = 0, 1, 2, 3, 4

# name -> (restype, argtypes); mirrors include/kvq.h line by line
SIGNATURES = {
    "kvq_version": (_int, []),
    "kvq_last_error": (C.c_char_p, []),
    "kvq_device_info": (_int, [C.POINTER(_int), C.c_char_p, _sz]),
    "kvq_graph_census": (_int, [_vp, C.POINTER(_i64)]),
    "kvq_pad_rows": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "kvq_prof_enable": (_int, [_int]),
    "kvq_prof_read": (_int, [C.POINTER(C.c_float), _int]),
    "kvq_clock_probe_rows": (_int, []),
    "kvq_clock_probe": (_int, [_vp, _sz, _vp]),
    "kvq_vq_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "kvq_vq_uses_mfma": (_int, [_i64, _int, _int]),
    "kvq_vq_forward": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kvq_vq_set_variant": (_int, [_int]),
    "kvq_vq_packed_bytes": (_sz, [_int, _int, _int]),
    "kvq_vq_pack_codebook": (_int, [_vp, _int, _int, _int, _vp, _vp]),
    "kvq_vq_forward_packed": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kvq_vq_backward": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _vp, _vp, _vp, _sz, _vp]),
    "kvq_vq_one_hot": (_int, [_vp, _i64, _int, _vp, _vp]),
    "kvq_vq_debug_distances": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp]),
    "kvq_vq_ema_update": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _f32, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kvq_kmeans_update": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "kvq_gumbel_forward": (_int, [_vp, _vp, _i64, _int, _f32, _int, C.c_uint64, C.c_uint32, _int, _vp, _vp, _vp, _vp, _vp]),
    "kvq_gumbel_backward": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _f32, _f32, _int, _vp, _vp]),
    "kvq_ce_forward": (_int, [_vp, _vp, _i64, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kvq_seq_acc": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "kvq_ce_backward": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _int, _vp, _vp]),
    "kvq_ce_bwd_partial_rows": (_i64, [_i64]),
    "kvq_ce_backward_bias": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _int, _vp, _vp, _sz, _vp]),
    "kvq_dropout_residual_ln_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _f32, _f32, C.c_uint64, C.c_uint32, _int, _vp, _vp, _vp, _vp, _vp]),
    "kvq_ln_bwd_workspace_bytes": (_sz, [_i64, _int]),
    "kvq_dropout_residual_ln_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _f32, C.c_uint64, C.c_uint32, _int, _vp, _vp, _vp, _vp, _vp,
                                           _int, _int, _vp, _sz, _vp]),
    "kvq_colsum_workspace_bytes": (_sz, [_i64, _i64]),
    "kvq_colsum": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _int, _f32, _int, _vp, _sz, _vp]),
    "kvq_sum_slabs": (_int, [_vp, _int, _i64, _int, _vp, _vp]),
    "kvq_reduce_batch": (_int, [C.POINTER(ReduceItem), _int, _vp]),
    "kvq_ln_bwd_partial_rows": (_i64, [_i64]),
    "kvq_dropout_residual_ln_bwd_partial": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _f32, C.c_uint64, C.c_uint32, _int, _vp, _vp,
                                                   _int, _vp, _sz, _vp]),
    "kvq_embed_ln_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _i64, _f32, _f32, C.c_uint64, C.c_uint32, _int, _vp, _vp,
                                _vp, _vp, _vp]),
    "kvq_ln_dropout_bwd_partial": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _f32, C.c_uint64, C.c_uint32, _int, _vp, _vp, _sz, _vp]),
    "kvq_colsum_partial_rows": (_i64, [_i64]),
    "kvq_colsum_partial": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _sz, _vp]),
    "kvq_gelu_fwd": (_int, [_vp, _vp, _i64, _int, _vp]),
    "kvq_gelu_bwd": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "kvq_gelu_bwd_partial_rows": (_i64, [_i64]),
    "kvq_gelu_bwd_bias": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _sz, _vp]),
    "kvq_attn_fwd": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _int, _int, _f32, _f32,
                            C.c_uint64, C.c_uint32, _int, _vp, _vp, _vp]),
    "kvq_attn_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _int, _int, _f32, _f32,
                            C.c_uint64, C.c_uint32, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp]),
    "kvq_attn_bwd_saved": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _int, _int, _f32, _f32,
                                  C.c_uint64, C.c_uint32, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp]),
    "kvq_gemm_bf16": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "kvq_gemm_any_bf16": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "kvq_gemm_grouped_bf16": (_int, [C.POINTER(GemmProblem), _int, _int, _int, _vp]),
    "kvq_fp8_quantize": (_int, [_vp, _i64, _int, _i64, _vp, _vp, _vp, _vp]),
    "kvq_fp8_state_floats": (_int, []),
    "kvq_fp8_quantize_delayed": (_int, [_vp, _i64, _int, _i64, _vp, _vp, _vp]),
    "kvq_fp8_update_scales": (_int, [_vp, _int, _f32, _vp]),
    "kvq_fp8_quantize_segments": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp]),
    "kvq_adam_step_dev_fp8": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp]),
    "kvq_fp8_quantize_segments_periodic": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _int, _vp]),
    "kvq_gemm_fp8_nt": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "kvq_dropout_residual_ln_fwd_fp8": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _f32, _f32, C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kvq_attn_fwd_fp8_ok": (_int, [_int, _int]),
    "kvq_attn_fwd_fp8": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _int, _int, _f32, _f32,
                                C.c_uint64, C.c_uint32, _vp, _vp, _vp, _int, _vp, _vp]),
    "kvq_gemm_fp8_nt_gelu": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "kvq_gemm_bf16_dropres": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _f32, C.c_uint64, C.c_uint32, _vp]),
    "kvq_gemm_bf16_gelu": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "kvq_gemm_dgelu_partial_rows": (_i64, [_i64, _int]),
    "kvq_gemm_bf16_dgelu": (_int, [_vp, _vp, _vp, _vp, _vp, _sz, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "kvq_attn_set_variant": (_int, [_int]),
    "kvq_adam_step": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _vp]),
    "kvq_ce_forward_stats": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kvq_gemm_ce_stats_bytes": (_sz, [_int, _int]),
    "kvq_gemm_bf16_ce": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp, _sz, _vp]),
    "kvq_step_state_advance": (_int, [_vp, _f32, _f32, C.POINTER(_i64), _int, _f32, _f32, _vp]),
    "kvq_step_state_prepare": (_int, [_vp, _f32, _f32, C.POINTER(_i64), _int, _f32, _f32, _vp]),
    "kvq_step_state_commit": (_int, [_vp, _vp]),
    "kvq_adam_step_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp, _f32, _f32, _f32, _f32, _f32, _vp]),
    "kvq_set_seed_offset": (_int, [_vp]),
    "kvq_embed_grad_workspace_bytes": (_sz, [_i64, _int]),
    "kvq_embed_grad": (_int, [_vp, _vp, _vp, _i64, _int, _i64, _int, _vp, _int, _int, _vp, _sz, _vp]),
    "kvq_dropout": (_int, [_vp, _i64, _f32, C.c_uint64, C.c_uint32, _int, _vp, _vp]),
    "kvq_zero_ranges": (_int, [C.POINTER(_vp), C.POINTER(_i64), _int, _vp]),
    "kvq_code_census": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _vp]),
}

_lib = None


class KvqError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises KvqError when libkvq.so has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KvqError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` or kindergarten-vq-vae_amd/build.sh. "
                f"There is no CPU/eager fallback for the kvq ops.")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 not found
            raise KvqError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:
                raise KvqError(f"{LIB_PATH} does not export {name} (stale build?)") from e
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("KVQ_VQ_FUSED") in ("0", "1"):       # A/B runs of the quantiser forward (tools/, bench.py): main thread only
            l.kvq_vq_set_variant(int(os.environ["KVQ_VQ_FUSED"]))
        _lib = l
    return _lib if _fam is None else _FamilyProxy(_lib)


# ---- per-family kernel time (bench.py): event pairs around every entry point of a few EAGER steps -------------------------------
_FAMILIES = (("kvq_gemm_", "gemm"), ("kvq_attn_", "attention"), ("kvq_dropout_residual_ln", "layernorm"), ("kvq_ln_", "layernorm"),
             ("kvq_embed_ln_fwd", "layernorm"), ("kvq_ce_", "loss"), ("kvq_seq_acc", "loss"), ("kvq_adam_", "adam"),
             ("kvq_step_state_", "adam"), ("kvq_vq_", "vq"), ("kvq_gumbel_", "vq"), ("kvq_fp8_", "fp8_quantize"), ("kvq_reduce_batch", "reduce"),
             ("kvq_colsum", "reduce"), ("kvq_sum_slabs", "reduce"), ("kvq_gelu_", "gelu"), ("kvq_embed_grad", "embedding_grad"),
             ("kvq_zero_ranges", "embedding_grad"))
_fam = None          # None = off; else dict(events=[(family, e0, e1)], cal=[(e0, e1)])


class _FamilyProxy:
    """Stands in for the CDLL while family_profile_begin() is active: every entry point that launches on a stream (last argument:
    the stream) runs between two events of the current torch stream, filed under its kernel family."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        sig = SIGNATURES.get(name)
        if _fam is None or sig is None or not sig[1] or sig[1][-1] is not _vp or name.endswith(("_bytes", "_rows")) \
                or name in ("kvq_set_seed_offset", "kvq_clock_probe"):
            return fn
        family = next((f for pre, f in _FAMILIES if name.startswith(pre)), "other")

        def call(*args):
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            _fam["events"].append((family, e0, e1))
            return rc
        return call


def family_profile_begin():
    """From now on lib() hands out the timing proxy.  Run EAGER steps (no hipGraph replay: captured launches cannot carry events)."""
    global _fam
    import torch
    lib()
    _fam = dict(events=[], cal=[])
    for _ in range(32):                       # what an EMPTY pair measures: subtracted per launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        _fam["cal"].append((e0, e1))


def family_profile_end():
    """Stop; returns ({family: ms summed over the launches seen}, {family: launches}, ms of an empty event pair).  Synchronises."""
    global _fam
    import torch
    torch.cuda.synchronize()
    rec, _fam = _fam, None
    cal = sorted(e0.elapsed_time(e1) for e0, e1 in rec["cal"])
    empty = cal[len(cal) // 2] if cal else 0.0
    ms, n = {}, {}
    for family, e0, e1 in rec["events"]:
        ms[family] = ms.get(family, 0.0) + max(e0.elapsed_time(e1) - empty, 0.0)
        n[family] = n.get(family, 0) + 1
    return ms, n, empty


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().kvq_last_error()
        raise KvqError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")


def io_dtype_of(t) -> int:
    import torch
    if t.dtype == torch.float32:
        return KVQ_F32
    if t.dtype == torch.bfloat16:
        return KVQ_BF16
    raise KvqError(f"kvq ops take float32 or bfloat16 activations, got {t.dtype}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise KvqError(
                "kvq ops run only on an MI355X (HIP) device: got a CPU tensor. There is deliberately no CPU "
                "fallback; the CPU restatement lives in oracle/ and is test infrastructure only.")


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
