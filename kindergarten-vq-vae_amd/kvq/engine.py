"""TrainEngine -- the MI355X execution plan of one Shelgon / Bagon training step (models/shelgon3/Trainer.py:65-124).

The reference builds an autograd graph of ~2k ATen ops per step and lets torch.optim.Adam walk 400 parameter tensors.
This engine runs the same mathematics as an explicit forward / backward schedule over FLAT buffers:

  memory    every parameter of the model lives in one f32 master buffer (the nn.Parameters are re-pointed at views
            of it, so state_dict / checkpoints / the HF modules keep working), mirrored by one bf16 "shadow" buffer the
            GEMMs read, one bf16 gradient buffer the weight-gradient GEMMs write into directly, and f32 Adam moments.
            Q/K/V (and cross-attention K/V) weights are adjacent, so the fused [3H,H] projection is a plain view.
  forward   per block: one GEMM (+bias epilogue) -> one fused kernel (attention core | dropout+residual+LayerNorm | GELU)
  backward  hand-written mirror of forward; dropout masks are regenerated (Philox), never stored; bias gradients are
            column sums; weight gradients land in the flat gradient buffer with no accumulate/cast passes
  update    ONE Adam kernel over the flat buffers that also refreshes the bf16 shadow weights for the next step
  multi-GPU gradients are laid out in forward order, so backward finishes them from the end of the buffer towards
            the start: fixed-size tail chunks are all-reduced (RCCL, AVG, bf16) on a side stream while backward continues

GEMMs: every product of the bf16 step runs in libkvq.so -- csrc/kvq_gemm2.hip (LDS-DMA + MFMA; the tile is chosen by the cost
model kvq.nnops.pick_tile over (M, N, K), the persistent form by nnops.persistent_pays) or, for shapes that miss its divisibility /
alignment requirements, csrc/kvq_gemm_any.hip.  No vendor-library GEMM is reachable from a bf16 engine
(tests/test_engine_small_batches_gpu.py patches torch.mm / addmm / bmm / matmul to raise); the f32 engine, which exists as the
checker of the bf16 one, multiplies with torch (hipBLASLt, tuning/tunableop_gfx950.csv).
Math restated from HuggingFace modeling_bert.py (see kvq/bert.py for the line map); tests/test_engine_gpu.py checks
losses and every parameter gradient against the autograd path (kvq.bert + torch autograd).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import nnops
from ._ffi import KvqError, check, io_dtype_of, lib, stream_ptr
from .functional import _workspace

V_ALIGN = 64   # vocabulary rows of the LM head are padded so logits rows are 128-byte aligned

# model.__dict__["_kvq_engine"] = the engine that owns the model's flat buffers (Bagon / Shelgon.forward reuse it).  Model and
# engine reference each other, so the pair is collected together -- a registry keyed by the model would keep both alive
# (the engine holds the model strongly) together with ~4 GB of flat buffers at bert-base.
_QUANTIZERS = ("VectorQuantizer", "MultiVectorQuantizer", "GumbelQuantizer")


def engine_of(model, create=True):
    """The TrainEngine of `model` (its parameters live in that engine's flat buffers), created on first use."""
    eng = model.__dict__.get("_kvq_engine")
    if eng is not None and eng.flat.master.device != next(model.parameters()).device:
        eng = None                                         # the model was moved after the engine was built
    if eng is None and create:
        eng = TrainEngine(model)
    return eng


def fp8_span_table(offs, ns, n_total, span=2048):
    """span_segment of kvq_adam_step_dev_fp8 (include/kvq.h): for every `span`-element stretch of the flat parameter buffer the
    quantisation segment (offs[i], ns[i]) that covers ALL of it, -1 when no segment touches it, -2 when it holds a segment boundary
    (the kernel then walks the segment table for its 8 elements).  Segments must not overlap."""
    import numpy as np
    table = np.full((n_total + span - 1) // span, -1, dtype=np.int32)
    for si, (o, n) in enumerate(zip(offs, ns)):
        if n <= 0:
            continue
        k0, k1 = o // span, (o + n - 1) // span
        for k in range(k0, k1 + 1):
            lo, hi = span * k, min(span * k + span, n_total)
            whole = o <= lo and o + n >= hi
            table[k] = si if (whole and table[k] == -1) else -2
    return table


class _EngineForward(torch.autograd.Function):
    """model.forward with autograd ON, on the engine's kernels: forward = TrainEngine.forward_backward(defer_backward=True),
    backward = TrainEngine.backward_from(d logits, d loss_vq).  The model's parameters are the function's inputs, so autograd
    accumulates into their .grad as for any other op.  (The ATen restatement kvq/bert.py stays the default when torch must
    differentiate: it is the independent checker of this schedule.  Opt in per model: model.autograd_backend = "engine".)"""

    @staticmethod
    def forward(ctx, eng, ids, mask, training, q_training, *params):
        out = eng.forward_backward(ids, mask, training=training, compute_grads=False, want_logits=True, quantizer_training=q_training,
                                   defer_backward=True)
        ctx.eng, ctx.resume, ctx.params = eng, out["_resume"], params
        logits = out["logits"]
        ctx.logits_shape = tuple(logits.shape)
        loss_vq = out["loss_vq_raw"] if out.get("loss_vq_raw") is not None else torch.zeros((), device=logits.device)
        perp = out["perplexity"] if out.get("perplexity") is not None else torch.zeros((), device=logits.device)
        idx = out["indices"] if out.get("indices") is not None else torch.zeros(0, dtype=torch.int64, device=logits.device)
        loss_vq, perp = loss_vq.clone(), perp.clone()          # (views of one two-element result buffer otherwise)
        ctx.mark_non_differentiable(perp, idx)
        return logits, loss_vq, perp, idx

    @staticmethod
    def backward(ctx, g_logits, g_loss_vq, _g_perp, _g_idx):
        eng = ctx.eng
        if g_logits is None:                                   # a loss of the quantiser term alone
            g_logits = torch.zeros(ctx.logits_shape, device=eng.dev, dtype=eng.dtype)
        eng.backward_from(ctx.resume, g_logits.to(eng.dtype), g_loss_vq)
        by_p = eng.grads_by_parameter()
        return (None, None, None, None, None) + tuple(by_p.get(p) if p.requires_grad else None for p in ctx.params)


def engine_autograd_forward(model, ids, mask, q_training=None):
    """(logits [B, S, V], loss_vq (unweighted), perplexity, indices) of `model` with an autograd edge to every parameter."""
    eng = engine_of(model)
    eng.refresh_if_params_changed()
    params = tuple(p for p in model.parameters())
    return _EngineForward.apply(eng, ids, mask, model.training, q_training, *params)


def _round_up(x, a):
    return (x + a - 1) // a * a


_TUNING_LOADED = False


def _load_gemm_tuning():
    """hipBLASLt/rocBLAS solution choices for this step's GEMM shapes, tuned once on an MI355X with PyTorch's TunableOp
    and shipped in tuning/ (a lookup table of library solution ids; no tuning happens at run time).  Shapes not in the
    table use the library's default heuristic."""
    global _TUNING_LOADED
    if _TUNING_LOADED:
        return
    _TUNING_LOADED = True
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning", "tunableop_gfx950.csv")
    if os.environ.get("KVQ_GEMM_TUNING", "1") == "0" or not os.path.exists(path):
        return
    try:
        import torch.cuda.tunable as tn
        tn.enable(True)
        tn.tuning_enable(os.environ.get("KVQ_GEMM_TUNING") == "tune")
        if os.environ.get("KVQ_GEMM_TUNING") == "tune":
            tn.set_filename(os.environ.get("KVQ_GEMM_TUNING_OUT", "gpurun_out/tunableop_new.csv"), False)
            tn.set_max_tuning_duration(int(os.environ.get("KVQ_GEMM_TUNING_MS", 15)))
            if os.environ.get("KVQ_GEMM_TUNING_ITERS"):
                tn.set_max_tuning_iterations(int(os.environ["KVQ_GEMM_TUNING_ITERS"]))
        else:
            tn.read_file(path)
            if hasattr(tn, "write_file_on_exit"):
                tn.write_file_on_exit(False)
    except Exception as e:   # tuning tables are an optimisation only
        print(f"[kvq] GEMM tuning table not loaded: {e}")


class FlatParams:
    """Flat master / shadow / gradient / moment buffers over a list of (name, parameter, padded_numel)."""

    def __init__(self, entries: List[Tuple[str, torch.nn.Parameter, int]], device, compute_dtype, amsgrad: bool):
        self.compute_dtype = compute_dtype
        self.seg: Dict[str, Tuple[int, int, torch.Size]] = {}
        self.trainable: Dict[str, bool] = {}
        off = 0
        for name, p, padded in entries:
            self.seg[name] = (off, p.numel(), p.shape)
            self.trainable[name] = p.requires_grad
            off += _round_up(padded, 16)       # 16 elements: an fp8 mirror of a segment stays 16-byte aligned
        self.n = _round_up(off, 16)
        self.master = torch.zeros(self.n, dtype=torch.float32, device=device)
        for name, p, _ in entries:
            o, n, shape = self.seg[name]
            self.master[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.master[o:o + n].view(shape)          # the module's parameter IS the master copy now
        if compute_dtype == torch.float32:
            self.shadow = self.master
        else:
            self.shadow = self.master.to(compute_dtype)
        # gradient and Adam-moment buffers on first use: an engine that only ever serves Bagon / Shelgon.forward (no autograd,
        # engine_of(model)) never touches them -- 2.5 of the 4.5 GB of flat buffers at bert-base.  The first training steps run
        # eagerly, so the allocation never falls inside a hipGraph capture.
        self._device, self._amsgrad = device, amsgrad
        self._grad = self._m = self._v = self._vmax = None
        # contiguous trainable ranges (Adam is launched once per range; one range in `full` mode)
        self.ranges: List[Tuple[int, int]] = []
        for name, p, padded in entries:
            if not p.requires_grad:
                continue
            o = self.seg[name][0]
            e = o + _round_up(padded, 16)
            if self.ranges and self.ranges[-1][1] == o:
                self.ranges[-1] = (self.ranges[-1][0], e)
            else:
                self.ranges.append((o, e))

    def _lazy(self, attr, dtype):
        t = getattr(self, attr)
        if t is None:
            t = torch.zeros(self.n, dtype=dtype, device=self._device)
            setattr(self, attr, t)
        return t

    @property
    def grad(self):
        return self._lazy("_grad", self.compute_dtype)

    @property
    def m(self):
        return self._lazy("_m", torch.float32)

    @property
    def v(self):
        return self._lazy("_v", torch.float32)

    @property
    def vmax(self):
        return self._lazy("_vmax", torch.float32) if self._amsgrad else None

    def optimizer_state_allocated(self) -> bool:
        return self._grad is not None or self._m is not None or self._v is not None

    def w(self, name, rows=None):
        o, n, shape = self.seg[name]
        if rows is not None:                                       # padded 2-D view [rows, shape[1]]
            return self.shadow[o:o + rows * shape[1]].view(rows, shape[1])
        return self.shadow[o:o + n].view(shape)

    def g(self, name, rows=None):
        o, n, shape = self.seg[name]
        if rows is not None:
            return self.grad[o:o + rows * shape[-1]].view(rows, shape[-1]) if len(shape) > 1 else self.grad[o:o + rows]
        return self.grad[o:o + n].view(shape)

    def w32(self, name):
        o, n, shape = self.seg[name]
        return self.master[o:o + n].view(shape)

    def fused(self, names, buf):
        """View spanning adjacent segments (e.g. q,k,v weights) as one [sum rows, cols] matrix / [sum] vector."""
        o0, _, shape0 = self.seg[names[0]]
        total = 0
        for nm in names:
            o, n, _ = self.seg[nm]
            assert o == o0 + total, f"segments {names} are not adjacent"
            total += n
        flat = buf[o0:o0 + total]
        return flat.view(-1, shape0[1]) if len(shape0) == 2 else flat

    def refresh_shadow(self):
        if self.shadow is not self.master:
            self.shadow.copy_(self.master)


class _StepGraphs:
    """One training step of a fixed batch shape as a chain of hipGraphs with eager launches in between.

    The step is captured once; wherever the engine has something that must stay an ordinary launch it calls interlude(fn):
    the running capture ends, fn is kept (and run once), the next capture begins in the same memory pool.  Interludes are
      * the quantiser forward (kvq_vq_forward: the kernel bench.py times with HIP events on every launch), and
      * on multi-GPU runs every gradient all-reduce (RCCL, issued on the side stream while the next graph runs) and the
        final wait for them before the Adam graph.
    Replay = graph, interlude, graph, ...: ~3 host launches per step on one GPU instead of ~800.  Nothing inside the graphs
    depends on host values that change between steps: see the device step state in include/kvq.h."""

    def __init__(self, eng, ids, mask, dec=None):
        self.eng = eng
        dev = eng.dev
        # static input buffers the captured kernels read: ids | mask | ids sorted (stable, pads under -1) | their order
        # [| the decoder's own ids | mask | sorted | order | the loss target -- a Bagon step, models/bagon/Trainer.py:65-130]
        self.pack = eng.pack_batch(ids, mask, *(dec or ())).reshape(-1)
        N, H = ids.numel(), eng.H
        self.ids, self.mask, srt, perm, self.dec = eng.unpack_batch(self.pack, ids.shape, dec[0].shape if dec else None)
        self.sorted = (srt, perm)
        self.z_q = torch.empty((N, H), dtype=eng.dtype, device=dev)
        self.idx = torch.empty(N * max(eng.G, 1), dtype=torch.int64, device=dev)
        self.scal = torch.zeros(4, dtype=torch.float32, device=dev)         # loss, accuracy | quantiser loss, perplexity: one clone per step
        self.vq_out = self.scal[2:4]
        # keep_graph: the captured hipGraph_t stays readable behind the executable one (node_census; a few hundred nodes)
        self.graphs, self.inter = [torch.cuda.CUDAGraph(keep_graph=True)], []
        side = torch.cuda.Stream(device=dev)
        # no garbage collection while a capture is open (what torch.cuda.graph does as well): a collected cycle may own HIP
        # objects -- an older engine's kept hipGraphs -- whose destruction is not permitted while this thread captures
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        torch.cuda.synchronize(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        step0 = eng._step_host
        ok = False
        with torch.cuda.stream(side):
            eng._cap = self
            try:
                # thread-local capture mode: the process group's helper threads (gloo copies, the RCCL watchdog's event queries)
                # keep issuing HIP calls on their own streams while this thread captures
                self.graphs[0].capture_begin(capture_error_mode="thread_local")
                eng._prepared = dict(enc=self.sorted, dec=self.dec[2:4] if self.dec else None)
                dkw = dict(dec_ids=self.dec[0], dec_mask=self.dec[1], target_ids=self.dec[4]) if self.dec else {}
                self.out = eng.forward_backward(self.ids, self.mask, training=eng.model.training, compute_grads=True, fuse_optimizer=True,
                                                **dkw)
                eng.optimizer_step()
                self.graphs[-1].capture_end()
                ok = True
            finally:
                eng._cap = None
                eng._prepared = None
                eng._step_host = step0      # capturing ran no captured kernel: the device step state did not move either
                if not ok:                  # close the capture that was open when the error struck
                    try:
                        eng._adam_join()    # (a forked side stream must be back in the origin stream before the capture can end)
                        self.graphs[-1].capture_end()
                    except Exception:
                        pass
                if gc_was_on:
                    gc.enable()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

    def interlude(self, fn):
        """Called by the engine in the middle of the captured step: fn stays an eager launch between two graphs."""
        self.graphs[-1].capture_end()
        fn()
        self.inter.append(fn)
        g = torch.cuda.CUDAGraph(keep_graph=True)
        g.capture_begin(pool=self.graphs[0].pool(), capture_error_mode="thread_local")
        self.graphs.append(g)

    def node_census(self):
        """Per captured graph of the chain: {kind: nodes} (kvq_graph_census).  A step graph must consist of kernel nodes (and the
        empty / event nodes of a stream fork) only: a memset or memcpy node is a launch of another kind whose order against the
        neighbouring kernel nodes this stack did not keep (profiles/r04_fp8.md) -- tests/test_graph_nodes_gpu.py."""
        import ctypes
        kinds = ("kernel", "memset", "memcpy", "empty", "event", "other")
        out = []
        for g in self.graphs:
            counts = (ctypes.c_int64 * len(kinds))()
            check(lib().kvq_graph_census(g.raw_cuda_graph(), counts), "kvq_graph_census")
            out.append(dict(zip(kinds, (int(c) for c in counts))))
        return out

    def run(self, ids, mask, prep, dec=None):
        """prep: TrainEngine._normalise_prepared()'s dict (pack = the whole batch as one tensor, or the sorted ids of either side)."""
        if prep["pack"] is not None:
            self.pack.copy_(prep["pack"])           # a batch packed where it was built (TrainEngine.pack_batch): one copy
        else:                                       # otherwise the sort happens here -- outside the graphs, but in the step
            self.ids.copy_(ids)
            self.mask.copy_(mask)
            srt, perm = prep["enc"] if prep["enc"] is not None else self.eng.prepare_batch(ids)
            self.sorted[0].copy_(srt)
            self.sorted[1].copy_(perm)
            if self.dec:
                d_ids, d_mask, tgt = dec
                self.dec[0].copy_(d_ids)
                self.dec[1].copy_(d_mask)
                srt, perm = prep["dec"] if prep["dec"] is not None else self.eng.prepare_batch(d_ids)
                self.dec[2].copy_(srt)
                self.dec[3].copy_(perm)
                self.dec[4].copy_(d_ids if tgt is None else tgt)
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.inter):
                self.inter[i]()
        self.eng._step_host += 1
        # the graph's buffers are overwritten by the next step: hand out copies (the four scalars as views of ONE copy)
        sc = self.scal.clone()
        res = {}
        for k, v in self.out.items():
            if v is None:
                res[k] = None
            elif v.untyped_storage().data_ptr() == self.scal.untyped_storage().data_ptr():
                res[k] = sc[v.storage_offset()]
            else:
                res[k] = v.clone()
        return res


class TrainEngine:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False,
                 milestones=None, gamma=0.1, loss_recon_scale=1.0, loss_vq_scale=1.0, seed=1234,
                 bucket_mib=64, process_group=None, fp8_forward=None):
        self.model = model
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise KvqError("TrainEngine needs the model on an MI355X (there is no CPU path)")
        lib()
        self.dev = dev
        self.dtype = model.compute_dtype
        self.io = 1 if self.dtype == torch.bfloat16 else 0
        self.has_vq = hasattr(model, "vector_quantizer")
        self.sentence_acc = not self.has_vq       # the plain Bagon step reports the accuracy of every sentence as well
        self.vq_kind = type(model.vector_quantizer).__name__ if self.has_vq else None
        if self.has_vq and self.vq_kind not in _QUANTIZERS:
            raise KvqError(f"TrainEngine schedules {', '.join(_QUANTIZERS)}; got {self.vq_kind}")
        self.G = 1
        enc, dec = model.encoder, model.decoder
        self.ecfg, self.dcfg = enc.config, dec.config
        if self.ecfg.hidden_size // self.ecfg.num_attention_heads != 64:
            raise KvqError("TrainEngine: attention kernels are written for head dim 64")
        self.H = self.ecfg.hidden_size
        self.V = self.dcfg.vocab_size
        self.Vp = _round_up(self.V, V_ALIGN)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.milestones, self.gamma = sorted(milestones or []), gamma
        self.w_recon, self.w_vq = float(loss_recon_scale), float(loss_vq_scale)
        self.seed = int(seed)
        if len(self.milestones) > 8:
            raise KvqError("TrainEngine: at most 8 LR milestones (kvq_step_state_advance)")
        # what changes from step to step lives on the device (include/kvq.h "step state"): dropout kernels add the step count
        # to their seed, Adam reads lr and the bias corrections -- so a captured hipGraph of the whole step can be replayed
        self._state = nnops.new_step_state(dev)
        self._step_host = 0
        self._step_seed = (self.seed * 1000003) & 0x7FFFFFFFFFFFFFF      # + step count on the device
        self._graphs, self._eager_seen, self._cap, self._prepared = {}, {}, None, None
        self._own_fwd = os.environ.get("KVQ_OWN_GEMM", "1") != "0"
        self._gumbel_own = 0                  # products of the Gumbel mode that ran on the own GEMM (tests read it)
        if os.environ.get("KVQ_GEMM_TILE"):       # "NxK:tile;NxK:tile" (A/B runs on the GPU box)
            self._TILE_OVERRIDE = {tuple(int(v) for v in e.split(":")[0].split("x")): e.split(":")[1]
                                   for e in os.environ["KVQ_GEMM_TILE"].split(";") if e}
        self._own_epi = os.environ.get("KVQ_OWN_EPI", "1") != "0"     # A/B switch: 0 = activations as separate kernels behind the GEMMs
        # round 5: dropout + residual of a BertSelfOutput / BertOutput block in the dense layer's epilogue, LayerNorm alone behind it
        # (8.2 us instead of 11.9: one tensor read, one written).  Bit-identical -- and only worth it WITHOUT dropout: the Philox
        # rounds of 6 M elements are ~5 us of vector-ALU time per SIMD, which the LayerNorm kernel hides behind its loads and a
        # GEMM epilogue does not (same-box traces, profiles/r05_gemm_ceiling.md: +7 us per GEMM against -3.7 us per LayerNorm).
        # "eval" (default): steps without dropout (validation / test stages, model.forward); "1": always (tests); "0": never.
        self._fuse_dropres = os.environ.get("KVQ_FUSE_DROPRES", "eval")
        # weight gradients of a whole layer as ONE grouped launch of csrc/kvq_gemm2.hip (KVQ_OWN_WGRAD=0: library + split-K slabs)
        self._own_wgrad = self._own_fwd and os.environ.get("KVQ_OWN_WGRAD", "1") != "0" and self.dtype == torch.bfloat16
        self._wg_items, self._wg_keep = [], []
        self._wg_pair = os.environ.get("KVQ_WG_PAIR", "1") != "0"     # A/B switch: 0 = one grouped launch per layer (128 x 256 tiles)
        self._red_pair = os.environ.get("KVQ_RED_PAIR", "1") != "0"   # A/B switch: 0 = the batched reductions stay per layer
        # LM head + the loss' forward statistics in one kernel (see _forward_backward): built and parity-tested in round 2, and
        # 0.08 ms/step SLOWER than library GEMM + kvq_ce_forward (18.55 against 18.47 ms, gpurun_out/ab16.log): opt-in
        self._own_lmce = os.environ.get("KVQ_OWN_LMCE", "0") == "1"
        # opt-in (KVQ_EARLY_ADAM=1): Adam for a layer's parameters as soon as their gradients are final, on a side stream beside the
        # rest of backward (one GPU only).  The update is pure HBM streaming and the GEMMs beside it live on L2 -> LDS bandwidth,
        # yet on MI355X the step got SLOWER: 18.88 against 18.18 ms (gpurun_out/ab6.log), and 19.5-19.8 against 18.86 with the
        # Adam grid capped at 128-1024 workgroups (ab7.log) -- like the side-stream weight gradients, a second kernel on the
        # CUs costs the one-workgroup-per-CU GEMMs more than the overlap returns.  Off by default.
        self._early_adam = os.environ.get("KVQ_EARLY_ADAM", "0") == "1"
        self.adam_stream = torch.cuda.Stream(device=dev) if self._early_adam else None
        self._fuse_opt, self._adam_hi, self._adam_forked = False, 0, False
        # opt-in (KVQ_WG_STREAM=1): weight-gradient GEMMs on a side stream.  Measured on MI355X with the step replayed from
        # hipGraphs: 23.6 ms/step against 22.5 ms on one stream -- two concurrent hipBLASLt kernels share CUs and L2 badly
        self.wg_stream = torch.cuda.Stream(device=dev) if os.environ.get("KVQ_WG_STREAM", "0") == "1" else None
        self._wg_pending = False
        self._wg_keep_step = []
        self.use_graph = os.environ.get("KVQ_GRAPH", "1") != "0"
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # the gradient exchange runs when there is more than one rank -- or, for a rehearsal of the RCCL branch on a one-GPU box,
        # with a ONE-rank process group and KVQ_DP_SINGLE_RANK=1 (every all-reduce, stream hand-over and graph interlude executes;
        # the averages are over one rank)
        self._dp = self.world > 1 or (dist.is_initialized() and os.environ.get("KVQ_DP_SINGLE_RANK", "0") == "1")

        # ---- flat parameter layout, in FORWARD order (gradients then complete from the tail backwards)
        entries, seen = [], set()
        self.param_of: Dict[str, torch.nn.Parameter] = {}

        def add(name, p, padded=None):
            if id(p) in seen:
                return
            seen.add(id(p))
            self.param_of[name] = p
            entries.append((name, p, padded if padded is not None else p.numel()))

        def add_emb(prefix, emb, pad_word_rows=None):
            w = emb.word_embeddings.weight
            add(prefix + "word", w, (pad_word_rows or w.shape[0]) * w.shape[1])
            add(prefix + "pos", emb.position_embeddings.weight)
            add(prefix + "type", emb.token_type_embeddings.weight)
            add(prefix + "ln.w", emb.LayerNorm.weight); add(prefix + "ln.b", emb.LayerNorm.bias)

        def add_attn(prefix, att, cross):
            s = att.self
            if cross:
                add(prefix + "q.w", s.query.weight)
                add(prefix + "k.w", s.key.weight); add(prefix + "v.w", s.value.weight)
                add(prefix + "q.b", s.query.bias)
                add(prefix + "k.b", s.key.bias); add(prefix + "v.b", s.value.bias)
            else:
                add(prefix + "q.w", s.query.weight); add(prefix + "k.w", s.key.weight); add(prefix + "v.w", s.value.weight)
                add(prefix + "q.b", s.query.bias); add(prefix + "k.b", s.key.bias); add(prefix + "v.b", s.value.bias)
            add(prefix + "o.w", att.output.dense.weight); add(prefix + "o.b", att.output.dense.bias)
            add(prefix + "ln.w", att.output.LayerNorm.weight); add(prefix + "ln.b", att.output.LayerNorm.bias)

        def add_layer(prefix, layer, cross):
            add_attn(prefix + "sa.", layer.attention, False)
            if cross:
                add_attn(prefix + "ca.", layer.crossattention, True)
            add(prefix + "f1.w", layer.intermediate.dense.weight); add(prefix + "f1.b", layer.intermediate.dense.bias)
            add(prefix + "f2.w", layer.output.dense.weight); add(prefix + "f2.b", layer.output.dense.bias)
            add(prefix + "ln2.w", layer.output.LayerNorm.weight); add(prefix + "ln2.b", layer.output.LayerNorm.bias)

        add_emb("enc.emb.", enc.embeddings)
        for i, layer in enumerate(enc.encoder.layer):
            add_layer(f"enc.{i}.", layer, False)
        add_emb("dec.emb.", dec.bert.embeddings, pad_word_rows=self.Vp)     # tied to the LM head: padded to Vp rows
        # Every decoder layer projects the SAME encoder output (the quantised z_q) to its cross-attention keys / values
        # (modeling_bert.py:83-85 with encoder_hidden_states): their weights sit side by side so that the 12 projections, their
        # input gradients and their weight gradients each run as ONE GEMM over [L*2H, H] instead of 12 narrow ones.
        dlayers = list(dec.bert.encoder.layer)
        self._cakv_w = [n for i in range(len(dlayers)) for n in (f"dec.{i}.ca.k.w", f"dec.{i}.ca.v.w")]
        self._cakv_b = [n for i in range(len(dlayers)) for n in (f"dec.{i}.ca.k.b", f"dec.{i}.ca.v.b")]
        for i, layer in enumerate(dlayers):
            sx = layer.crossattention.self
            add(f"dec.{i}.ca.k.w", sx.key.weight); add(f"dec.{i}.ca.v.w", sx.value.weight)
        for i, layer in enumerate(dlayers):
            sx = layer.crossattention.self
            add(f"dec.{i}.ca.k.b", sx.key.bias); add(f"dec.{i}.ca.v.b", sx.value.bias)
        for i, layer in enumerate(dlayers):
            add_layer(f"dec.{i}.", layer, True)
        head = dec.cls.predictions
        add("head.t.w", head.transform.dense.weight); add("head.t.b", head.transform.dense.bias)
        add("head.ln.w", head.transform.LayerNorm.weight); add("head.ln.b", head.transform.LayerNorm.bias)
        add("head.bias", head.decoder.bias, self.Vp)
        if head.decoder.weight is not dec.bert.embeddings.word_embeddings.weight:
            raise KvqError("TrainEngine expects the LM head tied to the decoder word embeddings (HF default)")
        # parameters the step never touches (pooler) stay outside the flat buffers and receive no gradient
        self.flat = FlatParams(entries, dev, self.dtype, amsgrad)
        pads = {enc.embeddings.word_embeddings.padding_idx, dec.bert.embeddings.word_embeddings.padding_idx}
        if len(pads) != 1:
            raise KvqError("TrainEngine expects the same padding_idx in the encoder and decoder word embeddings")
        self._pad_idx = pads.pop()
        self.n_enc_layers = len(enc.encoder.layer)
        self.n_dec_layers = len(dec.bert.encoder.layer)
        self.nh = self.ecfg.num_attention_heads
        flags = {self.flat.trainable[n] for n in self._cakv_w + self._cakv_b}
        self._cakv_batched = len(flags) == 1 and self.n_dec_layers > 0       # mixed frozen / trained layers: per-layer GEMMs
        # small f32 parameters outside the flat buffers (codebook, Gumbel projection / embedding): own gradient + Adam state
        self.aux = []
        self._bufs = {}

        def add_aux(p):
            a = dict(p=p, g=torch.zeros_like(p.data), m=torch.zeros_like(p.data), v=torch.zeros_like(p.data),
                     vmax=torch.zeros_like(p.data) if amsgrad else None)
            self.aux.append(a)
            return a["g"]

        self.vq_ema = False
        if self.vq_kind in ("VectorQuantizer", "MultiVectorQuantizer"):
            vq = model.vector_quantizer
            self.G = int(getattr(vq, "n_factors", 1))         # codebooks = slices of the encoder output, one grouped launch
            self.K, self.Dg = int(vq.n_e), int(getattr(vq, "d_factor", self.H))       # (padded) slice width
            self.E = vq.embedding.weight                      # f32 [G*K, H/G]
            if self.E.shape != (self.G * self.K, self.Dg):
                raise KvqError(f"TrainEngine: codebook {tuple(self.E.shape)} does not match {self.G} x {self.K} x {self.Dg}")
            self.beta_vq = float(vq.beta)
            self.vq_ema = getattr(vq, "ema_decay", None) is not None     # extension: EMA codebook update instead of the gradient
            self.gE = add_aux(self.E) if self.E.requires_grad else torch.zeros_like(self.E.data)
            # the codebook in MFMA-fragment order (kvq_vq_pack_codebook): rebuilt after every codebook update -- Adam at the end
            # of the step, the EMA step, or a write from outside (noticed through the tensor version) -- not per forward call
            self._epack = torch.empty(lib().kvq_vq_packed_bytes(self.K, self.Dg, self.G), dtype=torch.uint8, device=dev) \
                if self.Dg % 32 == 0 else None
            self._E_version = None
        elif self.vq_kind == "GumbelQuantizer":
            gq = model.vector_quantizer
            if gq.n_embed > 1024:
                raise KvqError("TrainEngine: the Gumbel row kernel holds at most 1024 codes")
            self.g_pw, self.g_pb, self.g_emb = add_aux(gq.proj.weight), add_aux(gq.proj.bias), add_aux(gq.embed.weight)
        # extension (BASELINE.json configs[4], default off): forward GEMMs on the fp8 matrix cores; per-tensor scales -- weights from
        # the tensor itself after every update, activations delayed by one training step (kvq_fp8_quantize_delayed).
        # True / KVQ_FP8=1: the products where fp8 + the activation's quantisation pass beat the own bf16 kernel -- outputs at least
        # _FP8_MIN_ROWS wide: the LM head and the all-layer cross-K/V projection (tools/gemm2_probe_fp8_own.py, profiles/r04_fp8.md:
        # 292 against 350 us and 184 against 217; every per-layer GEMM LOSES 4 - 8 us to its quantisation pass).
        # "all" / KVQ_FP8=all: every forward GEMM (rounds 2 - 3; kept for the numerics tests and as the A/B arm).
        # "fused" / KVQ_FP8=fused (round 5): every forward GEMM of the layers, each reading an fp8 copy of its input that the kernel
        # PRODUCING that input wrote beside the bf16 one (LayerNorm forward, attention forward, the GELU epilogue of the fp8
        # BertIntermediate GEMM: kvq_*_fp8) -- the 120 quantisation passes of "all" are gone; what still takes a pass is the input
        # of a layer-0 QKV projection (embedding kernel) and of the cross-K/V projection (quantiser output).
        # Since round 5 this is what fp8_forward=True / KVQ_FP8=1 means (same box, nine codebooks, profiles/r05_fp8.md: 16.96 ms against
        # 17.05 for "wide" and 17.14 for bf16); "wide" = rounds 4's two GEMMs, "all" = every GEMM behind a quantisation pass (A/B arm).
        if fp8_forward is None:
            fp8_forward = {"0": False, "1": True, "all": "all", "fused": "fused", "wide": "wide"}.get(os.environ.get("KVQ_FP8", "0"), False)
        if fp8_forward is True:
            fp8_forward = "fused"
        if fp8_forward not in (False, "fused", "wide", "all"):
            raise KvqError(f"TrainEngine: fp8_forward must be False, True / \"fused\", \"wide\" or \"all\", got {fp8_forward!r}")
        self.fp8 = bool(fp8_forward)
        self._fp8_fused = fp8_forward == "fused"
        self._fp8_all = fp8_forward in ("all", "fused")
        self._x8 = {}                      # data_ptr of a bf16 activation -> its fp8 copy, left by the producer for ONE fp8 GEMM
        if self.fp8:
            if self.dtype != torch.bfloat16:
                raise KvqError("TrainEngine: fp8 forward GEMMs need the bf16 compute dtype")
            self._fp8_setup()
        model.__dict__["_kvq_engine"] = self
        self._param_versions = self._versions()
        # gradient all-reduce chunks (tail first)
        self.comm_stream = torch.cuda.Stream(device=dev) if self._dp else None
        self._avg_in_comm = self._dp and dist.get_backend(process_group) == "nccl"   # RCCL averages itself; gloo sums
        self.chunk = max(bucket_mib * (1 << 20) // torch.empty(0, dtype=self.flat.compute_dtype).element_size(), 1 << 16)     # elements per all-reduce chunk
        self._pending_hi = self._wg_done_lo = self.flat.n
        self._works, self._works_late = [], []
        self._ev_early = torch.cuda.Event() if self._dp else None
        self._time_comm = False
        self._comm_ev = []                       # (start, end) event pairs around the points where the compute stream waits for RCCL
        self._ones = torch.ones((), dtype=torch.float32, device=dev)
        self._g_recon = torch.full((), self.w_recon, dtype=torch.float32, device=dev)     # d total / d loss term: constants of the run
        self._g_vq = torch.full((), self.w_vq, dtype=torch.float32, device=dev)
        self._g_vq_ext = None                  # backward_from(): the caller's d L / d loss_vq_raw instead of the run's constant
        _load_gemm_tuning()

    # ------------------------------------------------------------------------------------------------------------
    # small helpers
    # ------------------------------------------------------------------------------------------------------------
    def __deepcopy__(self, memo):
        return None                   # copy.deepcopy(model): the copy builds its own engine on first use (engine_of)

    def __reduce__(self):
        return (type(None), ())       # torch.save(model): the engine (streams, graphs, flat mirrors) is not part of a checkpoint

    @property
    def pad_idx(self):
        """padding_idx of the word embeddings (None = no padding row): the id a packed batch files under -1 (prepare_batch)."""
        return self._pad_idx

    @property
    def step_count(self):
        """Optimiser steps applied so far (host mirror of the device step state)."""
        return self._step_host

    @step_count.setter
    def step_count(self, n):
        self._step_host = int(n)
        self._state[0] = int(n)          # lr / bias corrections are recomputed by the next kvq_step_state_advance

    def _lr_now(self):
        """MultiStepLR ticked once per optimiser step (Trainer.py:114-115): the s-th step (1-based) sees s-1 ticks."""
        ticks = self.step_count - 1
        k = sum(1 for m in self.milestones if ticks >= m)
        return self.lr * (self.gamma ** k)

    def _site(self):
        self._site_ctr += 1
        return self._site_ctr

    # Tile choice: kvq.nnops.pick_tile / persistent_pays (a cost model over (M, N, K) fitted to tools/gemm2_probe_small.py and
    # consistent with the per-shape choices measured at 8192 rows in rounds 2 - 3, which it replaced).  KVQ_GEMM_TILE="NxK:tile;..."
    # overrides it per weight shape for A/B runs on the GPU box.
    _TILE_OVERRIDE: Dict[Tuple[int, int], str] = {}

    # ---- fp8 forward GEMMs -------------------------------------------------------------------------------------------------
    _FP8_MIN_ROWS = 8192          # output width from which fp8 + its quantisation pass beat the bf16 kernel (see __init__)

    def _fp8_setup(self):
        """One fp8 mirror of the flat bf16 shadow buffer; one quantisation segment (= one scale) per forward GEMM weight: the
        fused q|k|v block, the all-layer cross-attention k|v block, every other [out, in] matrix, the padded LM-head table."""
        fl = self.flat
        segs = {}

        def seg(key, names, rows=None):
            o0 = fl.seg[names[0]][0]
            n = sum(fl.seg[nm][1] for nm in names) if rows is None else rows * fl.seg[names[0]][2][1]
            only = os.environ.get("KVQ_FP8_ONLY")          # diagnostic: "all" restricted to weights whose name contains one of these
            if only and not any(t in key for t in only.split(",")):
                return
            if self._fp8_all or n // fl.seg[names[0]][2][1] >= self._FP8_MIN_ROWS:
                segs[key] = (o0, n)
        for i in range(self.n_enc_layers):
            pre = f"enc.{i}."
            seg(pre + "sa.q.w", [pre + "sa.q.w", pre + "sa.k.w", pre + "sa.v.w"])
            for nm in ("sa.o.w", "f1.w", "f2.w"):
                seg(pre + nm, [pre + nm])
        for i in range(self.n_dec_layers):
            pre = f"dec.{i}."
            seg(pre + "sa.q.w", [pre + "sa.q.w", pre + "sa.k.w", pre + "sa.v.w"])
            for nm in ("sa.o.w", "ca.q.w", "ca.o.w", "f1.w", "f2.w"):
                seg(pre + nm, [pre + nm])
            if not self._cakv_batched:
                seg(pre + "ca.k.w", [pre + "ca.k.w", pre + "ca.v.w"])
        if self._cakv_batched:
            seg(self._cakv_w[0], self._cakv_w)
        seg("head.t.w", ["head.t.w"])
        seg("dec.emb.word", ["dec.emb.word"], rows=self.Vp)
        keys = list(segs)
        if not keys:
            raise KvqError("TrainEngine: fp8 forward GEMMs asked for, but no weight of this model is wide enough for them to pay "
                           f"(>= {self._FP8_MIN_ROWS} rows); fp8_forward=True / \"fused\" puts every forward GEMM on fp8")
        self._w8_index = {k: i for i, k in enumerate(keys)}
        offs = [segs[k][0] for k in keys]
        ns = [segs[k][1] for k in keys]
        assert all(o % 16 == 0 and n % 16 == 0 for o, n in zip(offs, ns))
        self._w8 = torch.zeros(fl.n, dtype=torch.uint8, device=self.dev)
        self._w8_off = torch.tensor(offs, dtype=torch.int64, device=self.dev)
        self._w8_n = torch.tensor(ns, dtype=torch.int64, device=self.dev)
        self._w8_max = max(ns)
        self._w8_period = max(int(os.environ.get("KVQ_FP8_W_PERIOD", "16")), 1)
        # the Adam kernel writes the fp8 mirror itself (kvq_adam_step_dev_fp8) when the scales are not refreshed on every step:
        # span_segment[e >> 11] = the segment covering elements [2048 k, 2048 k + 2048), -1 none, -2 several things
        self._w8_in_adam = self._w8_period > 1 and os.environ.get("KVQ_FP8_ADAM", "1") != "0"
        self._w8_span = torch.from_numpy(fp8_span_table(offs, ns, fl.n)).to(self.dev)
        self._w8_amax = torch.zeros(len(keys), dtype=torch.float32, device=self.dev)
        self._w8_scale = torch.ones(len(keys), dtype=torch.float32, device=self.dev)
        self._fp8_quantize_weights()
        # activations: one {scale, amax} pair per GEMM input of the step, in call order (delayed scaling: include/kvq.h)
        self._a8_sites = len(keys)                    # one record per fp8 GEMM = per weight key (each is used once per forward)
        st = torch.zeros((self._a8_sites, lib().kvq_fp8_state_floats()), dtype=torch.float32, device=self.dev)
        st[:, 0] = 1.0
        self._a8_state = st

    def _a8_state_of(self, key):
        """The delayed-scaling record of the fp8 GEMM with weight `key` (its input's scale and amax partials)."""
        return self._a8_state[self._w8_index[key]]

    def _emits_fp8_for(self, key):
        """True when the kernel that produces the input of GEMM `key` should write its fp8 copy too (scope "fused")."""
        return self.fp8 and self._fp8_fused and key is not None and key in self._w8_index

    def _fp8_quantize_weights(self, in_step=False):
        """The GEMM weights of the whole model to fp8 (two passes over the bf16 shadow: amax per weight, then the conversion).
        in_step: after an optimiser step of the engine -- the amax pass (and with it the scales) runs on every
        KVQ_FP8_W_PERIOD-th step only (default 16, decided from the device step count, so that graph replay follows): between
        refreshes a weight that outgrew its amax saturates, and an Adam step moves a weight by ~lr."""
        if in_step and self._w8_period > 1:
            # (negative period: the Adam kernel has written this step's bytes, the conversion pass runs on refresh steps only)
            per = -self._w8_period if self._w8_in_adam else self._w8_period
            check(lib().kvq_fp8_quantize_segments_periodic(self.flat.shadow.data_ptr(), self._w8_off.data_ptr(), self._w8_n.data_ptr(),
                                                           len(self._w8_index), self._w8_max, self._w8.data_ptr(), self._w8_amax.data_ptr(),
                                                           self._w8_scale.data_ptr(), self._state.data_ptr(), per, stream_ptr()),
                  "kvq_fp8_quantize_segments_periodic")
            return
        check(lib().kvq_fp8_quantize_segments(self.flat.shadow.data_ptr(), self._w8_off.data_ptr(), self._w8_n.data_ptr(), len(self._w8_index),
                                              self._w8_max, self._w8.data_ptr(), self._w8_amax.data_ptr(), self._w8_scale.data_ptr(),
                                              stream_ptr()), "kvq_fp8_quantize_segments")

    def _linear_fp8(self, x, W, b, key):
        si = self._w8_index[key]
        o = self.flat.seg[key][0]
        W8 = self._w8[o:o + W.numel()].view(W.shape)
        st = self._a8_state[si]
        x8 = self._fp8_input(x, st, key)
        return nnops.gemm_fp8_nt(x8, W8, st[0:], self._w8_scale[si:], bias=b)

    def _fp8_input(self, x, st, key):
        """The fp8 copy of activation x for the GEMM with weight `key` (record `st`): the one the producer of x left for exactly
        this GEMM (scope "fused"), or a quantisation pass."""
        x8 = self._x8.pop((x.data_ptr(), key), None)
        if x8 is not None and tuple(x8.shape) == tuple(x.shape):
            return x8
        x8 = torch.empty(x.shape, dtype=torch.uint8, device=self.dev)
        check(lib().kvq_fp8_quantize_delayed(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), x8.data_ptr(), st.data_ptr(), stream_ptr()),
              "kvq_fp8_quantize_delayed")
        return x8

    def _linear(self, x, wname, bname, fused=None, Wb=None, key=None):
        if Wb is not None:
            W, b = Wb
        else:
            W = self.flat.fused(fused[0], self.flat.shadow) if fused else self.flat.w(wname)
            b = self.flat.fused(fused[1], self.flat.shadow) if fused else self.flat.w(bname)
            key = fused[0][0] if fused else wname
        if self.fp8 and key in self._w8_index and x.shape[0] >= 256 and W.shape[1] % 128 == 0 and x.stride(1) == 1:
            return self._linear_fp8(x, W, b, key)
        return self._gemm(x, W, "nt", bias=b)

    def _gemm(self, a, b, layout, bias=None, out=None, accumulate=False):
        """op(a) . op(b) (+ bias) (accumulated into out).  bf16: libkvq.so, always (nnops.gemm: MFMA kernel or the any-shape one).
        f32 (the checker engine) and KVQ_OWN_GEMM=0 (A/B against the vendor library): torch."""
        if self._own_fwd and self.dtype == torch.bfloat16:
            M, N, K = nnops._gemm_dims(a, b, layout)
            wshape = (N, K) if layout == "nt" else ((K, N) if layout == "nn" else (M, N))
            return nnops.gemm(a, b, layout, bias=bias, out=out, accumulate=accumulate, tile=self._TILE_OVERRIDE.get(wshape))
        A = a.t() if layout == "tn" else a
        Bm = b.t() if layout == "nt" else b
        if out is not None and accumulate:
            return out.addmm_(A, Bm)
        if out is not None:
            return torch.mm(A, Bm, out=out) if bias is None else torch.addmm(bias, A, Bm, out=out)
        return torch.addmm(bias, A, Bm) if bias is not None else torch.mm(A, Bm)

    def _epilogue_tile(self, M, N, K):
        """Tile of the fused-activation GEMMs (they exist for 256 x 192 and 128 x 256), or None when the plain GEMM on its best
        tile + the separate activation kernel (one pass over [M, N]: ~3 us + 4 bytes per element at ~3 TB/s) is modelled cheaper --
        outputs too small to give half the CUs one of the large tiles."""
        if not (self._own_fwd and self._own_epi and self.dtype == torch.bfloat16):
            return None
        t = nnops.pick_tile(M, N, K, candidates=("256x192", "128x256"))
        plain = nnops.tile_cost_us(nnops.pick_tile(M, N, K), M, N, K) + 3.0 + M * N * 4 / 3e6
        return nnops.TILE_NAMES[t] if nnops.tile_cost_us(t, M, N, K) <= plain else None

    def _dense_residual_ln(self, a, wname, bname, resid, gname, betaname, eps, p_drop, site, next_key=None):
        """LayerNorm(dropout(a . W^T + b) + resid) of a BertSelfOutput / BertOutput block (modeling_bert.py:282-296, 339-352):
        (out, pre, mean, rstd) as nnops.ln_fwd returns them.  bf16 engine: the dense layer's epilogue draws the dropout mask, adds
        the residual and stores `pre` (nnops.gemm_dropres -- bit for bit the `pre` of the two-kernel form), then LayerNorm alone."""
        fl = self.flat
        W, b = fl.w(wname), fl.w(bname)
        gamma, beta = fl.w32(gname), fl.w32(betaname)
        if (self._fuse_dropres == "1" or (self._fuse_dropres == "eval" and p_drop == 0.0)) and self._own_fwd and self._own_epi \
                and self.dtype == torch.bfloat16 \
                and not (self.fp8 and wname in self._w8_index) and a.is_contiguous() and resid.is_contiguous() \
                and nnops.gemm_mfma_ok(a, W, None, "nt", b):
            pre = nnops.gemm_dropres(a, W, b, resid, p_drop, self._step_seed, site)
            out, _, mean, rstd = nnops.ln_fwd(pre, None, gamma, beta, eps, 0.0, 0, 0, save_pre=False)
            return out, pre, mean, rstd
        y = self._linear(a, wname, bname)
        if self._emits_fp8_for(next_key):             # next_key: the GEMM that reads this block's output
            out, pre, mean, rstd, out8 = nnops.ln_fwd_fp8(y, resid, gamma, beta, eps, p_drop, self._step_seed, site, self._a8_state_of(next_key))
            self._x8[(out.data_ptr(), next_key)] = out8
            return out, pre, mean, rstd
        return nnops.ln_fwd(y, resid, gamma, beta, eps, p_drop, self._step_seed, site)

    def _linear_gelu(self, x, wname, bname, next_key=None):
        """(h, gelu(h)), h = x . W^T + b: one kernel where the own GEMM carries the activation in its epilogue."""
        W, b = self.flat.w(wname), self.flat.w(bname)
        on_fp8 = self.fp8 and wname in self._w8_index         # (fp8 "all": the fp8 GEMM has no activation epilogue)
        if on_fp8 and self._fp8_fused and x.shape[0] >= 256 and W.shape[1] % 128 == 0 and x.is_contiguous():
            # scope "fused": BertIntermediate on the fp8 matrix cores WITH the GELU epilogue, which also leaves the fp8 copy of
            # gelu(h) for BertOutput.dense (next_key)
            si = self._w8_index[wname]
            o = self.flat.seg[wname][0]
            W8 = self._w8[o:o + W.numel()].view(W.shape)
            st = self._a8_state[si]
            x8 = self._fp8_input(x, st, wname)
            if self._emits_fp8_for(next_key):
                h, a, a8 = nnops.gemm_fp8_nt_gelu(x8, W8, st[0:], self._w8_scale[si:], b, state_out=self._a8_state_of(next_key))
                self._x8[(a.data_ptr(), next_key)] = a8
                return h, a
            return nnops.gemm_fp8_nt_gelu(x8, W8, st[0:], self._w8_scale[si:], b)
        tile = None if on_fp8 else self._epilogue_tile(x.shape[0], W.shape[0], W.shape[1])
        if tile is not None and x.is_contiguous() and nnops.gemm_mfma_ok(x, W, None, "nt", b):
            return nnops.gemm_gelu(x, W, b, tile=tile)
        h = self._linear(x, wname, bname)
        return h, nnops.gelu_fwd(h)

    # ------------------------------------------------------------------------------------------------------------
    # deferred small reductions: a layer's split-K slab sums and LayerNorm / bias partial sums go out as ONE launch
    # (kvq_reduce_batch) instead of 8-14 launch-latency-bound kernels; _flush_reductions() runs before anything reads them
    # ------------------------------------------------------------------------------------------------------------
    def _defer(self, src, dst, count, cols, ld, src_offset=0):
        self._red_items.append(nnops.reduce_item(src, dst, count, cols, ld, src_offset=src_offset))
        self._red_keep.append(src)            # the partials must outlive the launch

    _WG_MAX = 16                                          # problems per grouped launch (csrc/kvq_gemm2.hip MAX_PROBLEMS)

    @staticmethod
    def _wg_tiles(items, bm, bn):
        return sum(-(-p.M // bm) * -(-p.N // bn) for p in items)

    def _flush_wgrads(self, force=True):
        """The weight gradients queued so far as grouped launches, written straight into the flat bf16 gradient buffer, each
        tile contracting over all tokens (no split-K).  The kernel's rate is set by bytes staged per flop (the L2 -> LDS fill
        is the limit), so the 256 x 256 tile is the fast one -- but one BERT layer is only 108-126 of them.  The
        queue is therefore held (force=False) until TWO layers' worth is there: 216-252 tiles = one round of the 256 CUs.
        Returns True when nothing is left queued."""
        items = self._wg_items
        if not items:
            return True
        t256 = self._wg_tiles(items, 256, 256)
        if not force and self._wg_pair and self.wg_stream is None and len(items) <= self._WG_MAX // 2 and 2 * t256 <= 256:
            return False                                   # another layer like this one still fits the same round
        tile = "256x256" if self._wg_tiles(items, 128, 256) > 256 and t256 <= 320 else "128x256"
        if self.wg_stream is not None:
            # KVQ_WG_STREAM=1: the layer's weight gradients run on a side stream while the main stream goes on with the next
            # layer's backward chain (nothing there reads them); joined at the end of backward / before an all-reduce
            self.wg_stream.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self.wg_stream):
                for i in range(0, len(items), self._WG_MAX):
                    nnops.gemm_grouped(items[i:i + self._WG_MAX], "tn", tile)
            self._wg_pending = True
            self._wg_keep_step += self._wg_keep            # operands stay alive until the join
        else:
            for i in range(0, len(items), self._WG_MAX):
                nnops.gemm_grouped(items[i:i + self._WG_MAX], "tn", tile)
        self._wg_items, self._wg_keep = [], []
        return True

    def _join_wgrads(self):
        if self._wg_pending:
            torch.cuda.current_stream(self.dev).wait_stream(self.wg_stream)
            self._wg_pending = False
        self._wg_keep_step = []

    def _flush_reductions(self, force=True):
        """Launch the queued batched reductions and (see _flush_wgrads) the queued weight gradients; True if none stay queued."""
        done = self._flush_wgrads(force)
        if not done and self._red_pair:
            return False                       # ... and the small sums wait with them: one launch per two layers as well
        if self._red_items:
            nnops.reduce_batch(self._red_items)       # (round 5: on a side stream, joined with the weight gradients, the step was
                                                      #  0.5 ms SLOWER -- profiles/r05_gemm_ceiling.md)
        self._red_items, self._red_keep = [], []
        return done

    def _defer_colsum(self, x, dst, cols=None):
        part = nnops.colsum_partial(x, cols)
        self._defer(part, dst, part.shape[0], part.shape[1], part.shape[1])

    def _ln_bwd(self, g_out, pre, mean, rstd, gamma, p_drop, seed, site, g_gamma, g_beta, g_bias_prev=None, need_g_resid=True):
        """LayerNorm backward; the gamma / beta / previous-dense-bias gradients are queued as deferred reductions."""
        g_y, g_resid, part = nnops.ln_bwd_partial(g_out, pre, mean, rstd, gamma, p_drop, seed, site, need_g_resid=need_g_resid,
                                                  want_dbias=g_bias_prev is not None)
        H = g_out.shape[1]
        runs = []                                            # [column offset in part, dst, cols]: adjacent targets merge
        for off, dst in ((0, g_bias_prev), (H, g_gamma), (2 * H, g_beta)):
            if dst is None:
                continue
            if runs and runs[-1][0] + runs[-1][2] == off and \
                    runs[-1][1].data_ptr() + runs[-1][2] * dst.element_size() == dst.data_ptr() and runs[-1][1].dtype == dst.dtype:
                runs[-1][2] += H
            else:
                runs.append([off, dst, H])
        for off, dst, cols in runs:
            self._defer(part, dst, part.shape[0], cols, 3 * H, src_offset=off)
        return g_y, g_resid

    def _wgrad(self, gy, x, out):
        """gW = gy^T x.  Nothing on the way to the next layer's gradient needs it; with KVQ_WG_STREAM=1 it runs on a side stream
        next to the input-gradient GEMM (a fork / join inside the captured hipGraph, joined by the layer's batched reduction).
        Off by default: it measured 5 % slower than one stream."""
        return self._wgrad_on_current_stream(gy, x, out)

    def _wgrad_on_current_stream(self, gy, x, out):
        """bf16: a launch of its own when the output alone fills half the CUs with 256 x 256 tiles (the LM head / all-layer
        cross-K/V weight gradients), else queued for the layer's grouped launch; both contract over ALL tokens per tile (no split-K).
        A token count that is not a multiple of 64 (the MFMA kernel's k-tile: 100 sentences x 12 tokens, the last batch of an
        epoch) is zero-padded first -- zero rows of gy / x add nothing -- so that the step stays on the MFMA kernels; only what
        cannot be padded in 16-byte pieces goes to the any-shape kernel."""
        Ntok, M = gy.shape
        N = x.shape[1]
        if self._own_wgrad and Ntok % 64 != 0 and Ntok > 64:
            padded = nnops.tn_operands_k64(gy, x)
            if padded is not None:
                gy, x = padded
        if self._own_wgrad and nnops.gemm_mfma_ok(gy, x, out, "tn"):
            if -(-M // 256) * -(-N // 256) >= 128:
                self._gemm(gy, x, "tn", out=out)
            else:
                self._wg_items.append(nnops.gemm_problem(gy, x, out, "tn"))
                self._wg_keep += [gy, x]                       # alive until the grouped launch
            return
        self._gemm(gy, x, "tn", out=out)

    def _linear_bwd(self, gy, x, wnames, bnames, need_gx=True, gx_accum=None, bias_done=False):
        """Weight / bias gradients straight into the flat gradient buffer; returns gx (or accumulates into gx_accum)."""
        fl = self.flat
        if len(wnames) == 1:
            W, gW, gb = fl.w(wnames[0]), fl.g(wnames[0]), fl.g(bnames[0])
        else:
            W, gW, gb = fl.fused(wnames, fl.shadow), fl.fused(wnames, fl.grad), fl.fused(bnames, fl.grad)
        if fl.trainable[wnames[0]]:
            self._wgrad(gy, x, gW)
        if fl.trainable[bnames[0]] and not bias_done:     # bias_done: the LayerNorm backward kernel already produced it
            self._defer_colsum(gy, gb)
        if gx_accum is not None:
            return self._gemm(gy, W, "nn", out=gx_accum, accumulate=True)
        return self._gemm(gy, W, "nn") if need_gx else None

    def _dgrad(self, gy, W):
        """gx = gy . W for a gradient that is not paired with a weight / bias gradient here (LM head, batched cross-K/V)."""
        return self._gemm(gy, W, "nn")

    # ------------------------------------------------------------------------------------------------------------
    # blocks: forward returns (output, saved); backward consumes saved
    # ------------------------------------------------------------------------------------------------------------
    def _emb_fwd(self, prefix, cfg, ids, training, word_rows=None):
        """BertEmbeddings (modeling_bert.py:53-110) as ONE kernel: gather + position / type rows + LayerNorm + dropout."""
        fl = self.flat
        B, S = ids.shape
        word = fl.w(prefix + "word", rows=word_rows) if word_rows else fl.w(prefix + "word")
        p = cfg.hidden_dropout_prob if training else 0.0
        keep = (p, self._site()) if p > 0 else None               # the mask is regenerated in backward from (seed, site)
        out, pre, mean, rstd = nnops.embed_ln_fwd(ids.reshape(-1), word, fl.w(prefix + "pos"), fl.w(prefix + "type")[0],
                                                  fl.w32(prefix + "ln.w"), fl.w32(prefix + "ln.b"), cfg.layer_norm_eps, S,
                                                  p, self._step_seed, keep[1] if keep else 0)
        return out, (ids, pre, mean, rstd, keep)

    def _emb_bwd(self, prefix, g, saved, tied_accumulate=False):
        fl = self.flat
        ids, pre, mean, rstd, keep = saved
        B, S = ids.shape
        tr = fl.trainable
        zeroed = False                                        # position / type tables already zeroed together with the word table
        # dropout(LayerNorm(x)): the mask applies to the incoming gradient, inside the LayerNorm backward kernel
        g_y, part = nnops.ln_dropout_bwd_partial(g, pre, mean, rstd, fl.w32(prefix + "ln.w"), keep[0] if keep else 0.0,
                                                 self._step_seed, keep[1] if keep else 0)
        g_pt = g_y                                            # the sum's gradient reaches the word, position and type rows alike
        Hn = g.shape[1]
        gg, gb = (fl.g(prefix + "ln.w") if tr[prefix + "ln.w"] else None), (fl.g(prefix + "ln.b") if tr[prefix + "ln.b"] else None)
        if gg is not None and gb is not None and gg.data_ptr() + Hn * gg.element_size() == gb.data_ptr():
            self._defer(part, fl.fused([prefix + "ln.w", prefix + "ln.b"], fl.grad), part.shape[0], 2 * Hn, 3 * Hn, src_offset=Hn)
        else:
            if gg is not None:
                self._defer(part, gg, part.shape[0], Hn, 3 * Hn, src_offset=Hn)
            if gb is not None:
                self._defer(part, gb, part.shape[0], Hn, 3 * Hn, src_offset=2 * Hn)
        if tr[prefix + "word"]:
            o, n, shape = fl.seg[prefix + "word"]
            gw = fl.grad[o:o + n].view(shape)
            if g_y.shape[1] % 4 == 0 and g_y.shape[1] <= 1024:
                # deterministic segmented sum over the tokens sorted by id: the order is a property of the BATCH, prepared once
                # where the batch is built (prepare_batch: dsentences.token_cache / the trainer), not inside the replayed step
                srt = self._sorted[prefix]
                if isinstance(srt, str):                  # the decoder reads the encoder's ids: one sort serves both tables
                    if self._sorted[srt] is None:
                        self._sorted[srt] = self.prepare_batch(ids)
                    srt = self._sorted[srt]
                elif srt is None:
                    srt = self._sorted[prefix] = self.prepare_batch(ids)
                # the tables are zeroed by ONE launch (word unless it accumulates into the LM-head gradient, position, token type)
                zero = [None if tied_accumulate else gw, fl.g(prefix + "pos") if tr[prefix + "pos"] else None,
                        fl.g(prefix + "type") if tr[prefix + "type"] else None]
                if all(t is None or (t.data_ptr() % 16 == 0 and (t.numel() * t.element_size()) % 16 == 0) for t in zero):
                    nnops.zero_ranges(zero)
                    zeroed = True
                elif not tied_accumulate:
                    gw.zero_()
                nnops.embed_grad(g_y, srt[1], srt[0], gw, accumulate=tied_accumulate)
            else:
                acc = torch.zeros(shape, dtype=torch.float32, device=self.dev)
                acc.index_add_(0, ids.reshape(-1), g_y.float())
                if self._pad_idx is not None:
                    acc[self._pad_idx] = 0
                if tied_accumulate:
                    gw.add_(acc.to(gw.dtype))          # LM-head weight gradient is already in there
                else:
                    gw.copy_(acc)
        # position rows get the sum over sentences, token-type row 0 the sum over all tokens: two column sums of g_pt seen as
        # [B, S*H] and [N, H], finished by the layer's batched reduction (rows that are not looked up keep a zero gradient)
        Hh = g_pt.shape[1]
        if tr[prefix + "pos"]:
            gp = fl.g(prefix + "pos")
            if not zeroed:
                gp.zero_()
            part = nnops.colsum_partial(g_pt.view(B, S * Hh))
            self._defer(part, gp[:S], part.shape[0], S * Hh, S * Hh)
        if tr[prefix + "type"]:
            gt = fl.g(prefix + "type")
            if not zeroed:
                gt.zero_()
            part = nnops.colsum_partial(g_pt)
            self._defer(part, gt[0], part.shape[0], Hh, Hh)

    def _attn_block_fwd(self, pre, x, kv_src, mask, causal, cfg, training, B, Sq, Sk, kv_pre=None, next_key=None):
        """self-attention (kv_src is None) or cross-attention on kv_src (kv_pre: its already projected [N, 2H] keys | values,
        a column slice of the all-layer projection); returns LN(dropout(dense(ctx)) + x)."""
        fl, H, nh = self.flat, self.H, self.nh
        p_attn = cfg.attention_probs_dropout_prob if training else 0.0
        p_hid = cfg.hidden_dropout_prob if training else 0.0
        site_a, site_o = self._site(), self._site()
        if kv_src is None:
            qkv = self._linear(x, None, None, fused=([pre + "q.w", pre + "k.w", pre + "v.w"], [pre + "q.b", pre + "k.b", pre + "v.b"]))
            q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
            kvbuf = None
        else:
            qkv = self._linear(x, pre + "q.w", pre + "q.b")
            kvbuf = kv_pre if kv_pre is not None else \
                self._linear(kv_src, None, None, fused=([pre + "k.w", pre + "v.w"], [pre + "k.b", pre + "v.b"]))
            q, k, v = qkv, kvbuf[:, :H], kvbuf[:, H:]
        if self._emits_fp8_for(pre + "o.w") and q.dtype == torch.bfloat16 and nnops.attn_fwd_fp8_ok(Sq, Sk):
            ctx, lse, ctx8 = nnops.attn_fwd_fp8(q, k, v, mask, B, nh, Sq, Sk, causal, p_attn, self._step_seed, site_a,
                                                self._a8_state_of(pre + "o.w"))
            self._x8[(ctx.data_ptr(), pre + "o.w")] = ctx8
        else:
            ctx, lse = nnops.attn_fwd(q, k, v, mask, B, nh, Sq, Sk, causal, p_attn, self._step_seed, site_a)
        out, lnpre, mean, rstd = self._dense_residual_ln(ctx, pre + "o.w", pre + "o.b", x, pre + "ln.w", pre + "ln.b", cfg.layer_norm_eps,
                                                         p_hid, site_o, next_key=next_key)
        # (lse: only the kernels above 32 tokens work from the saved log-sum-exp)
        return out, (x, kv_src, qkv, kvbuf, ctx, lnpre, mean, rstd, mask, causal, p_attn, p_hid, site_a, site_o, B, Sq, Sk,
                     lse if max(Sq, Sk) > 32 else None)

    def _attn_block_bwd(self, pre, g_out, saved, g_kv_src=None, g_kv_out=None, pb_kv_out=None):
        """returns g_x; for cross-attention accumulates the gradient of kv_src into g_kv_src -- or, in the batched layout, only
        leaves g_kv (and its bias partial rows) in the given column slices for the all-layer GEMMs after the decoder loop."""
        fl, H, nh = self.flat, self.H, self.nh
        x, kv_src, qkv, kvbuf, ctx, lnpre, mean, rstd, mask, causal, p_attn, p_hid, site_a, site_o, B, Sq, Sk, lse = saved
        long_kw = dict(ctx=ctx, lse=lse) if lse is not None else {}
        tr = fl.trainable
        g_ao, g_x = self._ln_bwd(g_out, lnpre, mean, rstd, fl.w32(pre + "ln.w"), p_hid, self._step_seed, site_o,
                                 g_gamma=fl.g(pre + "ln.w") if tr[pre + "ln.w"] else None,
                                 g_beta=fl.g(pre + "ln.b") if tr[pre + "ln.b"] else None,
                                 g_bias_prev=fl.g(pre + "o.b") if tr[pre + "o.b"] else None)
        g_ctx = self._linear_bwd(g_ao, ctx, [pre + "o.w"], [pre + "o.b"], bias_done=True)
        # the attention backward kernel also emits the per-sentence column sums of g_q / g_k / g_v: the q/k/v bias gradients
        # are then one deferred reduction over B rows instead of a second pass over the [N, 3H] gradient
        g_qkv = torch.empty_like(qkv)
        want_b = tr[pre + "q.b"]
        if kv_src is None:
            q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
            pb = torch.empty((B, 3 * H), dtype=torch.float32, device=self.dev) if want_b else None
            nnops.attn_bwd(q, k, v, mask, g_ctx, B, nh, Sq, Sk, causal, p_attn, self._step_seed, site_a,
                           g_qkv[:, :H], g_qkv[:, H:2 * H], g_qkv[:, 2 * H:],
                           *((pb[:, :H], pb[:, H:2 * H], pb[:, 2 * H:]) if want_b else ()), **long_kw)
            if want_b:
                self._defer(pb, fl.fused([pre + "q.b", pre + "k.b", pre + "v.b"], fl.grad), B, 3 * H, 3 * H)
            self._linear_bwd(g_qkv, x, [pre + "q.w", pre + "k.w", pre + "v.w"], [pre + "q.b", pre + "k.b", pre + "v.b"], gx_accum=g_x,
                             bias_done=want_b)
        else:
            batched = g_kv_out is not None
            g_kv = g_kv_out if batched else torch.empty_like(kvbuf)
            pbq = torch.empty((B, H), dtype=torch.float32, device=self.dev) if want_b else None
            want_bkv = tr[pre + "k.b"]
            pbkv = (pb_kv_out if batched else torch.empty((B, 2 * H), dtype=torch.float32, device=self.dev)) if want_bkv else None
            nnops.attn_bwd(qkv, kvbuf[:, :H], kvbuf[:, H:], mask, g_ctx, B, nh, Sq, Sk, causal, p_attn, self._step_seed, site_a,
                           g_qkv, g_kv[:, :H], g_kv[:, H:], pbq, pbkv[:, :H] if want_bkv else None, pbkv[:, H:] if want_bkv else None,
                           **long_kw)
            if want_b:
                self._defer(pbq, fl.g(pre + "q.b"), B, H, H)
            self._linear_bwd(g_qkv, x, [pre + "q.w"], [pre + "q.b"], gx_accum=g_x, bias_done=want_b)
            if not batched:
                if want_bkv:
                    self._defer(pbkv, fl.fused([pre + "k.b", pre + "v.b"], fl.grad), B, 2 * H, 2 * H)
                self._linear_bwd(g_kv, kv_src, [pre + "k.w", pre + "v.w"], [pre + "k.b", pre + "v.b"], gx_accum=g_kv_src, bias_done=want_bkv)
        return g_x

    def _ffn_fwd(self, pre, x, cfg, training, next_key=None):
        fl = self.flat
        p_hid = cfg.hidden_dropout_prob if training else 0.0
        site = self._site()
        h, a = self._linear_gelu(x, pre + "f1.w", pre + "f1.b", next_key=pre + "f2.w")
        out, lnpre, mean, rstd = self._dense_residual_ln(a, pre + "f2.w", pre + "f2.b", x, pre + "ln2.w", pre + "ln2.b", cfg.layer_norm_eps,
                                                         p_hid, site, next_key=next_key)
        return out, (x, h, a, lnpre, mean, rstd, p_hid, site)

    def _ffn_bwd(self, pre, g_out, saved):
        fl = self.flat
        x, h, a, lnpre, mean, rstd, p_hid, site = saved
        tr = fl.trainable
        g_f, g_x = self._ln_bwd(g_out, lnpre, mean, rstd, fl.w32(pre + "ln2.w"), p_hid, self._step_seed, site,
                                g_gamma=fl.g(pre + "ln2.w") if tr[pre + "ln2.w"] else None,
                                g_beta=fl.g(pre + "ln2.b") if tr[pre + "ln2.b"] else None,
                                g_bias_prev=fl.g(pre + "f2.b") if tr[pre + "f2.b"] else None)
        W2 = fl.w(pre + "f2.w")
        tile = self._epilogue_tile(g_f.shape[0], W2.shape[1], W2.shape[0])
        if tile is not None and g_f.is_contiguous() and h.is_contiguous() and nnops.gemm_mfma_ok(g_f, W2, None, "nn"):
            # input gradient of f2, the activation's derivative and the f1-bias partial sums in ONE kernel
            self._linear_bwd(g_f, a, [pre + "f2.w"], [pre + "f2.b"], need_gx=False, bias_done=True)      # queues the f2 weight gradient
            g_h, pb = nnops.gemm_dgelu(g_f, W2, h, tile=tile)
            if tr[pre + "f1.b"]:
                self._defer(pb, fl.g(pre + "f1.b"), pb.shape[0], pb.shape[1], pb.shape[1])
            self._linear_bwd(g_h, x, [pre + "f1.w"], [pre + "f1.b"], gx_accum=g_x, bias_done=True)
            return g_x
        g_a = self._linear_bwd(g_f, a, [pre + "f2.w"], [pre + "f2.b"], bias_done=True)
        if tr[pre + "f1.b"] and h.is_contiguous() and h.shape[1] % 8 == 0:
            g_h, pb = nnops.gelu_bwd_bias(h, g_a, out=g_a)           # bias gradient partials come out of the same pass
            self._defer(pb, fl.g(pre + "f1.b"), pb.shape[0], pb.shape[1], pb.shape[1])
            self._linear_bwd(g_h, x, [pre + "f1.w"], [pre + "f1.b"], gx_accum=g_x, bias_done=True)
        else:
            g_h = nnops.gelu_bwd(h, g_a, out=g_a)
            self._linear_bwd(g_h, x, [pre + "f1.w"], [pre + "f1.b"], gx_accum=g_x)
        return g_x

    # ------------------------------------------------------------------------------------------------------------
    # gradient exchange (multi-GPU): all-reduce finished tail chunks of the flat gradient buffer while backward runs
    # ------------------------------------------------------------------------------------------------------------
    def _grads_done_down_to(self, name, partial=False):
        """Every gradient located at or after segment `name` is final (once the queued reductions have been launched).
        `partial`: also send the incomplete chunk above `name` (used before the last, late part of backward)."""
        lo = self.flat.seg[name][0]
        if not self._flush_reductions(force=partial):
            lo = self._wg_done_lo                          # weight gradients of this layer still queued: final only above there
        else:
            self._wg_done_lo = lo
        if not self._dp:
            if self._fuse_opt and lo < self._adam_hi:
                self._adam_early(lo, self._adam_hi)
                self._adam_hi = lo
            return
        self._join_wgrads()
        ranges = []
        while self._pending_hi - self.chunk >= lo:
            ranges.append((self._pending_hi - self.chunk, self._pending_hi))
            self._pending_hi -= self.chunk
        if partial and self._pending_hi > lo:
            ranges.append((lo, self._pending_hi))
            self._pending_hi = lo
        if ranges:                               # ONE eager interlude for all chunks that became final here (no empty graphs between)
            grad = self.flat.grad
            self._eager(lambda: [self._all_reduce_avg(grad[a:b]) for a, b in ranges])

    def _all_reduce_avg(self, t, late=False):
        works = self._works_late if late else self._works
        self.comm_stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.comm_stream):
            if self._avg_in_comm:
                works.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
            else:
                works.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), t))

    def _eager(self, fn):
        """Run fn now; while a step is being captured it also becomes an eager launch between two graphs of the replay."""
        self._adam_join()                  # a captured graph ends here: no side-stream work may be left open in it
        if self._cap is not None:
            self._cap.interlude(fn)
        else:
            fn()

    def _settle(self, works):
        """On the side stream: wait for these collectives (and finish the average where the backend only sums)."""
        with torch.cuda.stream(self.comm_stream):     # work.wait() orders the CURRENT stream behind the collective's result:
            for w, t in works:                        # the side stream must be the one that waits before it divides
                w.wait()
                if t is not None:
                    t.div_(self.world)
        works.clear()

    def _exchange_head(self, cut):
        """End of backward.  Everything above `cut` was sent while backward ran: settle it and let the optimiser start on
        that part; the head of the buffer [0, cut) -- the embedding gradients, final only now -- and the codebook gradient
        go out behind it and are waited for by _exchange_tail(), i.e. their all-reduce overlaps with Adam on the rest."""
        self._settle(self._works)
        self._ev_early.record(self.comm_stream)
        if cut > 0:
            self._all_reduce_avg(self.flat.grad[:cut], late=True)
        for a in self.aux:
            if a["p"].requires_grad:
                self._all_reduce_avg(a["g"], late=True)
        self._timed_wait(lambda main: main.wait_event(self._ev_early))

    def _exchange_tail(self):
        self._settle(self._works_late)
        self._timed_wait(lambda main: main.wait_stream(self.comm_stream))

    def _timed_wait(self, wait):
        main = torch.cuda.current_stream(self.dev)
        if not self._time_comm:
            wait(main)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        wait(main)
        e1.record(main)
        self._comm_ev.append((e0, e1))

    def reset_comm_timing(self, enable=True):
        """Start (or stop) recording event pairs around the compute stream's waits for the gradient all-reduces."""
        self._comm_ev, self._time_comm = [], bool(enable)

    def exposed_comm_ms(self):
        """Total time the compute stream spent blocked on gradient all-reduces since reset_comm_timing() (the two wait points of
        a step: before Adam on the tail of the buffer, before Adam on its head).  Call after a device synchronisation."""
        tot = 0.0
        for e0, e1 in self._comm_ev:
            tot += e0.elapsed_time(e1)
        return tot

    # ------------------------------------------------------------------------------------------------------------
    # one training step
    # ------------------------------------------------------------------------------------------------------------
    def forward_backward(self, input_ids, attention_mask, training=True, compute_grads=True, dec_ids=None, dec_mask=None,
                         want_logits=False, quantizer_training=None, fuse_optimizer=False, stop_after_quantizer=False,
                         defer_backward=False, target_ids=None):
        """Forward (+ backward when compute_grads).  Returns dict(loss_recon, loss_vq, perplexity, acc, acc_per_sentence,
        recon_ids, indices [, logits]).  dec_ids / dec_mask: the decoder's own input (Bagon.forward takes one, and the Bagon step
        tokenises and perturbs the two sides separately, models/bagon/Trainer.py:78-96; default = the encoder's).  target_ids:
        what the loss and the accuracy score the logits against (default = the decoder's input, as models/bagon/Trainer.py:103-110
        and models/shelgon3/Trainer.py:94-101 do).
        fuse_optimizer (train_step only; optimizer_step() MUST follow): parameters are updated while backward still runs.
        defer_backward (with compute_grads=False): the forward's activations stay alive and out["_resume"] can be handed to
        backward_from() once the gradient of the returned logits is known (kvq.engine.engine_autograd_forward)."""
        self._fuse_opt = bool(fuse_optimizer) and compute_grads and self._early_adam and not self._dp
        self._adam_hi, self._adam_forked = self.flat.n, False
        self._stop_after_quantizer = bool(stop_after_quantizer) and not compute_grads
        S = max(input_ids.shape[1], dec_ids.shape[1] if dec_ids is not None else 0)
        if S > (128 if self.dtype == torch.bfloat16 else 32):
            raise KvqError(f"TrainEngine: sequence length {S} above the attention kernels' limit (32 tokens in f32, 128 in bf16 "
                           f"through the blocked kernels; use the autograd path)")
        if dec_ids is not None and (dec_mask is None or dec_ids.shape[0] != input_ids.shape[0] or dec_mask.shape != dec_ids.shape):
            raise KvqError("TrainEngine: dec_ids needs a dec_mask of its shape and the encoder's batch size")
        if target_ids is not None and target_ids.shape != (dec_ids if dec_ids is not None else input_ids).shape:
            raise KvqError("TrainEngine: target_ids must have the shape of the decoder's input")
        if defer_backward and (compute_grads or self._cap is not None or self._dp):
            raise KvqError("TrainEngine: defer_backward is a single-process, eager, forward-first call")
        self._site_ctr = 0
        self._red_items, self._red_keep = [], []
        self._wg_items, self._wg_keep = [], []
        pre = getattr(self, "_prepared", None) or {}            # this batch's (sorted ids, order) per side: _normalise_prepared()
        self._sorted = {"enc.emb.": pre.get("enc"), "dec.emb.": pre.get("dec") if dec_ids is not None else "enc.emb."}
        nnops.set_seed_offset(self._state)        # dropout seeds of this engine's launches = _step_seed + device step count
        try:
            self._q_training = training if quantizer_training is None else bool(quantizer_training)
            with torch.no_grad():            # the schedule IS the backward pass: no autograd graph over the few torch ops in it
                if self.fp8:
                    self._x8.clear()
                out = self._forward_backward(input_ids, attention_mask, training, compute_grads, dec_ids, dec_mask, want_logits,
                                             defer=bool(defer_backward), target_ids=target_ids)
                if defer_backward:
                    out["_resume"] = dict(gen=out.pop("_gen"), sorted_ids=self._sorted, step=self._step_host,
                                          versions=self._versions())
                if self.fp8 and compute_grads:
                    # the next TRAINING step's activation scales from this step's amax (4x headroom).  A forward-only call
                    # (evaluation, Shelgon.forward) uses the scales as they are and leaves them alone: an eval batch must not
                    # move the scales of the training step that follows it
                    check(lib().kvq_fp8_update_scales(self._a8_state.data_ptr(), self._a8_sites, 4.0, stream_ptr()), "kvq_fp8_update_scales")
                elif self.fp8:
                    # (the producers note amax by atomic maxima: what an evaluation batch left in the partial slots must go)
                    check(lib().kvq_fp8_update_scales(self._a8_state.data_ptr(), self._a8_sites, 0.0, stream_ptr()), "kvq_fp8_update_scales")
                return out
        finally:
            nnops.set_seed_offset(None)

    def forward_logits(self, enc_ids, enc_mask, dec_ids=None, dec_mask=None, training=False, quantizer_training=None):
        """Forward only, on the engine's kernels, with the [B, S, V] logits in the result: what Bagon.forward / Shelgon.forward
        return (models/bagon/Bagon.py:40-55, models/shelgon3/Shelgon.py:50-73)."""
        self.refresh_if_params_changed()
        if getattr(self, "_epack", None) is not None:
            self._E_version = None             # an evaluation forward never trusts a cached pack (a 3-us kernel; see _codebook_stamp)
        return self.forward_backward(enc_ids, enc_mask, training=training, compute_grads=False, dec_ids=dec_ids, dec_mask=dec_mask,
                                     want_logits=True, quantizer_training=quantizer_training)

    def code_indices(self, enc_ids, enc_mask, quantizer_training=False):
        """Encoder + quantiser only: dict(indices, perplexity, loss_vq_raw).  What a consumer of the codes needs -- the reference's
        analysis (analyses/unsupervised_vq_disentanglement/unsupervised_vq_disentanglement.py:164) runs the whole model.forward
        and drops everything but min_encoding_indices; the decoder and the LM head are 60 % of a forward."""
        if not self.has_vq:
            raise KvqError("TrainEngine.code_indices: the model has no quantiser")
        self.refresh_if_params_changed()
        if getattr(self, "_epack", None) is not None:
            self._E_version = None
        return self.forward_backward(enc_ids, enc_mask, training=False, compute_grads=False, quantizer_training=quantizer_training,
                                     stop_after_quantizer=True)

    def _forward_backward(self, input_ids, attention_mask, training, compute_grads, dec_ids=None, dec_mask=None, want_logits=False,
                          defer=False, target_ids=None):
        gen = self._fb_gen(input_ids, attention_mask, training, compute_grads, dec_ids, dec_mask, want_logits, defer, target_ids)
        try:
            out = next(gen)                 # only a deferred call yields: forward done, the generator holds the activations
        except StopIteration as done:
            return done.value
        out["_gen"] = gen
        return out

    def backward_from(self, resume, g_logits, g_loss_vq=None):
        """The backward half of a forward_backward(..., defer_backward=True) call, seeded with d L / d logits ([B, S, V], the
        logits' dtype) and d L / d loss_vq_raw (scalar tensor or None = 0) instead of this engine's own loss: gradients land in
        the flat buffer exactly as in a training step (grads_by_parameter() hands them out)."""
        if resume.get("done"):
            raise KvqError("TrainEngine.backward_from: this forward's activations were already consumed")
        if resume["step"] != self._step_host or resume["versions"] != self._versions():
            raise KvqError("TrainEngine.backward_from: parameters changed since the forward (an optimiser step or an in-place write)")
        resume["done"] = True
        self._red_items, self._red_keep = [], []
        self._wg_items, self._wg_keep = [], []
        self._sorted = resume["sorted_ids"]
        self._fuse_opt = False
        self._g_vq_ext = (g_loss_vq.detach().to(torch.float32).reshape(()) if g_loss_vq is not None
                          else torch.zeros((), dtype=torch.float32, device=self.dev)) if self.has_vq else None
        nnops.set_seed_offset(self._state)          # the step count has not moved: backward regenerates the forward's dropout masks
        try:
            with torch.no_grad():
                try:
                    resume["gen"].send(g_logits.detach())
                except StopIteration:
                    pass
                else:
                    raise KvqError("TrainEngine.backward_from: the schedule did not finish")
        finally:
            nnops.set_seed_offset(None)
            self._g_vq_ext = None

    def grads_by_parameter(self):
        """{nn.Parameter: float32 gradient} of the last backward, for every trainable parameter of the model (what autograd would
        have put into .grad): views of the flat bf16 / f32 gradient buffer converted to the parameter's dtype."""
        out = {}
        for name, p in self.param_of.items():
            if p.requires_grad:
                g = self.flat.g(name)
                out[p] = torch.empty_like(p).copy_(g.reshape(p.shape))           # a copy: the flat buffer is rewritten by the next backward
        for a in self.aux:
            if a["p"].requires_grad:
                out[a["p"]] = torch.empty_like(a["p"]).copy_(a["g"].reshape(a["p"].shape))
        return out

    def _fb_gen(self, input_ids, attention_mask, training, compute_grads, dec_ids, dec_mask, want_logits, defer, target_ids=None):
        m = self.model
        fl, H = self.flat, self.H
        B, S = input_ids.shape
        N = B * S
        mask = attention_mask.contiguous()
        ecfg, dcfg = self.ecfg, self.dcfg
        d_ids = input_ids if dec_ids is None else dec_ids
        d_mask = mask if dec_mask is None else dec_mask.contiguous()
        Sd = d_ids.shape[1]
        Nd = B * Sd
        if d_ids.shape[0] != B:
            raise KvqError("TrainEngine: encoder and decoder batches differ")

        # ---------------- forward ----------------
        x, emb_saved = self._emb_fwd("enc.emb.", ecfg, input_ids, training)
        enc_saved = []
        for i in range(self.n_enc_layers):
            x, sa = self._attn_block_fwd(f"enc.{i}.sa.", x, None, mask, False, ecfg, training, B, S, S, next_key=f"enc.{i}.f1.w")
            x, ff = self._ffn_fwd(f"enc.{i}.", x, ecfg, training, next_key=f"enc.{i + 1}.sa.q.w" if i + 1 < self.n_enc_layers else None)
            enc_saved.append((sa, ff))
        z = x
        gum_saved = None
        if self.vq_kind in ("VectorQuantizer", "MultiVectorQuantizer"):
            cap = self._cap
            if cap is not None:                     # graph capture: the quantiser stays an eager launch between two graphs
                z_q, idx, vq_out = cap.z_q, cap.idx, cap.vq_out
            else:
                z_q = torch.empty_like(z)
                idx = torch.empty(N * self.G, dtype=torch.int64, device=self.dev)
                vq_out = torch.empty(2, dtype=torch.float32, device=self.dev)
            self._eager(lambda: self._vq_forward(z, z_q, idx, vq_out))
            loss_vq, perplexity = vq_out[0], vq_out[1]
            enc_out = z_q
            indices = idx.view(self.G, N).t().reshape(B, S, self.G)          # [B, S, 1] for the reference's single codebook
        elif self.vq_kind == "GumbelQuantizer":
            enc_out, loss_vq, perplexity, ind, gum_saved = self._gumbel_forward(z, self._q_training)
            idx, indices = ind, ind.view(B, S)                               # GumbelQuantizer.py:76 returns [B, S]
        else:
            idx, loss_vq, perplexity, enc_out, indices = None, None, None, z, None
        if getattr(self, "_stop_after_quantizer", False):
            return dict(indices=indices, perplexity=perplexity, loss_vq_raw=loss_vq)

        y, demb_saved = self._emb_fwd("dec.emb.", dcfg, d_ids, training, word_rows=None)
        dec_saved = []
        kv_all = None
        if self._cakv_batched:             # keys | values of every decoder layer's cross-attention in one [N, L*2H] GEMM
            kv_all = self._linear(enc_out, None, None, Wb=(fl.fused(self._cakv_w, fl.shadow), fl.fused(self._cakv_b, fl.shadow)),
                                  key=self._cakv_w[0])
        for i in range(self.n_dec_layers):
            y, sa = self._attn_block_fwd(f"dec.{i}.sa.", y, None, d_mask, True, dcfg, training, B, Sd, Sd, next_key=f"dec.{i}.ca.q.w")
            y, ca = self._attn_block_fwd(f"dec.{i}.ca.", y, enc_out, None, False, dcfg, training, B, Sd, S,
                                         kv_pre=kv_all[:, 2 * H * i: 2 * H * (i + 1)] if kv_all is not None else None,
                                         next_key=f"dec.{i}.f1.w")
            y, ff = self._ffn_fwd(f"dec.{i}.", y, dcfg, training, next_key=f"dec.{i + 1}.sa.q.w" if i + 1 < self.n_dec_layers else "head.t.w")
            dec_saved.append((sa, ca, ff))
        t, ta = self._linear_gelu(y, "head.t.w", "head.t.b")
        if self.fp8 and "dec.emb.word" in self._w8_index and ta.dtype == torch.bfloat16 and os.environ.get("KVQ_FP8_HEAD_EMIT", "1") != "0":
            # the LM head runs on fp8 in every fp8 scope: the LayerNorm in front of it writes the fp8 copy of its output itself
            hN, hpre, hmean, hrstd, hN8 = nnops.ln_fwd_fp8(ta, None, fl.w32("head.ln.w"), fl.w32("head.ln.b"), dcfg.layer_norm_eps, 0.0, 0, 0,
                                                           self._a8_state_of("dec.emb.word"))
            self._x8[(hN.data_ptr(), "dec.emb.word")] = hN8
        else:
            hN, hpre, hmean, hrstd = nnops.ln_fwd(ta, None, fl.w32("head.ln.w"), fl.w32("head.ln.b"), dcfg.layer_norm_eps)
        Wv = fl.w("dec.emb.word", rows=self.Vp)                              # [Vp,H], rows >= V are zero
        bv = fl.shadow[fl.seg["head.bias"][0]: fl.seg["head.bias"][0] + self.Vp]
        tgt = (d_ids if target_ids is None else target_ids).reshape(-1)
        # LM head; with the own kernel its epilogue also leaves the loss' forward statistics per (row, 256-column tile): the
        # [N, Vp] logits are then read by the loss only once more, in backward (opt-in, KVQ_OWN_LMCE=1; default: library GEMM + kvq_ce_forward)
        lm_stats = None
        if self._own_lmce and self._own_fwd and not self.fp8 and self.dtype == torch.bfloat16 and hN.shape[0] >= 2048 \
                and hN.is_contiguous() and self.Vp % 8 == 0:
            logits, lm_stats = nnops.gemm_ce(hN, Wv, bv, self.V)
        else:
            logits = self._linear(hN, None, None, Wb=(Wv, bv), key="dec.emb.word")  # [N,Vp]
        row_loss = torch.empty(Nd, dtype=torch.float32, device=self.dev)
        row_lse = torch.empty(Nd, dtype=torch.float32, device=self.dev)
        pred = torch.empty(Nd, dtype=torch.int64, device=self.dev)
        ce_out = self._cap.scal[0:2] if (self._cap is not None and compute_grads) else torch.empty(2, dtype=torch.float32, device=self.dev)
        if lm_stats is not None:
            nnops.ce_forward_stats(logits, tgt, lm_stats, row_loss, row_lse, pred, ce_out[0:], ce_out[1:])
        else:
            check(lib().kvq_ce_forward(logits.data_ptr(), tgt.data_ptr(), Nd, self.V, self.Vp, self.io, row_loss.data_ptr(),
                                       row_lse.data_ptr(), pred.data_ptr(), ce_out[0:].data_ptr(), ce_out[1:].data_ptr(), stream_ptr()),
                  "kvq_ce_forward")
        acc_sent = None
        if self.sentence_acc:               # seq_acc's second result (common/metrics.py:32-36), read by the Bagon trainer's decode step
            acc_sent = torch.empty(B, dtype=torch.float32, device=self.dev)
            check(lib().kvq_seq_acc(pred.data_ptr(), tgt.data_ptr(), B, Sd, acc_sent.data_ptr(), stream_ptr()), "kvq_seq_acc")
        out = dict(loss_recon=ce_out[0] if self.w_recon == 1.0 else ce_out[0] * self.w_recon,
                   loss_vq=(loss_vq if self.w_vq == 1.0 else loss_vq * self.w_vq) if self.has_vq else None,
                   perplexity=perplexity, acc=ce_out[1], acc_per_sentence=acc_sent, recon_ids=pred.view(B, Sd), indices=indices)
        if want_logits:
            out["logits"] = logits[:, :self.V].reshape(B, Sd, self.V)
            out["loss_vq_raw"] = loss_vq                          # without the trainer's loss weight
        g_ext = None
        if not compute_grads:
            if not defer:
                return out
            g_ext = yield out               # suspended here until backward_from() sends d L / d logits

        # ---------------- backward ----------------
        if self._fuse_opt:      # lr / bias corrections of the step about to be applied; the step COUNT (dropout seed offset) stays
            nnops.step_state_advance(self._state, self.lr, self.gamma, self.milestones, self.betas[0], self.betas[1], phase="prepare")
        tr = fl.trainable
        g_scale = self._g_recon
        if g_ext is not None:
            # the caller's loss: its gradient replaces this engine's (the [N, V] part of a fresh [N, Vp] buffer -- the logits handed
            # out stay as they are)
            g_logits = torch.zeros_like(logits)
            g_logits[:, :self.V].copy_(g_ext.reshape(Nd, self.V))
            if tr["head.bias"]:
                self._defer_colsum(g_logits, fl.g("head.bias", rows=self.Vp))
        elif tr["head.bias"] and self.Vp % 8 == 0:
            # in place: logits := d loss / d logits; the LM-head bias gradient leaves the same pass as partial column sums
            pb = torch.empty((lib().kvq_ce_bwd_partial_rows(Nd), self.Vp), dtype=torch.float32, device=self.dev)
            check(lib().kvq_ce_backward_bias(logits.data_ptr(), tgt.data_ptr(), row_lse.data_ptr(), g_scale.data_ptr(), Nd, self.V,
                                             self.Vp, self.io, logits.data_ptr(), pb.data_ptr(), pb.numel() * 4, stream_ptr()),
                  "kvq_ce_backward_bias")
            self._defer(pb, fl.g("head.bias", rows=self.Vp), pb.shape[0], self.Vp, self.Vp)
            g_logits = logits
        else:
            check(lib().kvq_ce_backward(logits.data_ptr(), tgt.data_ptr(), row_lse.data_ptr(), g_scale.data_ptr(), Nd, self.V, self.Vp,
                                        self.io, logits.data_ptr(), stream_ptr()), "kvq_ce_backward")      # in place
            g_logits = logits
            if tr["head.bias"]:
                self._defer_colsum(g_logits, fl.g("head.bias", rows=self.Vp))
        if tr["dec.emb.word"]:
            self._wgrad(g_logits, hN, fl.g("dec.emb.word", rows=self.Vp))            # [Vp,H] = g_logits^T hN
        g_hN = self._dgrad(g_logits, Wv)
        del logits, g_logits
        g_ta, _ = self._ln_bwd(g_hN, hpre, hmean, hrstd, fl.w32("head.ln.w"), 0.0, 0, 0,
                               g_gamma=fl.g("head.ln.w") if tr["head.ln.w"] else None,
                               g_beta=fl.g("head.ln.b") if tr["head.ln.b"] else None, need_g_resid=False)
        if tr["head.t.b"] and t.is_contiguous() and t.shape[1] % 8 == 0:
            g_t, pb = nnops.gelu_bwd_bias(t, g_ta, out=g_ta)
            self._defer(pb, fl.g("head.t.b"), pb.shape[0], pb.shape[1], pb.shape[1])
            g_y = self._linear_bwd(g_t, y, ["head.t.w"], ["head.t.b"], bias_done=True)
        else:
            g_t = nnops.gelu_bwd(t, g_ta, out=g_ta)
            g_y = self._linear_bwd(g_t, y, ["head.t.w"], ["head.t.b"])
        self._grads_done_down_to("head.t.w")
        g_kv_all = pb_kv_all = None
        if self._cakv_batched:
            g_kv_all = torch.empty_like(kv_all)
            if tr[self._cakv_b[0]]:
                pb_kv_all = torch.empty((B, kv_all.shape[1]), dtype=torch.float32, device=self.dev)
            g_enc = None
        else:
            g_enc = torch.zeros_like(enc_out)
        for i in reversed(range(self.n_dec_layers)):
            sa, ca, ff = dec_saved[i]
            g_y = self._ffn_bwd(f"dec.{i}.", g_y, ff)
            sl = slice(2 * H * i, 2 * H * (i + 1))
            g_y = self._attn_block_bwd(f"dec.{i}.ca.", g_y, ca, g_kv_src=g_enc,
                                       g_kv_out=g_kv_all[:, sl] if g_kv_all is not None else None,
                                       pb_kv_out=pb_kv_all[:, sl] if pb_kv_all is not None else None)
            g_y = self._attn_block_bwd(f"dec.{i}.sa.", g_y, sa)
            self._grads_done_down_to(f"dec.{i}.sa.q.w")
            dec_saved[i] = None
        if self._cakv_batched:             # all layers' key/value projections at once: weight, bias and input gradients
            if tr[self._cakv_w[0]]:
                self._wgrad(g_kv_all, enc_out, fl.fused(self._cakv_w, fl.grad))
            if pb_kv_all is not None:
                self._defer(pb_kv_all, fl.fused(self._cakv_b, fl.grad), B, pb_kv_all.shape[1], pb_kv_all.shape[1])
            g_enc = self._dgrad(g_kv_all, fl.fused(self._cakv_w, fl.shadow))
            del g_kv_all, kv_all
            self._grads_done_down_to(self._cakv_w[0])
        self._emb_bwd("dec.emb.", g_y, demb_saved, tied_accumulate=True)
        self._grads_done_down_to("dec.emb.word")
        if self.vq_kind in ("VectorQuantizer", "MultiVectorQuantizer"):
            g_x = self._vq_backward(z, idx, g_enc)
        elif self.vq_kind == "GumbelQuantizer":
            g_x = self._gumbel_backward(g_enc, gum_saved)
        else:
            g_x = g_enc
        any_enc = any(v for k, v in tr.items() if k.startswith("enc."))
        if any_enc:
            for i in reversed(range(self.n_enc_layers)):
                sa, ff = enc_saved[i]
                g_x = self._ffn_bwd(f"enc.{i}.", g_x, ff)
                g_x = self._attn_block_bwd(f"enc.{i}.sa.", g_x, sa)
                self._grads_done_down_to(f"enc.{i}.sa.q.w", partial=(i == 0))
                enc_saved[i] = None
            self._emb_bwd("enc.emb.", g_x, emb_saved)
        self._flush_reductions()
        self._join_wgrads()
        return out

    def _buf(self, name, shape, dtype):
        """Persistent scratch of the eager (between-graphs) launches: same storage every step."""
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(shape, dtype=dtype, device=self.dev)
        return t

    def _repack_codebook(self):
        if self._epack is not None:
            check(lib().kvq_vq_pack_codebook(self.E.data_ptr(), self.K, self.Dg, self.G, self._epack.data_ptr(), stream_ptr()),
                  "kvq_vq_pack_codebook")
            self._E_version = self._codebook_stamp()

    def _codebook_stamp(self):
        """What tells a codebook write from outside: the tensor version (in-place torch ops, optimiser steps) and the module's
        epoch counter (its EMA kernel writes through a raw pointer).  A `.data` write that bumps neither needs
        sync_from_model(); forward_logits() repacks unconditionally."""
        return (self.E._version, getattr(self.model.vector_quantizer, "codebook_epoch", 0))

    def _vq_fwd_call(self, z, N, D, G, z_q, idx, loss, perp, ws):
        if self._epack is None:
            check(lib().kvq_vq_forward(z.data_ptr(), self.E.data_ptr(), N, self.K, D, G, self.io, self.beta_vq, z_q.data_ptr(),
                                       idx.data_ptr(), loss.data_ptr(), perp.data_ptr(), None, ws.data_ptr(), ws.numel(), stream_ptr()),
                  "kvq_vq_forward")
            return
        if self._codebook_stamp() != self._E_version:   # first call, or the codebook was written from outside the engine
            self._repack_codebook()
        check(lib().kvq_vq_forward_packed(z.data_ptr(), self.E.data_ptr(), self._epack.data_ptr(), N, self.K, D, G, self.io,
                                          self.beta_vq, z_q.data_ptr(), idx.data_ptr(), loss.data_ptr(), perp.data_ptr(), None,
                                          ws.data_ptr(), ws.numel(), stream_ptr()), "kvq_vq_forward_packed")

    def _vq_forward(self, z, z_q, idx, vq_out):
        """kvq_vq_forward on the encoder output: one codebook (the reference's VectorQuantizer), or G codebooks on G column
        slices as one grouped launch (MultiVectorQuantizer: loss / perplexity = mean over the factors)."""
        N, H = z.shape
        G, K, Dg = self.G, self.K, self.Dg
        ws = _workspace(self.dev, lib().kvq_vq_workspace_bytes(N, K, Dg, G))
        if G == 1:
            self._vq_fwd_call(z, N, H, 1, z_q, idx, vq_out[0:], vq_out[1:], ws)
            zsrc = z
        else:
            vq = self.model.vector_quantizer
            zg = self._buf("zg", (G, N, Dg), z.dtype)
            zg.copy_(vq.split(z))
            zqg = self._buf("zqg", (G, N, Dg), z.dtype)
            lp = self._buf("lp", (2, G), torch.float32)
            self._vq_fwd_call(zg, N, Dg, G, zqg, idx, lp[0], lp[1], ws)
            z_q.copy_(vq.merge(zqg))
            vq_out[0:1].copy_(lp[0].sum(0, keepdim=True) * vq.loss_weight)          # (1 + beta) * MSE over all N * e_dim elements
            vq_out[1:2].copy_(lp[1].mean(0, keepdim=True))
            zsrc = zg
        if self.vq_ema:                      # the EMA step runs after backward: keep what it needs in storage of its own
            self._buf("ema_z", zsrc.shape, zsrc.dtype).copy_(zsrc)
            self._buf("ema_idx", idx.shape, idx.dtype).copy_(idx)
            self._ema_shapes = (tuple(zsrc.shape), zsrc.dtype, tuple(idx.shape))

    def _vq_backward(self, z, idx, g_enc):
        N, H = z.shape
        G, K, Dg = self.G, self.K, self.Dg
        ws = _workspace(self.dev, lib().kvq_vq_workspace_bytes(N, K, Dg, G))
        gE = self.gE.data_ptr() if self.E.requires_grad else None
        if G == 1:
            g_z = torch.empty_like(z)
            gl = self._g_vq if self._g_vq_ext is None else self._g_vq_ext
            check(lib().kvq_vq_backward(z.data_ptr(), self.E.data_ptr(), idx.data_ptr(), g_enc.data_ptr(), gl.data_ptr(), N, K, H, 1,
                                        self.io, self.beta_vq, g_z.data_ptr(), gE, ws.data_ptr(), ws.numel(), stream_ptr()),
                  "kvq_vq_backward")
            return g_z
        vq = self.model.vector_quantizer
        zg, gq = vq.split(z), vq.split(g_enc)
        if self._g_vq_ext is None:
            gl = torch.full((G,), self.w_vq * vq.loss_weight, dtype=torch.float32, device=self.dev)
        else:
            gl = (self._g_vq_ext * vq.loss_weight).expand(G).contiguous()
        gzg = torch.empty_like(zg)
        check(lib().kvq_vq_backward(zg.data_ptr(), self.E.data_ptr(), idx.data_ptr(), gq.data_ptr(), gl.data_ptr(), N, K, Dg, G,
                                    self.io, self.beta_vq, gzg.data_ptr(), gE, ws.data_ptr(), ws.numel(), stream_ptr()),
              "kvq_vq_backward")
        return vq.merge(gzg)

    def _ema_step(self):
        zs, zdt, ishape = self._ema_shapes
        z, idx = self._buf("ema_z", zs, zdt), self._buf("ema_idx", ishape, torch.int64)
        self.model.vector_quantizer.ema_update(z, idx.view(self.G, -1) if self.G > 1 else idx)
        self._repack_codebook()

    # ---- GumbelQuantizer (models/shelgon3/GumbelQuantizer.py:43-83): 1x1 conv = GEMM, row kernel, codebook GEMM ----------------
    def _mm(self, a, b, layout, bias=None):
        """op(a) . op(b) (+ bias) for the six products of the Gumbel mode (n_embed = 512 codes meets the MFMA kernel's requirements,
        the reference analysis' 9 codes go to the any-shape kernel).  layout as nnops.gemm: "nt" | "nn" | "tn"."""
        if self._own_fwd and self.dtype == torch.bfloat16:
            self._gumbel_own += 1
        return self._gemm(a, b, layout, bias=bias)

    def _gumbel_forward(self, z, training):
        gq = self.model.vector_quantizer
        N, H = z.shape
        K, dt = gq.n_embed, self.dtype
        Wp = gq.proj.weight.squeeze(-1).to(dt)                                        # [K, H]
        logits = self._mm(z, Wp, "nt", bias=gq.proj.bias.to(dt))
        hard = bool(gq.straight_through) if training else True                       # :54
        y = torch.empty_like(logits)
        y_soft = torch.empty((N, K), dtype=torch.float32, device=self.dev)
        ind = torch.empty(N, dtype=torch.int64, device=self.dev)
        kl_row = torch.empty(N, dtype=torch.float32, device=self.dev)
        noise = getattr(self, "gumbel_noise", None)           # tests: explicit Gumbel(0,1) samples [N, K] f32 instead of the Philox stream
        check(lib().kvq_gumbel_forward(logits.data_ptr(), noise.data_ptr() if noise is not None else None, N, K,
                                       float(gq.temperature), int(hard), int(self._step_seed), self._site(),
                                       self.io, y.data_ptr(), y_soft.data_ptr(), ind.data_ptr(), kl_row.data_ptr(), stream_ptr()),
              "kvq_gumbel_forward")
        diff = kl_row.mean() * float(gq.kld_scale)                                    # :73
        emb = gq.embed.weight.to(dt)
        z_q = self._mm(y, emb, "nn")                                                  # :66 einsum, already [N, D]
        used = torch.zeros(K, dtype=torch.float32, device=self.dev).scatter_add_(0, ind, self._ones.expand(N))
        perplexity = (used > 0).sum().float()                                         # codes in use (Shelgon.py:63)
        return z_q, diff, perplexity, ind, (z, logits, y, y_soft, Wp, emb)

    def _gumbel_backward(self, g_zq, saved):
        gq = self.model.vector_quantizer
        z, logits, y, y_soft, Wp, emb = saved
        N, K = logits.shape
        g_y = self._mm(g_zq, emb, "nt")
        self.g_emb.copy_(self._mm(y, g_zq, "tn"))
        gd = self._ones.reshape(1) * self.w_vq if self._g_vq_ext is None else self._g_vq_ext.reshape(1)
        g_logits = torch.empty_like(logits)
        check(lib().kvq_gumbel_backward(logits.data_ptr(), y_soft.data_ptr(), g_y.data_ptr(), gd.data_ptr(), N, K, float(gq.temperature),
                                        float(gq.kld_scale), self.io, g_logits.data_ptr(), stream_ptr()), "kvq_gumbel_backward")
        self.g_pw.copy_(self._mm(g_logits, z, "tn").unsqueeze(-1))
        torch.sum(g_logits.float(), dim=0, out=self.g_pb)     # (out=: a same-dtype copy_ would be a memcpy NODE of a captured step)
        return self._mm(g_logits, Wp, "nn")

    def _adam_ranges(self, lo, hi):
        fl = self.flat
        b1, b2 = self.betas
        for (a, b) in fl.ranges:
            a, b = max(a, lo), min(b, hi)
            if a < b and self.fp8 and self._w8_in_adam:
                if a % 8 or b % 8:           # (segments and chunk cuts are multiples of 16: cannot happen; a silent fallback would
                    raise KvqError(f"TrainEngine: Adam range [{a}, {b}) is not 8-aligned")       # leave the fp8 mirror stale)
                # the update also writes the fp8 mirror of the GEMM weights in [a, b) (scales of the last refresh)
                check(lib().kvq_adam_step_dev_fp8(fl.master[a:b].data_ptr(), fl.grad[a:b].data_ptr(), fl.m[a:b].data_ptr(), fl.v[a:b].data_ptr(),
                                                  fl.vmax[a:b].data_ptr() if fl.vmax is not None else None, fl.shadow[a:b].data_ptr(), b - a,
                                                  nnops.io_dtype_of(fl.grad), self._state.data_ptr(), b1, b2, self.eps, self.wd, 1.0,
                                                  self._w8.data_ptr(), self._w8_span.data_ptr(), self._w8_scale.data_ptr(),
                                                  self._w8_off.data_ptr(), self._w8_n.data_ptr(), len(self._w8_index), a, stream_ptr()),
                      "kvq_adam_step_dev_fp8")
            elif a < b:
                nnops.adam_step_dev(fl.master[a:b], fl.grad[a:b], fl.m[a:b], fl.v[a:b], self._state, b1, b2, self.eps, self.wd,
                                    vmax=fl.vmax[a:b] if fl.vmax is not None else None,
                                    shadow=fl.shadow[a:b] if fl.shadow is not fl.master else None)

    def _adam_early(self, lo, hi):
        """[lo, hi) of the flat buffer is final and no kernel of this step reads those weights any more: update it beside backward."""
        main = torch.cuda.current_stream(self.dev)
        self.adam_stream.wait_stream(main)
        with torch.cuda.stream(self.adam_stream):
            self._adam_ranges(lo, hi)
        self._adam_forked = True

    def _adam_join(self):
        if self._adam_forked:
            torch.cuda.current_stream(self.dev).wait_stream(self.adam_stream)
            self._adam_forked = False

    def optimizer_step(self):
        fl = self.flat
        b1, b2 = self.betas
        if self._fuse_opt:
            self._fuse_opt = False
            self._wg_done_lo = fl.n
            if self.vq_ema:
                self._eager(self._ema_step)
            self._step_host += 1
            self._adam_ranges(0, self._adam_hi)           # the head of the buffer: embeddings, final only now
            self._adam_hi = fl.n
            self._adam_aux()
            self._adam_join()
            nnops.step_state_commit(self._state)
            self._after_update()
            return
        cut = 0
        if self._dp:
            cut = self._pending_hi                    # [0, cut) has not been sent yet (embedding gradients)
            self._eager(lambda: self._exchange_head(cut))
            self._pending_hi = fl.n
        self._wg_done_lo = fl.n
        if self.vq_ema:          # the codebook follows the EMA of its assigned encoder outputs (backward has used the old one by now)
            self._eager(self._ema_step)
        self._step_host += 1
        # step += 1, lr after the milestones, bias corrections: computed on the device, read there by the Adam kernels
        nnops.step_state_advance(self._state, self.lr, self.gamma, self.milestones, b1, b2)
        self._adam_ranges(cut, fl.n)                  # multi-GPU: runs while the head of the buffer is still being reduced
        if self._dp:
            self._eager(self._exchange_tail)
            self._adam_ranges(0, cut)
        self._adam_aux()
        self._after_update()

    def _adam_aux(self):
        b1, b2 = self.betas
        for a in self.aux:
            if a["p"].requires_grad:
                nnops.adam_step_dev(a["p"].data.view(-1), a["g"].view(-1), a["m"].view(-1), a["v"].view(-1), self._state,
                                    b1, b2, self.eps, self.wd, vmax=a["vmax"].view(-1) if a["vmax"] is not None else None)

    def _after_update(self):
        if self.vq_kind in ("VectorQuantizer", "MultiVectorQuantizer") and self.E.requires_grad:
            self._repack_codebook()
        if self.fp8:
            self._fp8_quantize_weights(in_step=True)

    def _versions(self):
        return sum(p._version for p in self.param_of.values())

    def refresh_if_params_changed(self):
        """Parameters written from outside (an optimiser step of the autograd path, load_state_dict) since the bf16 shadow was
        last refreshed: refresh it.  The engine's own Adam kernel keeps the shadow current and does not bump tensor versions."""
        v = self._versions()
        if v != self._param_versions:
            self.flat.refresh_shadow()
            if self.fp8:
                self._fp8_quantize_weights()
            self._param_versions = v

    def sync_from_model(self):
        """Call after the model's parameters were written from outside (load_state_dict, manual init): refreshes the bf16
        shadow weights the GEMMs read.  (The f32 master buffer IS the parameters' storage, nothing to copy there.)"""
        self.flat.refresh_shadow()
        if self.fp8:
            self._fp8_quantize_weights()
        if getattr(self, "_epack", None) is not None:
            self._E_version = None             # the codebook pack is rebuilt by the next quantiser call

    @staticmethod
    def supports(model, seq_len: int) -> bool:
        """The engine covers BERT-shaped models with 64-wide heads and sentences of at most 32 tokens (the benchmarked kernels;
        up to 128 in bf16 through the blocked attention kernels)."""
        cfg = model.encoder.config
        kind = type(getattr(model, "vector_quantizer", None)).__name__
        s_max = 128 if getattr(model, "compute_dtype", torch.float32) == torch.bfloat16 else 32
        return kind in _QUANTIZERS + ("NoneType",) and cfg.hidden_size // cfg.num_attention_heads == 64 and seq_len <= s_max \
            and seq_len <= cfg.max_position_embeddings \
            and cfg.hidden_size % 32 == 0 and next(model.parameters()).is_cuda

    def prepare_batch(self, input_ids):
        """What the word-embedding gradient needs besides the ids: the tokens in a STABLE order by id (sorted_ids, perm), pad
        tokens filed under id -1 (nn.Embedding(padding_idx) of BertEmbeddings, modeling_bert.py:60: the pad row receives no
        gradient; kvq_embed_grad ignores negative ids).  Part of building a batch -- hand the result to train_step(prepared=...)
        (dsentences.token_cache does) and the step itself contains no sort; without it the engine sorts before the step."""
        flat_ids = input_ids.reshape(-1)
        if self._pad_idx is not None:
            flat_ids = torch.where(flat_ids == self._pad_idx, torch.full_like(flat_ids, -1), flat_ids)
        srt, perm = torch.sort(flat_ids, stable=True)
        return srt, perm

    def pack_batch(self, input_ids, attention_mask, dec_ids=None, dec_mask=None, target_ids=None):
        """ids | mask | sorted ids | order as ONE int64 tensor [4, B*S]: a replayed step then starts with one device copy.
        With the decoder's own input (a Bagon step): one flat tensor of 4*B*S + 5*B*Sd entries, the same four rows followed by
        the decoder's ids | mask | sorted ids | order | loss target (= the decoder ids unless target_ids is given)."""
        srt, perm = self.prepare_batch(input_ids)
        rows = [input_ids.reshape(-1), attention_mask.reshape(-1).to(torch.int64), srt, perm]
        if dec_ids is None:
            if dec_mask is not None or target_ids is not None:
                raise KvqError("TrainEngine.pack_batch: dec_mask / target_ids without dec_ids")
            return torch.stack(rows)
        dsrt, dperm = self.prepare_batch(dec_ids)
        rows += [dec_ids.reshape(-1), dec_mask.reshape(-1).to(torch.int64), dsrt, dperm,
                 (dec_ids if target_ids is None else target_ids).reshape(-1)]
        return torch.cat(rows)

    @staticmethod
    def unpack_batch(pack, enc_shape, dec_shape=None):
        """Views into a pack_batch() tensor: (ids, mask, sorted ids, order, None | (dec ids, dec mask, sorted, order, target))."""
        flat = pack.reshape(-1)
        N = enc_shape[0] * enc_shape[1]
        Nd = dec_shape[0] * dec_shape[1] if dec_shape is not None else 0
        if flat.numel() != 4 * N + 5 * Nd or flat.dtype != torch.int64:
            raise KvqError(f"TrainEngine: a packed batch of {flat.numel()} {flat.dtype} entries does not fit ids {tuple(enc_shape)}"
                           + (f" + decoder ids {tuple(dec_shape)}" if dec_shape is not None else "")
                           + " (pack_batch() of the same call's ids)")
        r = [flat[i * N:(i + 1) * N] for i in range(4)]
        dec = None
        if dec_shape is not None:
            d = [flat[4 * N + i * Nd: 4 * N + (i + 1) * Nd] for i in range(5)]
            dec = (d[0].view(dec_shape), d[1].view(dec_shape), d[2], d[3], d[4].view(dec_shape))
        return r[0].view(enc_shape), r[1].view(enc_shape), r[2], r[3], dec

    def _normalise_prepared(self, prepared, input_ids, dec_ids=None, check_ids=False):
        """`prepared` of train_step() in one form: dict(pack = the whole batch as pack_batch() laid it out | None,
        enc = (sorted ids, order) | None, dec = likewise for the decoder's ids).  Accepted: None; prepare_batch()'s tuple (the
        encoder side); a dict(enc=tuple, dec=tuple); a pack_batch() tensor.  A pack is authoritative: the step reads ids, masks
        and target from it (eagerly and on replay alike); check_ids compares them with the ids of the call (a device sync:
        done on the first, eager steps of a batch shape)."""
        none = dict(pack=None, enc=None, dec=None)
        if prepared is None:
            return none
        def pair(t, ids, what):
            if t is None:
                return None
            if not isinstance(t, (tuple, list)) or len(t) != 2 or any(not torch.is_tensor(x) or x.numel() != ids.numel() or x.dtype != torch.int64
                                                                      for x in t):
                raise KvqError(f"TrainEngine: prepared {what} = (sorted ids, order) must be prepare_batch() of this call's {what} ids")
            return (t[0].reshape(-1), t[1].reshape(-1))
        if isinstance(prepared, dict):
            if prepared.get("dec") is not None and dec_ids is None:
                raise KvqError("TrainEngine: prepared holds a decoder side but the call passes no dec_ids")
            return dict(none, enc=pair(prepared.get("enc"), input_ids, "encoder"),
                        dec=pair(prepared.get("dec"), dec_ids, "decoder") if dec_ids is not None else None)
        if isinstance(prepared, (tuple, list)):
            return dict(none, enc=pair(prepared, input_ids, "encoder"))
        if not torch.is_tensor(prepared):
            raise KvqError(f"TrainEngine: prepared must be pack_batch()'s tensor or prepare_batch()'s tuple, got {type(prepared).__name__}")
        ids, mask, srt, perm, dec = self.unpack_batch(prepared, input_ids.shape, dec_ids.shape if dec_ids is not None else None)
        if check_ids and not (torch.equal(ids, input_ids) and (dec is None or torch.equal(dec[0], dec_ids))):
            raise KvqError("TrainEngine: the packed batch holds other ids than the ones passed to train_step()")
        return dict(pack=prepared.reshape(-1), enc=(srt, perm), dec=dec[2:4] if dec is not None else None, views=(ids, mask, dec))

    def _train_step_eager(self, input_ids, attention_mask, prep, dec=None):
        if prep["pack"] is not None:              # the pack is what a replayed step reads: the eager step reads the same tensors
            input_ids, attention_mask, dv = prep["views"]
            dec = (dv[0], dv[1], dv[4]) if dv is not None else None
        self._prepared = prep
        dkw = dict(dec_ids=dec[0], dec_mask=dec[1], target_ids=dec[2]) if dec is not None else {}
        try:
            out = self.forward_backward(input_ids, attention_mask, training=self.model.training, compute_grads=True, fuse_optimizer=True,
                                        **dkw)
        finally:
            self._prepared = None
        self.optimizer_step()
        return out

    def train_step(self, input_ids, attention_mask, prepared=None, dec_ids=None, dec_mask=None, target_ids=None):
        """One optimiser step.  dec_ids / dec_mask / target_ids: the decoder's own input and the loss target of a Bagon step
        (models/bagon/Trainer.py:78-110; default: the autoencoding step, decoder input = target = input_ids).  prepared: see
        _normalise_prepared().  Once a batch shape has been seen twice the step is replayed from a chain of hipGraphs
        (_StepGraphs: the quantiser and, on multi-GPU runs, the RCCL all-reduces stay eager launches between the graphs); the
        host then issues a handful of launches per step instead of ~800.  KVQ_GRAPH=0 keeps every launch eager."""
        if dec_ids is None and (dec_mask is not None or target_ids is not None):
            if target_ids is None:
                raise KvqError("TrainEngine.train_step: dec_mask without dec_ids")
            dec_ids, dec_mask = input_ids, attention_mask           # a separate target alone: the decoder still reads the encoder's ids
        dec = (dec_ids, dec_mask, target_ids) if dec_ids is not None else None
        key = (tuple(input_ids.shape), bool(self.model.training), tuple(dec_ids.shape) if dec is not None else None)
        seen = self._eager_seen.get(key, 0)
        prep = self._normalise_prepared(prepared, input_ids, dec_ids, check_ids=seen < 2)
        if not self.use_graph:
            self._eager_seen[key] = seen + 1
            return self._train_step_eager(input_ids, attention_mask, prep, dec)
        g = self._graphs.get(key)
        if g is None:
            if seen < 2 or len(self._graphs) >= 4:      # warm the workspaces / GEMM plans eagerly first; few shapes only
                self._eager_seen[key] = seen + 1
                return self._train_step_eager(input_ids, attention_mask, prep, dec)
            try:
                if prep["pack"] is not None:
                    ids0, mask0, dv = prep["views"]
                    dec0 = (dv[0], dv[1], dv[4]) if dv is not None else None
                else:
                    ids0, mask0, dec0 = input_ids, attention_mask, dec
                g = self._graphs[key] = _StepGraphs(self, ids0, mask0, dec0)
            except Exception as e:                       # capture is an optimisation: never let it take a run down
                if os.environ.get("KVQ_GRAPH_STRICT", "0") == "1":      # (tests: a capture that fails is a failure)
                    self._abandon_capture()
                    raise
                import sys
                print(f"[kvq] hipGraph capture of the training step failed ({type(e).__name__}: {e}); "
                      f"continuing with eager launches", file=sys.stderr, flush=True)
                self._abandon_capture()
                return self._train_step_eager(input_ids, attention_mask, prep, dec)
        return g.run(input_ids, attention_mask, prep, dec)

    def _abandon_capture(self):
        """Leave a failed capture behind in a state from which eager steps can go on (same collectives, same order)."""
        self.use_graph, self._cap = False, None
        self._graphs.clear()
        try:
            torch.cuda.synchronize(self.dev)
        except Exception:
            pass
        if self._dp:
            try:
                self._settle(self._works)
                self._exchange_tail()
            except Exception:
                self._works, self._works_late = [], []
            self._pending_hi = self.flat.n
        self._wg_done_lo = self.flat.n
        self._red_items, self._red_keep, self._wg_pending = [], [], False
        self._wg_items, self._wg_keep, self._wg_keep_step = [], [], []
        self._fuse_opt, self._adam_forked, self._adam_hi = False, False, self.flat.n

    def eval_step(self, input_ids, attention_mask, dec_ids=None, dec_mask=None, target_ids=None):
        return self.forward_backward(input_ids, attention_mask, training=False, compute_grads=False, dec_ids=dec_ids, dec_mask=dec_mask,
                                     target_ids=target_ids)
