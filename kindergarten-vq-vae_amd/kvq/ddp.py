"""Data-parallel gradient exchange for the dSentences training loop: one process per GPU, RCCL over xGMI.

The reference is single-process; this is the build's multi-GPU design (§8e).  The model is
replicated, sentences are sharded, and the ONLY exchange per step is the sum of parameter gradients:

  * gradients live in a few large flat buckets (default 64 MiB of f32; `p.grad` of every trainable parameter is a
    VIEW into its bucket), filled in reverse parameter order so the buckets complete in the order backward
    produces them (LM head / last decoder layers first);
  * when the last gradient of a bucket has been accumulated (post-accumulate-grad hook) the bucket is all-reduced
    on a dedicated communication stream, overlapping the rest of backward; `finish()` makes the compute stream
    wait for the comm stream before the optimizer reads the gradients;
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, large messages keep every link busy, so buckets
    are sized in tens of MiB rather than DDP's 25 MB default, and the reduction is AVG in RCCL itself.

Works on CPU tensors with the gloo backend too (tests/test_ddp_gloo.py) -- there the "streams" degrade to
synchronous calls.  Losses are means over the local batch; with equal batch per rank the averaged gradient equals
the gradient of the global-batch mean.
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract) and create the process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (KVQ_DP_SINGLE_RANK=1: a ONE-rank group, so that the RCCL branch of the engine can be rehearsed on a one-GPU box)
    if (world > 1 or os.environ.get("KVQ_DP_SINGLE_RANK", "0") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" IS RCCL on ROCm.  KVQ_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
            # (all ranks then share the visible devices round-robin; the exchange goes through the host)
            backend = os.environ.get("KVQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend != "nccl" and torch.cuda.is_available():
            local = local % torch.cuda.device_count()
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class _Bucket:
    __slots__ = ("flat", "params", "pending", "total", "work")

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.total = len(params)
        self.pending = self.total
        self.work = None


class GradSync:
    """Bucketed, overlapped gradient all-reduce.  Usage per step:

        sync.zero_grad(); loss.backward(); sync.finish(); opt.step()
    """

    def __init__(self, params, bucket_mib: int = 64, process_group=None, grad_dtype: torch.dtype | None = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradSync: no trainable parameters")
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        dev = self.params[0].device
        self.device = dev
        self.on_gpu = dev.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=dev) if (self.on_gpu and self.world > 1) else None
        cap = bucket_mib * (1 << 20)
        self.buckets: List[_Bucket] = []
        cur, cur_bytes = [], 0
        for p in reversed(self.params):            # reverse registration order ~ order of gradient readiness
            nbytes = p.numel() * (grad_dtype or p.dtype).itemsize
            if cur and cur_bytes + nbytes > cap:
                self._close(cur, grad_dtype)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur, grad_dtype)
        self._hooks = []
        if self.world > 1:
            for b in self.buckets:
                for p in b.params:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(b)))

    def _close(self, plist, grad_dtype):
        dt = grad_dtype or plist[0].dtype
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=dt, device=plist[0].device)
        off = 0
        for p in plist:
            if (grad_dtype or p.dtype) != dt:
                raise ValueError("GradSync: mixed gradient dtypes inside one bucket")
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.buckets.append(_Bucket(flat, plist))

    def _make_hook(self, bucket: _Bucket):
        def hook(_p):
            bucket.pending -= 1
            if bucket.pending == 0:
                self._launch(bucket)
        return hook

    def _launch(self, b: _Bucket):
        op = dist.ReduceOp.AVG if self.on_gpu else dist.ReduceOp.SUM
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))   # gradients of this bucket are final
            with torch.cuda.stream(self.comm_stream):
                b.work = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)

    def zero_grad(self):
        """One memset per bucket (instead of one per parameter); re-arms the bucket counters."""
        for b in self.buckets:
            b.flat.zero_()
            b.pending = b.total
            b.work = None
            for p in b.params:                      # a foreign optimizer.zero_grad(set_to_none=True) would detach the views
                if p.grad is None or p.grad.data_ptr() < b.flat.data_ptr():
                    self._rebind(b)
                    break

    def _rebind(self, b: _Bucket):
        off = 0
        for p in b.params:
            p.grad = b.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def finish(self):
        """Block the compute stream (not the host) until every bucket is reduced.  Buckets whose hooks never fired
        (parameters unused this step) are reduced here so all ranks issue the same collectives."""
        if self.world == 1:
            return
        for b in self.buckets:
            if b.work is None:
                self._launch(b)
        for b in self.buckets:
            b.work.wait()
            if not self.on_gpu:
                b.flat.div_(self.world)
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)

    def grad_bytes(self) -> int:
        return sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """Make every rank start from rank `src`'s weights (what DDP's constructor does)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)


def all_reduce_mean_(t: torch.Tensor, process_group=None) -> torch.Tensor:
    """Logging-only reduction of scalar stats / code histograms."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
        t.div_(dist.get_world_size(process_group))
    return t


def agree(flag: bool, src: int = 0, process_group=None) -> bool:
    """Rank `src`'s view of a host-side decision, on every rank, behind a barrier (so that what the decision is about -- a
    checkpoint rank 0 has just written -- is on disk): every rank then enters or skips the SAME collectives."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return bool(flag)
    dist.barrier(group=process_group)
    box = [bool(flag)]
    dist.broadcast_object_list(box, src=src, group=process_group)
    return bool(box[0])


def gather_lists(items: list, dst: int = 0, process_group=None) -> list:
    """Every rank's list of (picklable) records concatenated in rank order on rank `dst`; the other ranks get []."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return items
    world = dist.get_world_size(process_group)
    out = [None] * world if dist.get_rank(process_group) == dst else None
    dist.gather_object(items, out, dst=dst, group=process_group)
    return [r for part in out for r in part] if out is not None else []


def same_everywhere(obj, src: int = 0, process_group=None):
    """Rank `src`'s value of a small picklable object on every rank (the run id: each rank reads its own clock)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=process_group)
    return box[0]
