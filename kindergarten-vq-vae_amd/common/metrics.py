"""Token accuracy of a reconstruction -- the contract of the reference's common/metrics.py:8-36.

In the training step the per-batch value comes for free out of the fused loss kernel (kvq_ce_forward); this torch
version serves callers that only hold id tensors (analyses, tests)."""
import torch
from torch import Tensor


def _check_ids(name: str, t: Tensor):
    if t.is_floating_point():
        raise AssertionError(f"{name} tensor must be integer type, not floating point")


def seq_acc(input: Tensor, target: Tensor):
    """Returns (accuracy over every token of the batch, accuracy of each sentence [B]) for id tensors of one shape."""
    if input.shape != target.shape:
        raise AssertionError("input and target shapes must match")
    _check_ids("input", input)
    _check_ids("target", target)
    hits = torch.eq(input, target)
    return hits.sum() / hits.numel(), hits.to(torch.float32).mean(dim=-1)
