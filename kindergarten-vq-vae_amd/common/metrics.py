"""Token accuracy (counterpart of the reference's common/metrics.py:8-36).

In the training step the per-batch accuracy comes for free out of the fused loss kernel (kvq_ce_forward);
this torch version serves callers that hold id tensors only (analyses, tests)."""
import torch
from torch import Tensor


def seq_acc(input: Tensor, target: Tensor):
    """(accuracy over all tokens of the batch, accuracy per sentence).  Integer tensors of equal shape."""
    assert input.shape == target.shape, "input and target shapes must match"
    assert not input.is_floating_point(), "input tensor must be integer type, not floating point"
    assert not target.is_floating_point(), "target tensor must be integer type, not floating point"
    same = input == target
    acc_per_batch = same.sum() / input.numel()
    acc_per_sentence = same.float().mean(dim=-1)
    return acc_per_batch, acc_per_sentence
