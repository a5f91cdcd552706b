"""Token-noise helper of the plain Bagon trainer (counterpart of common/tensor_utils.py:13-49).
Off the hot path: it returns its input untouched when the percentage is ~0, which is how Shelgon runs."""
import math

import torch


def replace_pct_rand_values(tensor: torch.Tensor, percentage: float, rand_int_low: int, rand_int_high: int):
    """Replace exactly int(numel*percentage) randomly placed ids by uniform random ids in [low, high)."""
    if math.isclose(percentage, 0):
        return tensor
    n = tensor.numel()
    n_noise = int(n * percentage)
    pos = torch.randperm(n, device=tensor.device)[:n_noise]
    out = tensor.reshape(-1).clone()
    out[pos] = torch.randint(rand_int_low, rand_int_high, (n_noise,), device=tensor.device, dtype=tensor.dtype)
    return out.reshape(tensor.shape)
