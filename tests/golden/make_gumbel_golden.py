#!/usr/bin/env python3
"""Generate the Gumbel-quantiser golden vectors by RUNNING the reference's own module.

Run only in the build container (the reference tree does not travel):

    python3 tests/golden/make_gumbel_golden.py

Imports `GumbelQuantizer` from /root/reference/models/shelgon3/GumbelQuantizer.py (it depends on torch and rich only), gives it
seeded parameters and inputs and stores inputs + outputs in tests/golden/gumbel_*.npz -- numbers only, no source text.
torch.nn.functional.gumbel_softmax draws its noise inside the call; it is the FIRST use of the generator in forward, so the
same numbers are reproduced by re-seeding and drawing `-empty(B,K,S).exponential_().log()` (checked below by replaying
forward with a patched sampler), and stored as noise[B,S,K].

Per case: z[B,S,H], W[K,H] (= proj.weight[:, :, 0]), b[K], E[K,D], noise[B,S,K], tau, kld_scale, straight_through, is_training,
G[B,S,D] (upstream gradient of z_q), c (upstream gradient of diff) -> z_q, diff, ind[B,S], grad_z, grad_W, grad_b, grad_E.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/models/shelgon3")
from GumbelQuantizer import GumbelQuantizer  # noqa: E402  (reference module)

OUT = os.path.dirname(os.path.abspath(__file__))
CASES = {
    # name: (seed, B, S, H, K, D, tau, kld_scale, straight_through, is_training)
    "gumbel_tiny_hard": (0, 2, 5, 16, 8, 12, 1.0, 5e-4, True, True),
    "gumbel_soft": (1, 3, 7, 32, 64, 32, 0.7, 1e-2, False, True),
    "gumbel_eval": (2, 2, 4, 16, 8, 12, 1.0, 5e-4, False, False),          # eval forces hard (:54)
    "gumbel_wide": (3, 4, 12, 64, 512, 64, 2.0, 5e-4, True, True),
    "gumbel_odd": (4, 1, 3, 24, 70, 20, 0.5, 1.0, True, True),             # K not a multiple of 64
}


def run(name, seed, B, S, H, K, D, tau, kld, st, training):
    torch.manual_seed(1000 + seed)
    m = GumbelQuantizer(enc_out_size=H, n_embed=K, embedding_dim=D, temperature=tau, kl_div_scale=kld, straight_through=st)
    z = torch.randn(B, S, H, requires_grad=True)
    G = torch.randn(B, S, D)
    c = float(torch.rand(()) + 0.5)
    torch.manual_seed(seed)
    z_q, diff, ind = m(z, training)
    torch.manual_seed(seed)
    noise_bks = -torch.empty(B, K, S).exponential_().log()                 # what gumbel_softmax drew
    (z_q * G).sum().add(diff * c).backward()
    # self-check of the captured noise: recompute the forward by hand
    logits = torch.einsum("bsh,kh->bks", z.detach(), m.proj.weight[:, :, 0]) + m.proj.bias[None, :, None]
    y_soft = ((logits + noise_bks) / tau).softmax(1)
    assert torch.equal(y_soft.argmax(1), ind), name
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        z=z.detach().numpy(), W=m.proj.weight[:, :, 0].detach().numpy(), b=m.proj.bias.detach().numpy(),
        E=m.embed.weight.detach().numpy(), noise=noise_bks.permute(0, 2, 1).contiguous().numpy(),
        tau=np.float32(tau), kld_scale=np.float32(kld), straight_through=np.bool_(st), is_training=np.bool_(training),
        G=G.numpy(), c=np.float32(c),
        z_q=z_q.detach().numpy(), diff=diff.detach().numpy(), ind=ind.numpy(),
        grad_z=z.grad.numpy(), grad_W=m.proj.weight.grad[:, :, 0].numpy(), grad_b=m.proj.bias.grad.numpy(), grad_E=m.embed.weight.grad.numpy())
    print(f"{name}: diff={float(diff):.6g} codes used={ind.unique().numel()}/{K}")


if __name__ == "__main__":
    for name, args in CASES.items():
        run(name, *args)
