"""CPU restatement of the reference's GumbelQuantizer.forward -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline may import oracle/; the product path never does).

Follows /root/reference/models/shelgon3/GumbelQuantizer.py line by line, with the Gumbel noise as an explicit input
(torch.nn.functional.gumbel_softmax draws it internally):
    :56      logits = proj(z)                         1x1 Conv1d == z . W^T + b per token
    :58      soft_one_hot = gumbel_softmax(logits, tau, dim=codes, hard)
                 y_soft = softmax((logits + g) / tau);  hard: one_hot(argmax y_soft) - y_soft.detach() + y_soft
    :66      z_q = einsum('b n s, n d -> b d s', soft_one_hot, embed.weight)          == y . E per token
    :70-73   qy = softmax(logits);  diff = kld_scale * sum_codes(qy * log(qy * n_embed + 1e-10)).mean()
    :76      ind = soft_one_hot.argmax(codes)
Pinned by tests/golden/gumbel_*.npz, which tests/golden/make_gumbel_golden.py produced by running the reference module itself
(tests/test_oracle_golden.py::test_gumbel_oracle_matches_reference).  numpy float32 for the forward; gradients through the same
expression in torch (CPU autograd).
"""
import numpy as np


def _softmax(x):
    x = x - x.max(-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(-1, keepdims=True)


def forward(z, W, b, E, noise, tau, hard, kld_scale):
    """z [B,S,H], W [K,H], b [K], E [K,D], noise [B,S,K] (Gumbel(0,1) samples) -> dict(z_q [B,S,D], diff, ind [B,S], y, y_soft)."""
    z, W, b, E, noise = (np.asarray(a, np.float32) for a in (z, W, b, E, noise))
    K = W.shape[0]
    logits = z @ W.T + b                                                     # :56
    y_soft = _softmax((logits + noise) / np.float32(tau))                    # :58
    ind = y_soft.argmax(-1)                                                  # :76 (first maximum)
    y = y_soft
    if hard:
        one_hot = np.eye(K, dtype=np.float32)[ind]
        y = (one_hot - y_soft) + y_soft
    z_q = y @ E                                                              # :66
    qy = _softmax(logits)                                                    # :70
    diff = np.float32(kld_scale) * (qy * np.log(qy * np.float32(K) + np.float32(1e-10))).sum(-1).mean()      # :73
    return dict(z_q=z_q.astype(np.float32), diff=np.float32(diff), ind=ind.astype(np.int64), y=y, y_soft=y_soft, logits=logits)


def forward_backward_torch(z, W, b, E, noise, tau, hard, kld_scale, G, c):
    """Same expression in torch on the CPU; returns the forward outputs and d[(z_q*G).sum() + c*diff] / d(z, W, b, E)."""
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.tensor(np.asarray(a, np.float32), requires_grad=True)
    z, W, b, E = t(z), t(W), t(b), t(E)
    noise = torch.tensor(np.asarray(noise, np.float32))
    K = W.shape[0]
    logits = z @ W.t() + b
    y_soft = ((logits + noise) / tau).softmax(-1)
    ind = y_soft.argmax(-1)
    y = (F.one_hot(ind, K).float() - y_soft.detach() + y_soft) if hard else y_soft
    z_q = y @ E
    qy = logits.softmax(-1)
    diff = kld_scale * (qy * torch.log(qy * K + 1e-10)).sum(-1).mean()
    ((z_q * torch.tensor(np.asarray(G, np.float32))).sum() + diff * float(c)).backward()
    return dict(z_q=z_q.detach().numpy(), diff=float(diff), ind=ind.numpy(), grad_z=z.grad.numpy(), grad_W=W.grad.numpy(),
                grad_b=b.grad.numpy(), grad_E=E.grad.numpy())
