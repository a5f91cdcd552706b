"""Run only the fused VQ forward/backward at the BASELINE config-2 shape (profiling target for rocprofv3 --pmc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kindergarten-vq-vae_amd"))
import kvq  # noqa: E402

N, K, D = int(os.environ.get("VQ_N", 8192)), int(os.environ.get("VQ_K", 512)), 768
dt = torch.bfloat16 if os.environ.get("VQ_DT", "bf16") == "bf16" else torch.float32
torch.manual_seed(0)
z = torch.randn(N, D, device="cuda").to(dt).requires_grad_(True)
E = torch.randn(K, D, device="cuda").requires_grad_(True)
for _ in range(int(os.environ.get("VQ_ITERS", 6))):
    loss, z_q, perp, idx, counts = kvq.vector_quantize(z, E, 0.25)
    (loss + z_q.float().sum()).backward()
torch.cuda.synchronize()
print("ok", float(loss), float(perp))
