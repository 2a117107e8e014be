"""the fused attention kernels at config 2's shapes (B 32, H 4, T' 249, d_k 64, legacy rel_shift, fp32): a few forward / backward
calls for a PMC pass (tools/pmc_attn.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import functional as F_  # noqa: E402

espnet_amd.set_precision("fp32")
DEV = torch.device("cuda")
B, T, H, dk = 32, 249, 4, 64
D = H * dk
g = torch.Generator().manual_seed(1)
r = lambda *s: (0.5 * torch.randn(*s, generator=g)).to(DEV)  # noqa: E731
qu, qv, k, v, p, dctx = r(B * T, D), r(B * T, D), r(B * T, D), r(B * T, D), r(T, D), r(B * T, D)
mask = torch.ones(B, 1, T, dtype=torch.uint8, device=DEV)
for _ in range(3):
    P1, _, _ = F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T, T, H, dk)
    F_.attn_core_bwd(dctx, P1, qu, qv, k, v, p, B, T, T, H, dk)
torch.cuda.synchronize()
print("done")
