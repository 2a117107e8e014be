#!/usr/bin/env python3
"""dwconv fwd / bwd_x / bwd_w against torch.conv1d (float64) over a sweep of shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402
DEV = "cuda"
g = torch.Generator().manual_seed(0)
for (B, T, Cc, K) in ((3, 29, 64, 15), (4, 80, 128, 31), (2, 80, 256, 31), (4, 80, 128, 29), (4, 16, 128, 31),
                      (4, 17, 128, 31), (1, 33, 128, 31), (4, 80, 300, 31), (32, 249, 256, 31)):
    x = torch.randn(B, T, Cc, generator=g); w = torch.randn(Cc, 1, K, generator=g); bias = torch.randn(Cc, generator=g)
    dy = torch.randn(B, T, Cc, generator=g)
    xd = x.double().requires_grad_(True); wd_ = w.double().requires_grad_(True); bd_ = bias.double().requires_grad_(True)
    yr = torch.nn.functional.conv1d(xd.transpose(1, 2), wd_, bd_, padding=(K - 1) // 2, groups=Cc).transpose(1, 2)
    yr.backward(dy.double())
    y = ops.dwconv_fwd(x.to(DEV), w.view(Cc, K).to(DEV), bias.to(DEV), B, T, Cc, K)
    dx = ops.dwconv_bwd_x(dy.to(DEV), w.view(Cc, K).to(DEV), B, T, Cc, K)
    dw, db = torch.zeros(Cc, K, device=DEV), torch.zeros(Cc, device=DEV)
    ops.dwconv_bwd_w(dy.to(DEV), x.to(DEV), dw, db, B, T, Cc, K)
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    e = wd_.grad.view(Cc, K)
    per_k = ((dw.double().cpu() - e).norm(dim=0) / e.norm(dim=0))
    print((B, T, Cc, K), "fwd %.1e dx %.1e dw %.1e db %.1e" % (rel(y, yr.detach()), rel(dx, xd.grad), rel(dw, e), rel(db, bd_.grad)),
          "bad k:", [int(i) for i in torch.nonzero(per_k > 1e-4).flatten()][:12])
