#!/usr/bin/env python3
"""Eager launches of the fused FFN kernels and the GEMM pairs (config-2 shapes) for rocprofv3 --pmc / --kernel-trace."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import ops  # noqa: E402

espnet_amd.set_precision("fp32")
M, D, F = 7968, 256, 2048
dev = "cuda"
x = torch.randn(M, D, device=dev); w1 = torch.randn(F, D, device=dev) * 0.05; b1 = torch.zeros(F, device=dev)
w2 = torch.randn(D, F, device=dev) * 0.05; b2 = torch.zeros(D, device=dev); R = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev)
ops.manual_seed(1)
drop = (0.1, 11, 0.1, 12)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    out, f, h = ops.ffn_fwd(x, w1, b1, w2, b2, act=ops.ACT_SWISH, alpha=0.5, R=R, drop=drop)
    dz, dx = ops.ffn_bwd(dy, w1, w2, f, alpha=0.5)
    hh = torch.empty(M, F, device=dev)
    z = ops.linear_fwd(x, w1, b1, drop=(0.1, 11), Hb=hh, h_act=ops.ACT_SWISH, act=ops.EPI_DACT_FACTOR)
    o2 = ops.linear_fwd(hh, w2, b2, R=R, alpha=0.5, drop=(0.1, 12))
    dz2 = ops.linear_bwd_x(dy, w2, epilogue=ops.EPI_MUL_AUX, aux=f, alpha=0.5)
    dx2 = ops.linear_bwd_x(dz2, w1)
torch.cuda.synchronize()
