#!/usr/bin/env python3
"""Ablation of the fused FFN kernels (EAMD_FFN_DEBUG bits: 2 no tile loads/stores, 4 no barriers, 8 no epilogue, 16 no fragment reads)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import ops  # noqa: E402
from tools.gemm_probe4 import graph_time  # noqa: E402
espnet_amd.set_precision("fp32")
M, D, F = 7968, 256, 2048
dev = "cuda"
x = torch.randn(M, D, device=dev); w1 = torch.randn(F, D, device=dev) * 0.05; b1 = torch.zeros(F, device=dev)
w2 = torch.randn(D, F, device=dev) * 0.05; b2 = torch.zeros(D, device=dev); R = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev)
ops.manual_seed(1)
drop = (0.1, 11, 0.1, 12)
out, f, h = ops.ffn_fwd(x, w1, b1, w2, b2, act=ops.ACT_SWISH, alpha=0.5, R=R, drop=drop)
for bits in (0, 2, 4, 8, 16, 2 | 4, 2 | 16, 4 | 16, 2 | 4 | 8, 2 | 4 | 16, 2 | 4 | 8 | 16):
    os.environ["EAMD_FFN_DEBUG"] = str(bits)
    tf = graph_time(lambda: ops.ffn_fwd(x, w1, b1, w2, b2, act=ops.ACT_SWISH, alpha=0.5, R=R, drop=drop), n=10)
    tb = graph_time(lambda: ops.ffn_bwd(dy, w1, w2, f, alpha=0.5), n=10)
    print("bits %2d  fwd %7.1f us   bwd %7.1f us" % (bits, tf, tb), flush=True)
