#!/usr/bin/env python3
"""Per-launch time of the HBM-bound kernels at the BASELINE C2 shapes (M = 32*249 rows of 256)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402
from tools.gemm_bench import time_call  # noqa: E402
DEV = "cuda"
B, T, D, K = 32, 249, 256, 31
M = B * T
x = torch.randn(M, D, device=DEV); dy = torch.randn(M, D, device=DEV)
g = torch.rand(D, device=DEV) + 0.5; b = torch.randn(D, device=DEV)
w = torch.randn(D, K, device=DEV)
dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12)
rm, rv = torch.zeros(D, device=DEV), torch.ones(D, device=DEV)
bm, br = ops.bn_stats(x, M, D, 1e-5, 0.1, rm, rv)
dw = torch.zeros(D, K, device=DEV)
xin = torch.randn(32, 1000, 80, device=DEV); c1w = torch.randn(256, 9, device=DEV); c1b = torch.randn(256, device=DEV)
y1 = ops.conv1_fwd(xin, c1w, c1b, 32, 1000, 80, 256, torch.bfloat16)
dc1w, dc1b = torch.zeros(256, 9, device=DEV), torch.zeros(256, device=DEV)
tests = {
    "layernorm_fwd": lambda: ops.layernorm_fwd(x, g, b, 1e-12, torch.bfloat16),
    "layernorm_bwd": lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dy, dg, db),
    "dwconv_fwd": lambda: ops.dwconv_fwd(x, w, b, B, T, D, K),
    "dwconv_bwd_w": lambda: ops.dwconv_bwd_w(dy, x, dw, db, B, T, D, K),
    "bn_stats": lambda: ops.bn_stats(x, M, D, 1e-5, 0.1, rm, rv),
    "bn_apply": lambda: ops.bn_apply(x, bm, br, g, b, M, D, 2, torch.bfloat16),
    "bn_bwd": lambda: ops.bn_bwd(dy, x, bm, br, g, b, dg, db, M, D, 2, True),
    "colsum": lambda: ops.colsum(x, db),
    "conv1_fwd": lambda: ops.conv1_fwd(xin, c1w, c1b, 32, 1000, 80, 256, torch.bfloat16),
    "conv1_bwd_w": lambda: ops.conv1_bwd_w(y1, xin, dc1w, dc1b, 32, 1000, 80, 256),
    "dropout": lambda: ops.dropout(x, 0.1, 5),
    "cast_bf16": lambda: ops.cast_bf16(x),
}
for k, f in tests.items():
    print("%-16s %8.1f us" % (k, time_call(f)))
