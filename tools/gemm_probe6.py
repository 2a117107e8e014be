#!/usr/bin/env python3
"""split-K on the long-reduction skinny GEMMs (ffn down-projection / its input gradient), graph-replay device time"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import ops  # noqa: E402
from tools.gemm_probe4 import graph_time  # noqa: E402
DEV = "cuda"
espnet_amd.set_precision("bf16")
M = 7968
for (N, K, tb) in ((256, 2048, 0), (256, 2048, 1), (256, 5000, 1), (512, 256, 0)):
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    B = (torch.randn(K, N, device=DEV) if tb else torch.randn(N, K, device=DEV)).to(torch.bfloat16)
    C = torch.zeros(M, N, device=DEV)
    for sk in (1, 2, 3, 4):
        def f():
            if sk > 1:
                C.zero_()
            ops.gemm(A, B, C, M, N, K, K, N if tb else K, N, transB=tb, splitk=sk)
        t = graph_time(f)
        print("M=%d N=%d K=%d tb=%d splitk=%d: %6.1f us  %6.1f TF/s" % (M, N, K, tb, sk, t, 2.0 * M * N * K / t / 1e6))
