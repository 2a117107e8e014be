#!/usr/bin/env python3
"""a handful of fp32 GEMM launches for a rocprofv3 --pmc pass (tools/gemm_f32_probe.py shapes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402

ops.set_precision("fp32")
g = torch.Generator().manual_seed(0)
M = 7968
for name, m, n, k, ta, tb, sk, tile in (("ffn_w1", M, 2048, 256, 0, 0, 1, 128), ("ffn_w2", M, 256, 2048, 0, 0, 1, 64),
                                         ("dW1", 2048, 256, M, 1, 1, 8, 64), ("proj", M, 256, 256, 0, 0, 1, 64)):
    A = torch.randn((k, m) if ta else (m, k), generator=g).to("cuda")
    B = torch.randn((k, n) if tb else (n, k), generator=g).to("cuda")
    C = torch.zeros(m, n, device="cuda")
    for _ in range(5):
        ops.gemm(A, B, C, m, n, k, (m if ta else k), (n if tb else k), n, transA=ta, transB=tb, splitk=sk, tile=tile, precision=0)
    torch.cuda.synchronize()
