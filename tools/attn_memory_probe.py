#!/usr/bin/env python3
"""Peak device memory of ONE Conformer encoder block (d 256, 4 heads, units 2048, kernel 31; forward + backward, dropout 0.1) at
B = 32 and T' = 249 / 1000 / 2048 / 4096 frames, next to the bytes of the stored attention probabilities (P and its dropped copy:
B x H x T' x ld(T') x 4 each - the O(T'^2) part; backward adds dS and dbd of the same size while it runs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import espnet_amd  # noqa: E402
from espnet_amd import functional as F_  # noqa: E402
from espnet_amd.nets import modules as M  # noqa: E402

espnet_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")
dev = torch.device("cuda")
D, H, U = 256, 4, 2048
for B, T in ((32, 249), (32, 1000), (8, 2048), (4, 4096)):
    torch.manual_seed(0)
    layer = M.ConformerEncoderLayer(D, M.RelPositionMultiHeadedAttention(H, D, 0.1), M.PositionwiseFeedForward(D, U, 0.1, "swish"),
                                    M.PositionwiseFeedForward(D, U, 0.1, "swish"), M.ConvolutionModule(D, 31, "swish"), 0.1).to(dev).train()
    x = torch.randn(B, T, D, device=dev, requires_grad=True)
    pos = M.RelPositionalEncoding(D, 0.0).to(dev)(x.detach())[1]
    mask = torch.ones(B, 1, T, dtype=torch.bool, device=dev)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    (y, _), _ = layer((x, pos), mask)
    fwd = torch.cuda.max_memory_allocated() - base
    kept = torch.cuda.memory_allocated() - base
    y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    esz = 4 if espnet_amd.ops.get_precision() == "fp32" else 2
    pbytes = B * H * T * F_._ldp(T) * esz
    print("B %2d T' %4d: kept for backward %7.1f MB, peak forward %7.1f MB, peak forward+backward %7.1f MB | P alone %7.1f MB (x2 with "
          "dropout, x4 while backward runs)" % (B, T, kept / 2**20, fwd / 2**20, peak / 2**20, pbytes / 2**20), flush=True)
    del layer, x, y, pos, mask
    torch.cuda.empty_cache()
