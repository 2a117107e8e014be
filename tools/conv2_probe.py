#!/usr/bin/env python3
"""conv2 of Conv2dSubsampling (3x3 stride 2, 256 -> 256 channels) at the headline shape: the implicit-GEMM forward,
weight gradient and the four stride-parity input-gradient launches, 64x64 vs 128x128 tiles, graph-replay device time"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import ops  # noqa: E402
from espnet_amd.functional import _CLASSES, _TAPS_FWD  # noqa: E402
from tools.gemm_probe4 import graph_time  # noqa: E402
DEV = "cuda"
espnet_amd.set_precision("bf16")
B, T, F, Cc = 32, 1000, 80, 256
H1, W1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
H2, W2 = (H1 - 3) // 2 + 1, (W1 - 3) // 2 + 1
M2 = B * H2 * W2
bf = torch.bfloat16
y1 = torch.randn(B * H1 * W1, Cc, device=DEV).to(bf)
wf = torch.randn(9 * Cc, Cc, device=DEV).to(bf)
y2 = torch.empty(M2, Cc, device=DEV, dtype=bf)
bias = torch.randn(Cc, device=DEV)
g = ops.make_gather(Cc, _TAPS_FWD, H2, W2, H1, W1, 2, 2)
fl = 2.0 * M2 * Cc * 9 * Cc
for tile in (64, 128):
    t = graph_time(lambda: ops.gemm(y1, wf, y2, M2, Cc, 9 * Cc, 9 * Cc, Cc, Cc, transB=1, bias=bias, epilogue=ops.EPI_RELU,
                                    gather=g, tile=tile), n=20)
    print("fwd  tile %3d: %7.1f us %6.1f TF/s" % (tile, t, fl / t / 1e6))
dy2 = torch.randn(M2, Cc, device=DEV).to(bf)
dwf = torch.zeros(9 * Cc, Cc, device=DEV)
for tile in (64, 128):
    for sk in (4, 8, 16, 32):
        def f():
            dwf.zero_()
            ops.gemm(y1, dy2, dwf, 9 * Cc, Cc, M2, 9 * Cc, Cc, Cc, transA=1, transB=1, gather=g, splitk=sk, tile=tile)
        t = graph_time(f, n=20)
        print("dW   tile %3d splitk %2d: %7.1f us %6.1f TF/s" % (tile, sk, t, fl / t / 1e6))
wd = torch.randn(9 * Cc, Cc, device=DEV).to(bf)
dy1 = torch.empty(B * H1 * W1, Cc, device=DEV, dtype=bf)
for tile in (64, 128):
    def f():
        q0 = 0
        for (ph, pw), taps in _CLASSES:
            Ho, Wo = (H1 - ph + 1) // 2, (W1 - pw + 1) // 2
            gt = ops.make_gather(Cc, [((ph - kh) // 2, (pw - kw) // 2) for kh, kw in taps], Ho, Wo, H2, W2, 1, 1)
            cm = ops.make_rowmap(Ho, Wo, H1, W1, 2, ph, 2, pw)
            nt = len(taps)
            ops.gemm(dy2, wd, dy1, B * Ho * Wo, Cc, nt * Cc, nt * Cc, Cc, Cc, transB=1, b_off=q0 * Cc * Cc,
                     gather=gt, cmap=cm, epilogue=ops.EPI_MUL_RELU_MASK, aux=y1, ldaux=Cc, tile=tile)
            q0 += nt
    t = graph_time(f, n=20)
    print("dX   tile %3d (4 launches): %7.1f us %6.1f TF/s" % (tile, t, fl / t / 1e6))
