#!/usr/bin/env python3
"""conv1 (C_in = 1, 3x3 stride 2) kernels of Conv2dSubsampling at the headline shape, graph-replay device time"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402
from tools.dwconv_probe import graph_time  # noqa: E402
DEV = "cuda"
B, T, F, C = 32, 1000, 80, 256
H, W = (T - 3) // 2 + 1, (F - 3) // 2 + 1
x = torch.randn(B, T, F, device=DEV)
w = torch.randn(C, 9, device=DEV)
b = torch.randn(C, device=DEV)
dw = torch.zeros(C, 9, device=DEV)
db = torch.zeros(C, device=DEV)
for dt in (torch.bfloat16, torch.float32):
    dy = torch.randn(B, H, W, C, device=DEV).to(dt)
    t = graph_time(lambda: ops.conv1_fwd(x, w, b, B, T, F, C, dt), n=20)
    print("fwd   %s %7.1f us  (%.2f TB/s written)" % (dt, t, dy.numel() * dy.element_size() / t / 1e6))
    t = graph_time(lambda: ops.conv1_bwd_w(dy, x, dw, db, B, T, F, C), n=20)
    print("bwd_w %s %7.1f us  (%.2f TB/s read)  EAMD_C1W_BLOCKS=%s" % (dt, t, dy.numel() * dy.element_size() / t / 1e6,
                                                                    os.environ.get("EAMD_C1W_BLOCKS")))
