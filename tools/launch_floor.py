#!/usr/bin/env python3
"""Per-node cost of dependent kernel launches inside a replayed hipGraph vs eager stream order."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402
DEV = "cuda"
x = torch.randn(64, 256, device=DEV)
y = torch.empty_like(x)
big = torch.randn(7968, 256, device=DEV)
bigy = torch.empty_like(big)
N = 1000


def body(kind):
    for _ in range(N):
        if kind == "tiny":
            ops.axpby(x, None, 1.0, 0.0, out=y)
        elif kind == "rows8k":
            ops.axpby(big, None, 1.0, 0.0, out=bigy)


for kind in ("tiny", "rows8k"):
    body(kind); torch.cuda.synchronize()
    t0 = time.perf_counter(); body(kind); torch.cuda.synchronize(); t1 = time.perf_counter()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body(kind)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            body(kind)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("%-8s eager %.2f us/launch   graph %.2f us/launch" % (kind, (t1 - t0) / N * 1e6, (t3 - t2) / 5 / N * 1e6))
