#!/usr/bin/env python3
"""depthwise conv kernels at the headline shape (B=32, T=249, C=256, K=31), graph-replay device time"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402
DEV = "cuda"


def graph_time(f, n=100):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                f()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 / n * 1e6


def main():
    x = torch.randn(B, T, C, device=DEV)
    dy = torch.randn(B, T, C, device=DEV)
    w = torch.randn(C, K, device=DEV)
    b = torch.randn(C, device=DEV)
    dw = torch.zeros(C, K, device=DEV)
    db = torch.zeros(C, device=DEV)
    print("fwd   %6.1f us" % graph_time(lambda: ops.dwconv_fwd(x, w, b, B, T, C, K)))
    print("bwd_x %6.1f us" % graph_time(lambda: ops.dwconv_bwd_x(dy, w, B, T, C, K)))
    print("bwd_w %6.1f us  (EAMD_DWW_TPB=%s EAMD_DWW_MODE=%s)" % (graph_time(lambda: ops.dwconv_bwd_w(dy, x, dw, db, B, T, C, K)),
                                                                os.environ.get("EAMD_DWW_TPB"), os.environ.get("EAMD_DWW_MODE")))


if __name__ == "__main__":
    main()
