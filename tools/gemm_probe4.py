#!/usr/bin/env python3
"""Scaling of the skinny projection GEMMs with M (latency-bound vs throughput-bound), device time via graph replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402
DEV = "cuda"


def graph_time(f, n=200):
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
    for (N, K) in ((256, 256), (768, 256), (256, 2048), (2048, 256)):
        for M in (996, 1992, 3984, 7968, 15936, 31872):
            A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
            W = torch.randn(N, K, device=DEV).to(torch.bfloat16)
            C = torch.empty(M, N, device=DEV)
            R = torch.randn(M, N, device=DEV)
            b = torch.randn(N, device=DEV)
            t = graph_time(lambda: ops.linear_fwd(A, W, b, out=C, R=R))
            t2 = graph_time(lambda: ops.linear_fwd(A, W, b, out=C))
            print("NT M=%6d N=%4d K=%4d: %6.1f us (+R) %6.1f us   %6.1f TF/s  blocks %d" %
                  (M, N, K, t, t2, 2.0 * M * N * K / t2 / 1e6, ((M + 63) // 64) * ((N + 63) // 64)))


if __name__ == "__main__":
    main()
