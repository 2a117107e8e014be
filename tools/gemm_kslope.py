#!/usr/bin/env python3
"""Main-loop rate vs fixed cost of the fp32 / bf16 GEMM tiles: M = N = 4096 (1024 tiles of 128x128) at growing K;
the slope of time over K is the main loop, the intercept the prologue + epilogue + launch.  usage: gemm_kslope.py [fp32|bf16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import ops  # noqa: E402
from tools.gemm_probe4 import graph_time  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    espnet_amd.set_precision(prec)
    dt = ops.act_dtype()
    M = N = 4096
    for tile in (64, 128):
        prev = None
        for K in (256, 512, 1024, 2048, 4096, 8192):
            a = torch.randn(M, K, device="cuda").to(dt)
            b = torch.randn(N, K, device="cuda").to(dt)
            c = torch.empty(M, N, device="cuda", dtype=torch.float32)
            f = lambda: ops.gemm(a, b, c, M, N, K, K, K, N, tile=tile)
            f()
            us = graph_time(f, n=10)
            tf = 2.0 * M * N * K / us / 1e6
            slope = "" if prev is None else "   marginal %.1f TF" % (2.0 * M * N * (K - prev[0]) / (us - prev[1]) / 1e6)
            print("%s tile %3d K %5d: %8.1f us  %6.1f TF%s" % (prec, tile, K, us, tf, slope))
            prev = (K, us)


if __name__ == "__main__":
    main()
