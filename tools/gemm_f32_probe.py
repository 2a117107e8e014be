#!/usr/bin/env python3
"""fp32-MFMA GEMM probe: the dense shapes of a config-2 training step at both tile sizes (and a split-K sweep for the
weight gradients), each timed by hipGraph replay of 10 launches.  Guides eamd_gemm's tile / split-K choice in fp32 mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402

DEV = "cuda"
M = 7968
SHAPES = [  # name, M, N, K, transA, transB, splitks
    ("ffn_w1  NT", M, 2048, 256, 0, 0, (1,)), ("ffn_w2  NT", M, 256, 2048, 0, 0, (1, 2)), ("proj    NT", M, 256, 256, 0, 0, (1,)),
    ("dz      NN", M, 2048, 256, 0, 1, (1,)), ("dxn     NN", M, 256, 2048, 0, 1, (1, 2)), ("dproj   NN", M, 256, 256, 0, 1, (1,)),
    ("dW1     TN", 2048, 256, M, 1, 1, (1, 2, 4, 8)), ("dW2     TN", 256, 2048, M, 1, 1, (1, 2, 4, 8)),
    ("dWproj  TN", 256, 256, M, 1, 1, (4, 8, 16, 32)), ("pw1     NT", M, 512, 256, 0, 0, (1,)),
    ("dec ffn NT", 3232, 2048, 256, 0, 0, (1,)), ("dec prj NT", 3232, 256, 256, 0, 0, (1,)),
    ("qkv     NT", M, 768, 256, 0, 0, (1,)), ("dqkv    NN", M, 256, 768, 0, 1, (1, 2)), ("dpw1    NN", M, 256, 512, 0, 1, (1, 2)),
    ("dec out NT", 3232, 5000, 256, 0, 0, (1,)), ("dec dout NN", 3232, 256, 5000, 0, 1, (1, 2, 4)),
    ("ctc_lo  NT", M, 5000, 256, 0, 0, (1,)), ("embed   NT", M, 256, 4864, 0, 0, (1, 2, 4)),
]


def timed(fn, reps=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


def main():
    ops.set_precision("fp32")
    g = torch.Generator().manual_seed(0)
    print("%-12s %6s %6s %6s | tile splitk %9s %7s" % ("shape", "M", "N", "K", "us", "TF/s"))
    for name, m, n, k, ta, tb, sks in SHAPES:
        A = torch.randn((k, m) if ta else (m, k), generator=g).to(DEV)
        B = torch.randn((k, n) if tb else (n, k), generator=g).to(DEV)
        C = torch.zeros(m, n, device=DEV)
        lda, ldb = (m if ta else k), (n if tb else k)
        for tile in (64, 128):
            for sk in sks:
                t = timed(lambda: ops.gemm(A, B, C, m, n, k, lda, ldb, n, transA=ta, transB=tb, splitk=sk, tile=tile, precision=0))
                print("%-12s %6d %6d %6d | %4d %6d %9.1f %7.1f" % (name, m, n, k, tile, sk, t, 2.0 * m * n * k / t / 1e6))
    Bb, T, H, dk = 32, 249, 4, 64
    D, ldp = H * dk, 256
    q = torch.randn(Bb, T, D, generator=g).to(DEV)
    sc = torch.zeros(H * Bb * T * ldp, device=DEV)
    ctxv = torch.zeros(Bb * T, D, device=DEV)
    for tile in (64, 128):
        t = timed(lambda: ops.gemm(q, q, sc, T, T, dk, D, D, ldp, batch=(Bb, H), sA=(T * D, dk), sB=(T * D, dk),
                                   sC=(T * ldp, Bb * T * ldp), tile=tile, precision=0))
        print("scores  249x249x64 x128 | %4d      1 %9.1f %7.1f" % (tile, t, 2.0 * T * T * dk * Bb * H / t / 1e6))
        t = timed(lambda: ops.gemm(sc, q, ctxv, T, dk, T, ldp, D, D, transB=1, batch=(Bb, H), sA=(T * ldp, Bb * T * ldp),
                                   sB=(T * D, dk), sC=(T * D, dk), tile=tile, precision=0))
        print("context 249x64x249 x128 | %4d      1 %9.1f %7.1f" % (tile, t, 2.0 * T * T * dk * Bb * H / t / 1e6))


if __name__ == "__main__":
    main()
