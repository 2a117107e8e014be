#!/usr/bin/env python3
"""Micro-benchmark of the GEMM shapes of BASELINE config 2 (one process, interleaved rounds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402

DEV = "cuda"
M = 7968
SHAPES = [  # name, M, N, K, transA, transB, splitk
    ("ffn_w1  NT", M, 2048, 256, 0, 0, 1), ("ffn_w2  NT", M, 256, 2048, 0, 0, 1), ("proj    NT", M, 256, 256, 0, 0, 1),
    ("dz      NN", M, 2048, 256, 0, 1, 1), ("dxn     NN", M, 256, 2048, 0, 1, 1), ("dproj   NN", M, 256, 256, 0, 1, 1),
    ("dW1     TN", 2048, 256, M, 1, 1, 0), ("dW2     TN", 256, 2048, M, 1, 1, 0), ("dWproj  TN", 256, 256, M, 1, 1, 0),
    ("ctc_lo  NT", M, 5000, 256, 0, 0, 1), ("dec_out NT", 3232, 5000, 256, 0, 0, 1),
    ("embed   NT", M, 256, 4864, 0, 0, 1),
]


def time_call(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    print(f"{'shape':12s} {'M':>6} {'N':>6} {'K':>6} | {'fp32in/bf16 us':>15} {'TF/s':>7} | {'bf16in us':>10} {'TF/s':>7} {'GB/s':>7}")
    for name, m, n, k, ta, tb, sk in SHAPES:
        A32 = torch.randn((k, m) if ta else (m, k), generator=g).to(DEV)
        B32 = torch.randn((k, n) if tb else (n, k), generator=g).to(DEV)
        A16, B16 = A32.to(torch.bfloat16), B32.to(torch.bfloat16)
        C = torch.zeros(m, n, device=DEV)
        skk = ops.auto_splitk(m, n, k) if sk == 0 else sk
        lda, ldb = (m if ta else k), (n if tb else k)
        f32 = lambda: ops.gemm(A32, B32, C, m, n, k, lda, ldb, n, transA=ta, transB=tb, splitk=skk, precision=1)  # noqa: E731
        f16 = lambda: ops.gemm(A16, B16, C, m, n, k, lda, ldb, n, transA=ta, transB=tb, splitk=skk)  # noqa: E731
        t32, t16 = time_call(f32), time_call(f16)
        fl = 2.0 * m * n * k
        by = 2.0 * (m * k + n * k) + 4.0 * m * n
        print(f"{name:12s} {m:6d} {n:6d} {k:6d} | {t32:15.1f} {fl / t32 / 1e6:7.1f} | {t16:10.1f} {fl / t16 / 1e6:7.1f} {by / t16 / 1e3:7.0f}  splitk={skk}")
    # batched attention products
    Bb, T, H, dk = 32, 249, 4, 64
    D = H * dk
    ldp = 256
    q = torch.randn(Bb, T, D, generator=g).to(DEV)
    sc = torch.zeros(H * Bb * T * ldp, device=DEV)
    q16 = q.to(torch.bfloat16)
    f32 = lambda: ops.gemm(q, q, sc, T, T, dk, D, D, ldp, batch=(Bb, H), sA=(T * D, dk), sB=(T * D, dk), sC=(T * ldp, Bb * T * ldp), precision=1)  # noqa: E731
    f16 = lambda: ops.gemm(q16, q16, sc, T, T, dk, D, D, ldp, batch=(Bb, H), sA=(T * D, dk), sB=(T * D, dk), sC=(T * ldp, Bb * T * ldp))  # noqa: E731
    print(f"scores batched: fp32in {time_call(f32):.1f} us, bf16in {time_call(f16):.1f} us")


if __name__ == "__main__":
    main()
