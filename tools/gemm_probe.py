#!/usr/bin/env python3
"""Probe: leading-dimension padding / tile / split-K sensitivity of the bf16 GEMM (diagnostics)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402
from tools.gemm_bench import time_call  # noqa: E402

DEV = "cuda"


def run(name, m, n, k, pad_a=0, pad_b=0, tile=0, splitk=1, out_bf16=False):
    lda, ldb = k + pad_a, k + pad_b
    A = torch.randn(m, lda, device=DEV).to(torch.bfloat16)
    B = torch.randn(n, ldb, device=DEV).to(torch.bfloat16)
    C = torch.zeros(m, n, device=DEV, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    f = lambda: ops.gemm(A, B, C, m, n, k, lda, ldb, n, tile=tile, splitk=splitk)  # noqa: E731
    t = time_call(f)
    print(f"{name:34s} M={m} N={n} K={k} lda={lda} ldb={ldb} tile={tile} sk={splitk} bf16out={out_bf16}: {t:7.1f} us  {2.0*m*n*k/t/1e6:7.1f} TF/s")


def main():
    M = 7968
    x = torch.randn(64 * 1024 * 1024 // 4, device=DEV)
    y = torch.empty(x.numel(), device=DEV, dtype=torch.bfloat16)
    t = time_call(lambda: ops.cast_bf16(x, y))
    print(f"cast 64MB fp32->bf16: {t:.1f} us = {(x.numel()*6)/t/1e6:.2f} TB/s")
    for pad in (0, 8, 64):
        run("ffn_w2 pad A/B", M, 256, 2048, pad, pad)
    run("ffn_w2 tile128", M, 256, 2048, 0, 0, tile=128)
    run("ffn_w2 splitk4 (atomics)", M, 256, 2048, 0, 0, splitk=4)
    run("ffn_w2 pad64 splitk4", M, 256, 2048, 64, 64, splitk=4)
    for pad in (0, 8, 64):
        run("proj pad", M, 256, 256, pad, pad)
    for pad in (0, 64):
        run("ffn_w1 pad", M, 2048, 256, pad, pad)
    run("ffn_w1 bf16 out", M, 2048, 256, 0, 0, out_bf16=True)
    run("ffn_w1 bf16 out tile64", M, 2048, 256, 0, 0, tile=64, out_bf16=True)
    run("big square", 4096, 4096, 4096)
    run("big square bf16out", 4096, 4096, 4096, out_bf16=True)


if __name__ == "__main__":
    main()
