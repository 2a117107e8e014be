#!/usr/bin/env python3
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops  # noqa: E402
from tools.gemm_bench import time_call  # noqa: E402
DEV = "cuda"

def run(name, m, n, k, ta, tb, tile, sk, out_bf16=False, aux=False, colsum=False):
    A = torch.randn((k, m) if ta else (m, k), device=DEV).to(torch.bfloat16)
    B = torch.randn((k, n) if tb else (n, k), device=DEV).to(torch.bfloat16)
    C = torch.zeros(m, n, device=DEV, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    ax = torch.randn(m, n, device=DEV).to(torch.bfloat16) if aux else None
    cs = torch.zeros(m, device=DEV) if colsum else None
    f = lambda: ops.gemm(A, B, C, m, n, k, m if ta else k, n if tb else k, n, transA=ta, transB=tb, tile=tile, splitk=sk,
                         aux=ax, ldaux=n, epilogue=4 if aux else 0, colsum=cs)
    t = time_call(f)
    print(f"{name:28s} {m}x{n}x{k} ta{ta} tb{tb} tile={tile} sk={sk} bf16out={out_bf16} aux={aux} cs={colsum}: {t:7.1f} us {2.0*m*n*k/t/1e6:7.1f} TF/s")

M = 7968
for sk in (2, 4, 8, 16):
    run("dW1 TN +colsum", 2048, 256, M, 1, 1, 64, sk, colsum=True)
for sk in (2, 4, 8, 16):
    run("dW2 TN +colsum", 256, 2048, M, 1, 1, 64, sk, colsum=True)
for sk in (4, 8, 16, 24, 32, 48):
    run("dWproj TN +colsum", 256, 256, M, 1, 1, 64, sk, colsum=True)
for sk in (4, 8, 16, 32):
    run("dWpw1 TN +colsum", 512, 256, M, 1, 1, 64, sk, colsum=True)
for tile in (64, 128):
    run("dz NN bf16out+aux", M, 2048, 256, 0, 1, tile, 1, True, True)
    run("dz NN bf16out", M, 2048, 256, 0, 1, tile, 1, True, False)
    run("ffn_w1 NT bf16out", M, 2048, 256, 0, 0, tile, 1, True)
    run("ffn_w2 NT", M, 256, 2048, 0, 0, tile, 1)
    run("dxn NN", M, 256, 2048, 0, 1, tile, 1)
    run("proj NT", M, 256, 256, 0, 0, tile, 1)
    run("dproj NN", M, 256, 256, 0, 1, tile, 1)
    run("pw1 NT", M, 512, 256, 0, 0, tile, 1)
    run("ctc_lo NT", M, 5000, 256, 0, 0, tile, 1)
    run("dlogits NN", M, 256, 5000, 0, 1, tile, 1)
