#!/usr/bin/env python3
"""FFN up-projection / input-gradient GEMMs with their real epilogues, device time via graph replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import ops  # noqa: E402
from tools.gemm_probe4 import graph_time  # noqa: E402
DEV = "cuda"
espnet_amd.set_precision("bf16")
M, F, D = 7968, 2048, 256
x = torch.randn(M, D, device=DEV).to(torch.bfloat16)
W1 = torch.randn(F, D, device=DEV).to(torch.bfloat16)
b1 = torch.randn(F, device=DEV)
z = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
h = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
z32 = torch.empty(M, F, device=DEV)
dob = torch.randn(M, D, device=DEV).to(torch.bfloat16)
W2 = torch.randn(D, F, device=DEV).to(torch.bfloat16)
dz = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
tests = {
    "w1 -> z bf16": lambda: ops.linear_fwd(x, W1, b1, out=z),
    "w1 -> z fp32": lambda: ops.linear_fwd(x, W1, b1, out=z32),
    "w1 -> z bf16 + h=drop(swish) bf16": lambda: ops.linear_fwd(x, W1, b1, out=z, drop=(0.1, 3), Hb=h, h_act=ops.ACT_SWISH),
    "dz = dob W2 (bf16 out)": lambda: ops.linear_bwd_x(dob, W2, out=dz),
    "dz = dswish(z) * dob W2": lambda: ops.linear_bwd_x(dob, W2, out=dz, epilogue=ops.EPI_MUL_DSWISH, aux=z),
    "dz = drop(dswish(z) * dob W2)": lambda: ops.linear_bwd_x(dob, W2, out=dz, epilogue=ops.EPI_MUL_DSWISH, aux=z, drop=(0.1, 3)),
}
for k, f in tests.items():
    print("%-40s %7.1f us" % (k, graph_time(f)))
