#!/usr/bin/env python3
"""fused long-row attention (fp32) against the GEMM + softmax path: time per layer-sized call"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd
from espnet_amd import ops, functional as F_
DEV = torch.device("cuda")
espnet_amd.set_precision("fp32")
H, dk = 4, 64
D = H * dk
for B, T in ((16, 1000), (8, 1500), (4, 2048), (32, 640)):
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (0.5 * torch.randn(*s, generator=g)).to(DEV)
    qu, qv, k, v, p, dctx = r(B * T, D), r(B * T, D), r(B * T, D), r(B * T, D), r(T, D), r(B * T, D)
    lens = torch.linspace(T, T // 2, B).long()
    mask = (torch.arange(T)[None, :] < lens[:, None]).to(torch.uint8).view(B, 1, T).contiguous().to(DEV)
    res = {}
    for fused in (True, False):
        ops.F32_FUSED_ATTN = fused
        F_.FUSE_ATTN = fused
        def fwd():
            if fused:
                return F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T, T, H, dk)
            return F_.attn_fwd_unfused(qu, qv, k, v, p, mask, B, T, T, H, dk) if hasattr(F_, "attn_fwd_unfused") else None
        out = fwd()
        if out is None:
            continue
        P1 = out[0]
        def bwd():
            return F_.attn_core_bwd(dctx, P1, qu, qv, k, v, p, B, T, T, H, dk)
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            res[(fused, name)] = (time.perf_counter() - t0) / 5 * 1e6
    ops.F32_FUSED_ATTN = True; F_.FUSE_ATTN = True
    print("B=%d T=%d: " % (B, T) + "  ".join("%s %s %.0f us" % ("fused" if f else "gemm", n, us) for (f, n), us in res.items()))
