#!/usr/bin/env python3
"""long-row fused attention (fp32, legacy rel_shift, d_k 64, 4 heads): time of the forward and the query-side backward per call, as
a function of the queries per workgroup (EAMD_ATTN_LONG_LQ = 16 / 32 read at the first call of a process: run once per setting).
FLOP counted: forward 3 products (ac, bd, P V), backward (query side) 2 (dP, dq) of 2 B H T^2 64 each."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import functional as F_, ops  # noqa: E402

DEV = torch.device("cuda")
espnet_amd.set_precision("fp32")
H, dk = 4, 64
D = H * dk
print("EAMD_ATTN_LONG_LQ =", os.environ.get("EAMD_ATTN_LONG_LQ", "(default: 32 up to 1132 keys)"))
for B, T in ((16, 530), (16, 640), (16, 874), (16, 960), (16, 1000), (8, 1500), (4, 2048), (4, 3000)):
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (0.5 * torch.randn(*s, generator=g)).to(DEV)  # noqa: E731
    qu, qv, k, v, p, dctx = r(B * T, D), r(B * T, D), r(B * T, D), r(B * T, D), r(T, D), r(B * T, D)
    mask = torch.ones(B, 1, T, dtype=torch.uint8, device=DEV)
    P1, _, _ = F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T, T, H, dk)
    ldp = F_._ldp(T)
    dS, dbd = torch.empty_like(P1), torch.empty_like(P1)
    dq = torch.empty(B * T, D, device=DEV)

    def fwd():
        return F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T, T, H, dk)

    def bwd():
        return ops.attn_bwd_q((dctx, 0, D), (k, 0, D), (v, 0, D), P1, dS, dbd, (dq, 0, D), B, T, T, H, dk, ldp, 0.125)
    out = {}
    for name, fn, nprod in (("fwd", fwd, 3), ("bwd_q", bwd, 2)):
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            out[name] = "n/a (%s)" % str(e)[:60]
            continue
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            for _ in range(10):
                fn()
        g_.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g_.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 50 * 1e6
        out[name] = "%.0f us = %.0f TFLOP/s" % (us, nprod * 2.0 * B * H * T * T * dk / us / 1e6)
    print("B=%2d T'=%4d: %s" % (B, T, "  ".join("%s %s" % kv for kv in out.items())), flush=True)
