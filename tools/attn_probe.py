#!/usr/bin/env python3
"""Fused attention forward (eamd_attn_fwd) vs score GEMMs + softmax + context GEMM at the config-2 shapes, graph-replay device time"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import functional as F_  # noqa: E402
from tools.gemm_probe4 import graph_time  # noqa: E402
DEV = "cuda"
espnet_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
H, dk = 4, 64
D = H * dk
for (B, T1, T2, rel, mk) in ((32, 249, 249, True, "len"), (32, 249, 249, False, "len"), (32, 101, 249, False, "len"), (32, 101, 101, False, "causal")):
    bf = lambda *s: torch.randn(*s, device=DEV).to(torch.bfloat16).to(espnet_amd.ops.act_dtype())
    if T1 == T2:
        qkv = bf(B * T1, 3 * D)
        k, v = F_._MV(qkv, D, 3 * D), F_._MV(qkv, 2 * D, 3 * D)
        qu = bf(B * T1, D) if rel else F_._MV(qkv, 0, 3 * D)
    else:
        qu, k, v = bf(B * T1, D), bf(B * T2, D), bf(B * T2, D)
    qv = bf(B * T1, D) if rel else None
    p = bf(T2, D) if rel else None
    if mk == "len":
        mask = torch.ones(B, 1, T2, dtype=torch.uint8, device=DEV)
    else:
        mask = torch.tril(torch.ones(T1, T2)).to(torch.uint8).expand(B, T1, T2).contiguous().to(DEV)
    tf = graph_time(lambda: F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T1, T2, H, dk), n=50)
    def unf():
        P = F_.attn_scores_fwd(qu, qv, k, p, mask, B, T1, T2, H, dk)
        F_.attn_context_fwd(P, v, B, T1, T2, H, dk)
    tu = graph_time(unf, n=50)
    print("B=%d T1=%d T2=%d rel=%d: fused %6.1f us   unfused %6.1f us" % (B, T1, T2, rel, tf, tu))
    # backward: whole attn_core_bwd with the query side fused / unfused
    P = F_.attn_scores_fwd(qu, qv, k, p, mask, B, T1, T2, H, dk)
    dctx = bf(B * T1, D)
    ts = []
    for fuse in (True, False):
        F_.FUSE_ATTN = fuse
        ts.append(graph_time(lambda: F_.attn_core_bwd(dctx, P, qu, qv, k, v, p, B, T1, T2, H, dk), n=30))
    F_.FUSE_ATTN = True
    espnet_amd.ops.FUSED_ATTN_KV = False
    tk = graph_time(lambda: F_.attn_core_bwd(dctx, P, qu, qv, k, v, p, B, T1, T2, H, dk), n=30)
    espnet_amd.ops.FUSED_ATTN_KV = True
    print("      backward (all products): fused %6.1f us   key side as GEMMs %6.1f us   unfused %6.1f us" % (ts[0], tk, ts[1]))
