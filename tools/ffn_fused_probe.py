#!/usr/bin/env python3
"""Fused FFN launches (eamd_ffn_fwd / eamd_ffn_bwd) against the GEMM pairs they replace, config-2 shapes (M = 7968,
D = 256, F = 2048), hipGraph replay device time.  usage: ffn_fused_probe.py [fp32|bf16]"""
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
    M, D, F = int(os.environ.get("FFN_PROBE_M", "7968")), 256, 2048
    dev = "cuda"
    x = torch.randn(M, D, device=dev).to(dt)
    w1 = (torch.randn(F, D, device=dev) * 0.05).to(dt)
    b1 = torch.zeros(F, device=dev)
    w2 = (torch.randn(D, F, device=dev) * 0.05).to(dt)
    b2 = torch.zeros(D, device=dev)
    R = torch.randn(M, D, device=dev)
    dy = torch.randn(M, D, device=dev).to(dt)
    ops.manual_seed(1)
    drop = (0.1, 11, 0.1, 12)
    gf = 4.0 * M * D * F / 1e6
    out, f, h = ops.ffn_fwd(x, w1, b1, w2, b2, act=ops.ACT_SWISH, alpha=0.5, R=R, drop=drop)
    packs = ops.ffn_pack(w1, w2)

    def pair_fwd():
        hh = torch.empty(M, F, device=dev, dtype=dt)
        z = ops.linear_fwd(x, w1, b1, out_dtype=dt, drop=(0.1, 11), Hb=hh, h_act=ops.ACT_SWISH, act=ops.EPI_DACT_FACTOR)
        return ops.linear_fwd(hh, w2, b2, R=R, alpha=0.5, drop=(0.1, 12))

    def pair_bwd():
        dz = ops.linear_bwd_x(dy, w2, epilogue=ops.EPI_MUL_AUX, aux=f, alpha=0.5, out_dtype=dt)
        return ops.linear_bwd_x(dz, w1)

    kw, kf = dict(packed=packs[2:]), dict(packed=packs[:2])
    rows = [("fused fwd (save)", lambda: ops.ffn_fwd(x, w1, b1, w2, b2, act=ops.ACT_SWISH, alpha=0.5, R=R, drop=drop, **kf)),
            ("fused fwd (no save)", lambda: ops.ffn_fwd(x, w1, b1, w2, b2, act=ops.ACT_SWISH, alpha=0.5, R=R, drop=drop, save=False, **kf)),
            ("pack (4 images)", lambda: ops.ffn_pack(w1, w2)),
            ("pair  fwd", pair_fwd),
            ("fused bwd", lambda: ops.ffn_bwd(dy, w1, w2, f, alpha=0.5, **kw)),
            ("pair  bwd", pair_bwd)]
    for rep in range(2):
        for name, fn in rows:
            fn()
            us = graph_time(fn, n=20)
            print("%s %-22s %7.1f us  %7.1f TF" % (prec, name, us, gf / us), flush=True)


if __name__ == "__main__":
    main()
