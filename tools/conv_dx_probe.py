#!/usr/bin/env python3
"""Input gradient of Conv2dSubsampling's second convolution at config 2 (B=32, T=1000): the four stride-parity products
issued one by one against eamd_gemm_multi.  usage: python tools/conv_dx_probe.py [fp32|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import functional as Fn, ops  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    espnet_amd.set_precision(prec)
    adt = torch.bfloat16 if prec == "bf16" else torch.float32
    dev = "cuda"
    B, Hi, Wi, Cc = 32, 499, 39, 256
    Ho, Wo = (Hi - 3) // 2 + 1, (Wi - 3) // 2 + 1
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(Cc, Cc, 3, 3, generator=g) / 48.0).to(dev)
    y_in = torch.relu(torch.randn(B * Hi * Wi, Cc, generator=g)).to(dev).to(adt)
    dy = torch.randn(B * Ho * Wo, Cc, generator=g).to(dev).to(adt)
    _wf, wd = ops.conv2_weight_prep(w, adt)
    flop = 2.0 * B * Ho * Wo * Cc * Cc * 9
    for per_call in (1, 4, 2):
        ops.GEMM_MULTI_MAX = per_call
        rec = []
        ops._gemm_record = rec
        dw, db = torch.zeros(Cc, Cc, 3, 3, device=dev), torch.zeros(Cc, device=dev)
        Fn._conv3s2_bwd(dy, y_in, wd, dw, db, B, Hi, Wi, Ho, Wo, Cc, adt)
        ops._gemm_record = None
        rec = rec[1:]                      # [0] is the weight gradient
        torch.cuda.synchronize()
        sp = ops.stream_ptr()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            for _p, _k, replay in rec:
                replay(sp)
        e0.record()
        for _ in range(20):
            for _p, _k, replay in rec:
                replay(sp)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%s: %d products per call, %d launches: %.1f us = %.1f TFLOP/s" % (prec, per_call, len(rec), us, flop / us / 1e6))


if __name__ == "__main__":
    main()
