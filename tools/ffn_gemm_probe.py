#!/usr/bin/env python3
"""What bounds the two epilogue-heavy FFN GEMMs (config-2 shapes, M=7968 N=2048 K=256): the same product timed with and
without its epilogue work and at both tile sizes, graph-replay device time.  usage: ffn_gemm_probe.py [bf16|fp32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402
from espnet_amd import ops  # noqa: E402
from tools.gemm_probe4 import graph_time  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    espnet_amd.set_precision(prec)
    dt = ops.act_dtype()
    M, N, K = 7968, 2048, 256
    dev = "cuda"
    x = torch.randn(M, K, device=dev).to(dt)
    w1 = (torch.randn(N, K, device=dev) * 0.05).to(dt)          # nn.Linear layout [out][in]
    dy = torch.randn(M, K, device=dev).to(dt)
    w2 = (torch.randn(K, N, device=dev) * 0.05).to(dt)          # [out=256][in=2048]
    zf = torch.randn(M, N, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    out2 = torch.empty(M, N, device=dev, dtype=dt)
    bias = torch.zeros(N, device=dev)
    ops.manual_seed(1)
    gf = 2.0 * M * N * K / 1e3
    for tile in (64, 128):
        rows = []
        # up-projection x W1^T: plain / + swish / + factor + dual output + dropout (what the model runs)
        rows.append(("up   plain", lambda: ops.gemm(x, w1, out, M, N, K, K, K, N, bias=bias, tile=tile)))
        rows.append(("up   swish", lambda: ops.gemm(x, w1, out, M, N, K, K, K, N, bias=bias, epilogue=ops.EPI_SWISH, tile=tile)))
        rows.append(("up   swish+drop", lambda: ops.gemm(x, w1, out, M, N, K, K, K, N, bias=bias, epilogue=ops.EPI_SWISH,
                                                        drop=(0.1, 77), tile=tile)))
        rows.append(("up   z+h+drop", lambda: ops.gemm(x, w1, out, M, N, K, K, K, N, bias=bias, Hb=out2, h_act=ops.ACT_SWISH,
                                                      drop=(0.1, 77), tile=tile)))
        rows.append(("up   factor+h+drop", lambda: ops.gemm(x, w1, out, M, N, K, K, K, N, bias=bias, epilogue=ops.EPI_DACT_FACTOR,
                                                           Hb=out2, h_act=ops.ACT_SWISH, drop=(0.1, 77), tile=tile)))
        # dz = dy W2 (.) zf      (W2: [256][2048])
        rows.append(("dz   plain", lambda: ops.gemm(dy, w2, out, M, N, K, K, N, N, transB=1, tile=tile)))
        rows.append(("dz   mul aux", lambda: ops.gemm(dy, w2, out, M, N, K, K, N, N, transB=1, epilogue=ops.EPI_MUL_AUX, aux=zf,
                                                     ldaux=N, tile=tile)))
        for name, f in rows:
            try:
                f()
                us = graph_time(f, n=20)
                print("%s tile %3d  %-22s %6.1f us  %6.1f TF" % (prec, tile, name, us, gf / us))
            except Exception as e:   # noqa: BLE001
                print("%s tile %3d  %-22s declined: %s" % (prec, tile, name, str(e)[:80]))


if __name__ == "__main__":
    main()
