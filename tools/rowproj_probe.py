#!/usr/bin/env python3
"""eamd_rowproj against the eamd_gemm launches it replaces, config-2 rows (M = 7968), fp32: us per launch, TFLOP/s.
Each case: 20 launches captured in one hipGraph, replayed 5 times between stream events (tools/gemm_f32_probe.py's method)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd
from espnet_amd import ops

espnet_amd.set_precision("fp32")
dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7968
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731


def timed(fn, reps=20, rounds=5):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    gph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)


def line(name, us, flop):
    print("%-58s %7.1f us  %6.1f TFLOP/s" % (name, us, flop / us / 1e6), flush=True)


gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
for K, N, what in ((256, 768, "QKV"), (256, 512, "pw_conv1"), (256, 256, "out-proj / pw_conv2")):
    W, x, b, R = rnd(N, K) / 16, rnd(M, K), rnd(N), rnd(M, N)
    img, = ops.rowproj_pack([(W, False)])
    flop = 2.0 * M * K * N
    line("gemm   y = x W^T + b            %-20s" % what, timed(lambda: ops.linear_fwd(x, W, b)), flop)
    line("rowproj                         %-20s" % what, timed(lambda: ops.rowproj(x, img, N, bias=b)), flop)
    xn, mean, rstd = torch.empty(M, K, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)
    line("layernorm_fwd + gemm            %-20s" % what, timed(lambda: ops.linear_fwd(ops.layernorm_fwd(x, gam, bet, 1e-12, torch.float32)[0], W, b)), flop)
    line("rowproj, LayerNorm in front     %-20s" % what, timed(lambda: ops.rowproj(xn, img, N, bias=b, ln=(x, gam, bet, 1e-12, mean, rstd))), flop)
    if N == 256:
        line("gemm + dropout + residual       %-20s" % what, timed(lambda: ops.linear_fwd(x, W, b, R=R, drop=(0.1, 5)) if ops.f32_epilogue_drop() else ops.linear_fwd(x, W, b, R=R)), flop)
        line("rowproj + dropout + residual    %-20s" % what, timed(lambda: ops.rowproj(x, img, N, bias=b, R=R, drop=(0.1, 5))), flop)
for K, what in ((768, "dxn = dqkv W3"), (512, "dxn = da W1"), (256, "dctx = dy Wo / de = dy W2")):
    W, dy, xin, dres = rnd(K, 256) / K ** 0.5, rnd(M, K), rnd(M, 256), rnd(M, 256)
    img, = ops.rowproj_pack([(W, True)])
    flop = 2.0 * M * K * 256
    line("gemm   dx = dy W                %-26s" % what, timed(lambda: ops.linear_bwd_x(dy, W)), flop)
    line("rowproj                         %-26s" % what, timed(lambda: ops.rowproj(dy, img, 256)), flop)
    _, mean, rstd = ops.layernorm_fwd(xin, gam, bet, 1e-12, torch.float32)
    dg, db = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    ws = ops.rowproj_lnb_ws(M, dev)
    dd = torch.empty(M, 256, device=dev)
    line("gemm + layernorm_bwd(+drop copy) %-26s" % what, timed(lambda: ops.layernorm_bwd(ops.linear_bwd_x(dy, W), xin, gam, mean, rstd, dres, dg, db, drop=(0.1, 9))), flop)
    line("rowproj, LayerNorm backward behind %-24s" % what, timed(lambda: ops.rowproj(dy, img, 256, lnb=(xin, gam, mean, rstd, dres, ws, dd, (0.1, 9)))), flop)
Ws = [rnd(768, 256), rnd(256, 256), rnd(512, 256), rnd(256, 256)]
jobs = [(w, t) for w in Ws for t in (False, True)]
print("pack of one layer's 8 images (one launch): %.1f us" % timed(lambda: ops.rowproj_pack(jobs)))
jobs12 = [(rnd(*w.shape), t) for _ in range(12) for w in Ws for t in (False, True)]
print("pack of 12 layers' 96 images (two launches): %.1f us" % timed(lambda: ops.rowproj_pack(jobs12), reps=5))
