#!/usr/bin/env python3
"""LSTM sequence (T steps, B x H) forward + backward through the one-launch step kernels, hipGraph replay.
(Round 2 tried a 2-D split of the backward step - unit blocks x reduction slices over all CUs, f32 atomics into a zeroed
accumulator, last-arriver cell backward: 11.4-11.8 ms per sequence against 10.3 ms for the one-dimensional kernel; dropped.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import ops, rnn_functional as R
DEV = "cuda"
T, B, H = 250, 32, 1024
ops.set_precision("fp32")
gx0 = torch.randn(T, B, 4 * H, device=DEV) * 0.1
w_hh = (torch.randn(4 * H, H, device=DEV) * 0.02).requires_grad_(True)
b_hh = torch.zeros(4 * H, device=DEV, requires_grad=True)
gy = torch.randn(T, B, H, device=DEV)
for split in (False, False):
    def run():
        gx = gx0.clone().requires_grad_(True)
        y = R.LSTMSeqFn.apply(gx, w_hh, b_hh, None, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y.backward(gy)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run()
    g = torch.cuda.CUDAGraph()
    gx = gx0.clone().requires_grad_(True)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = R.LSTMSeqFn.apply(gx, w_hh, b_hh, None, False); y.backward(gy)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            y = R.LSTMSeqFn.apply(gx, w_hh, b_hh, None, False); y.backward(gy)
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    print("split %s: fwd+bwd graph replay %.2f ms (T=%d)" % (split, (time.perf_counter() - t0) / 3 * 1e3, T))
