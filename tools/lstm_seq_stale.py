#!/usr/bin/env python3
"""Stale-read detector for the persistent LSTM kernels: the same output buffers, different inputs from launch to launch
(what a graph replay with a moving dropout mask does), every launch compared with the per-step kernels."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa
from espnet_amd import ops, rnn_functional as R
DEV = torch.device("cuda")
B, H, T, ndir = 32, 1024, 60, 2
g = torch.Generator().manual_seed(1)
ws = [(torch.randn(4 * H, H, generator=g).mul(1.0 / H ** 0.5).to(DEV), torch.randn(4 * H, generator=g).mul(0.1).to(DEV)) for _ in range(ndir)]
bufs = [[torch.empty(T, B, H, device=DEV), torch.empty(T, B, H, device=DEV), torch.empty(T, B, H, device=DEV), torch.empty(T, B, 4 * H, device=DEV)] for _ in range(ndir)]
dgs = [torch.empty(T, B, 4 * H, device=DEV) for _ in range(ndir)]
wts = [w.t().contiguous() for w, _ in ws]
bad = 0
for it in range(6):
    gxs = [torch.randn(T, B, 4 * H, generator=g).to(DEV) for _ in range(ndir)]
    dys = [torch.randn(T, B, H, generator=g).to(DEV) for _ in range(ndir)]
    ops.lstm_seq_fwd([(gxs[i], ws[i][0], ws[i][1], None, bufs[i][0], bufs[i][1], bufs[i][2], bufs[i][3], i == 1) for i in range(ndir)], T, B, H)
    st = ops.lstm_seq_status()
    ops.lstm_seq_bwd([(dys[i], wts[i], bufs[i][3], bufs[i][1], None, dgs[i], i == 1) for i in range(ndir)], T, B, H)
    st2 = ops.lstm_seq_status()
    for i in range(ndir):
        gx = gxs[i].clone().requires_grad_(True)
        ops.LSTM_PERSISTENT = False
        y2 = R.LSTMSeqFn.apply(gx, ws[i][0], ws[i][1], None, i == 1)
        y2.backward(dys[i])
        e = float((bufs[i][2] - y2).abs().max()); e2 = float((dgs[i] - gx.grad).abs().max())
        bad += (e > 1e-5) + (e2 > 1e-4)
        print("launch %d dir %d: status %d/%d  y max err %.2e  dgates max err %.2e" % (it, i, st, st2, e, e2))
# timing
for name, fn in (("fwd", lambda: ops.lstm_seq_fwd([(gxs[i], ws[i][0], ws[i][1], None, bufs[i][0], bufs[i][1], bufs[i][2], bufs[i][3], i == 1) for i in range(ndir)], T, B, H)),
                 ("bwd", lambda: ops.lstm_seq_bwd([(dys[i], wts[i], bufs[i][3], bufs[i][1], None, dgs[i], i == 1) for i in range(ndir)], T, B, H))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print("%s: %.2f us per time step (both directions side by side, T = %d)" % (name, (time.perf_counter() - t0) / 5 / T * 1e6, T))
print("STALE" if bad else "CLEAN")
