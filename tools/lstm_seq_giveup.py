#!/usr/bin/env python3
"""the give-up path of the persistent LSTM kernels (diagnostic build, EAMD_LSTM_PROBE=16: workgroup 0 never publishes):
every wait must give up within its wall-clock bound, the launch must drain, the status word must be set and the outputs NaN"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa
from espnet_amd import ops
DEV = torch.device("cuda")
B, H, T = 32, 1024, 6
g = torch.Generator().manual_seed(1)
w = torch.randn(4 * H, H, generator=g).mul(0.03).to(DEV)
gx = torch.randn(T, B, 4 * H, generator=g).to(DEV)
h, c, y, acts = (torch.zeros(T, B, H, device=DEV), torch.zeros(T, B, H, device=DEV), torch.zeros(T, B, H, device=DEV), torch.zeros(T, B, 4 * H, device=DEV))
t0 = time.time()
ops.lstm_seq_fwd([(gx, w, None, None, h, c, y, acts, False)], T, B, H)
st = ops.lstm_seq_status()
print("forward: %.1f s, status 0x%x, NaN in y: %s" % (time.time() - t0, st, bool(torch.isnan(y).any())))
dg = torch.zeros(T, B, 4 * H, device=DEV)
t0 = time.time()
ops.lstm_seq_bwd([(torch.ones(T, B, H, device=DEV), w.t().contiguous(), acts, c, None, dg, False)], T, B, H)
st = ops.lstm_seq_status()
print("backward: %.1f s, status 0x%x, NaN in dgates: %s" % (time.time() - t0, st, bool(torch.isnan(dg).any())))
