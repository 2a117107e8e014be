#!/usr/bin/env python3
"""config 4 at full size: eager loss, then graph replays; prints the losses and the persistent kernels' status word"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import espnet_amd
from espnet_amd import ops, train
from espnet_amd.nets.e2e_asr import E2E
from test_gpu_rnn import _config4_args
DEV = torch.device("cuda")
espnet_amd.set_precision("fp32")
torch.manual_seed(0)
V = 5000
m = E2E(80, V, _config4_args(1024, 1024, 1024)).to(DEV).train()
flat = train.FlatParams(m)
g = torch.Generator().manual_seed(0)
B, T, L = 32, 1000, 100
xs = torch.randn(B, T, 80, generator=g).to(DEV)
ilens = [T - 7 * i for i in range(B)]
ys = torch.randint(1, V - 1, (B, L), generator=g)
def step():
    flat.zero_grad()
    loss = m(xs, ilens, ys)
    ops.wgrad_group_begin()
    try:
        loss.backward()
    finally:
        ops.wgrad_group_end()
    return loss
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    e1 = float(step()); s1 = ops.lstm_seq_status()
    e2 = float(step()); s2 = ops.lstm_seq_status()
    torch.cuda.synchronize()
torch.cuda.current_stream().wait_stream(side)
print("eager %.6f %.6f status %d %d" % (e1, e2, s1, s2))
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    loss = step()
for i in range(4):
    gr.replay()
    torch.cuda.synchronize()
    print("replay %d: %.6f  grad sumsq %.6e status %d" % (i, float(loss), float(flat.grad.double().pow(2).sum()), ops.lstm_seq_status()))
