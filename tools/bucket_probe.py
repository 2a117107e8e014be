#!/usr/bin/env python3
"""Replay cost of train.BucketedGraphStep - one repeated shape, then alternating shapes"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, espnet_amd
from espnet_amd import ops, train, functional as F_
from espnet_amd.nets.e2e_asr_conformer import E2E
dev = torch.device("cuda", 0)
espnet_amd.set_precision("fp32")
for k in sys.argv[1:]:
    if k == "noshare": F_.SHARE_PROJ = False
    if k == "nogroup": ops.GROUP_WGRAD = False
    if k == "noattn": ops.F32_FUSED_ATTN = False
B, L, V = 32, 100, 5000
g = torch.Generator().manual_seed(17)
def batch(T):
    xs = torch.randn(B, T, 80, generator=g)
    ilens = [int(round(v)) for v in torch.linspace(T, 0.6 * T, B).tolist()]
    return xs.to(dev), ilens, torch.randint(1, V - 1, (B, L), generator=g).to(dev)
torch.manual_seed(0)
model = E2E(80, V, bench.c2_args(0.1)).to(dev).train()
model.sync_report = False
flat = train.FlatParams(model)
opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=256, warmup=25000, max_grad_norm=5.0)
step = train.BucketedGraphStep(model, flat, opt, t_edge=64, l_edge=8, max_graphs=8)
shapes = [batch(T) for T in (1000, 940, 880, 810, 750, 690, 620)]
for b in shapes * 3:
    step(*b)
torch.cuda.synchronize()
for name, seq in (("same shape", [shapes[0]] * 12), ("alternating", shapes * 4)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in seq:
        step(*b)
    torch.cuda.synchronize()
    print(name, "%.2f ms/step" % ((time.perf_counter() - t0) / len(seq) * 1e3), step.stats())
# host cost of one call without waiting for the GPU
t0 = time.perf_counter(); step(*shapes[0]); t1 = time.perf_counter(); torch.cuda.synchronize()
print("host time of one call %.2f ms" % ((t1 - t0) * 1e3))
