#!/usr/bin/env python3
"""debug: grouped weight-gradient launch vs separate launches, per-parameter gradients after one step"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import espnet_amd
from espnet_amd import ops, train
from conftest import e2e_dk64_model, load_golden
DEV = "cuda"
g = load_golden("e2e_conformer_dk64.npz")
xs, ilens, ys = torch.from_numpy(g["xs"]).to(DEV), torch.from_numpy(g["ilens"]), torch.from_numpy(g["ys"]).to(DEV)
grads = {}
for mode in ("sep", "grp"):
    ops.GROUP_WGRAD = mode == "grp"
    model, _ = e2e_dk64_model(dropout=0.0)
    model = model.to(DEV).train()
    flat = train.FlatParams(model)
    opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=256, warmup=100, max_grad_norm=5.0)
    batch = model.prepare(xs, ilens, ys)
    train.train_step(model, flat, opt, batch)
    torch.cuda.synchronize()
    grads[mode] = {n: p._eamd_grad.clone() for n, p in model.named_parameters()}
bad = []
for n in grads["sep"]:
    a, b = grads["grp"][n], grads["sep"][n]
    rel = float((a - b).norm() / (b.norm() + 1e-30))
    if rel > 1e-5:
        bad.append((rel, n, tuple(a.shape), float(a.norm()), float(b.norm())))
for r in sorted(bad, reverse=True)[:40]:
    print("%.3e %-60s %s |grp| %.4g |sep| %.4g" % r)
print(len(bad), "of", len(grads["sep"]), "differ")
