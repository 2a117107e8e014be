#!/usr/bin/env python3
"""Which torch ops (not espnet_amd kernels) run inside one training step: name, count, call stack tail."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import espnet_amd  # noqa: E402
from espnet_amd import ops, train  # noqa: E402
from espnet_amd.nets.e2e_asr_conformer import E2E  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

espnet_amd.set_precision("bf16")
torch.manual_seed(0)
model = E2E(80, 5000, bench.c2_args(0.1)).to("cuda").train()
model.sync_report = False
flat = train.FlatParams(model)
opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=256, warmup=25000, max_grad_norm=5.0)
xs, ilens, ys = bench.synth_batch(32, 1000, 100, 5000, seed=1)
batch = model.prepare(xs, ilens, ys)
for _ in range(2):
    train.train_step(model, flat, opt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train.train_step(model, flat, opt, batch)
    torch.cuda.synchronize()
from collections import Counter
cnt = Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.device_time_total > 0 and e.name in (
            "aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::clone",
            "aten::contiguous", "aten::_to_copy"):
        st = [s for s in (e.stack or []) if "espnet_amd" in s or "bench" in s]
        cnt[(e.name, st[0] if st else "?")] += 1
for (n, s), c in cnt.most_common(40):
    print("%4d %-16s %s" % (c, n, s[-110:]))
