#!/usr/bin/env python3
"""Cost of the data-parallel step structure at N=1: single graph vs GraphedDataParallelStep without a process group
(graph split only) vs with a one-rank RCCL group (graph split + collectives)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import espnet_amd  # noqa: E402
from espnet_amd import ops, train  # noqa: E402
from espnet_amd.nets.e2e_asr_conformer import E2E  # noqa: E402

dev = torch.device("cuda", 0)
espnet_amd.set_precision("bf16")
B, T, L, V = 32, 1000, 100, 5000


def build():
    torch.manual_seed(0)
    model = E2E(80, V, bench.c2_args(0.1)).to(dev).train()
    model.sync_report = False
    ops.manual_seed(1234)
    flat = train.FlatParams(model)
    opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=256, warmup=25000, max_grad_norm=5.0)
    xs, ilens, ys = bench.synth_batch(B, T, L, V)
    return model, flat, opt, model.prepare(xs, ilens, ys)


def timeit(run, n=30):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for phases in (True, False):
    model, flat, opt, batch = build()
    step = train.GraphedDataParallelStep(model, flat, opt, batch, world=1, phases=phases)
    print("no process group, phases=%s: %.3f ms" % (phases, timeit(step)))
    del step, model, flat, opt
torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1)
for phases in (True, False):
    model, flat, opt, batch = build()
    step = train.GraphedDataParallelStep(model, flat, opt, batch, world=1, phases=phases)
    print("one-rank RCCL group, phases=%s: %.3f ms" % (phases, timeit(step)))
    del step, model, flat, opt
torch.distributed.destroy_process_group()
