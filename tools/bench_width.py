#!/usr/bin/env python3
"""One training step (hipGraph replay) of the config-2 recipe at another WIDTH: the reference's large recipes use adim 512 / aheads 8
(egs/librispeech/asr1 conformer).  usage: python tools/bench_width.py [--adim 512] [--aheads 8] [--precision fp32|bf16]
[--batch 32] [--frames 1000] [--steps 20]; under rocprofv3 --kernel-trace --stats this gives the per-kernel picture of that width."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--adim", type=int, default=512)
    ap.add_argument("--aheads", type=int, default=8)
    ap.add_argument("--eunits", type=int, default=2048)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import bench
    import espnet_amd
    from espnet_amd import graphs, ops, train
    from espnet_amd.nets.e2e_asr_conformer import E2E
    espnet_amd.set_precision(a.precision)
    dev = torch.device("cuda")
    ns = bench.c2_args(0.1)
    ns.adim, ns.aheads, ns.eunits, ns.dunits = a.adim, a.aheads, a.eunits, a.eunits
    torch.manual_seed(0)
    V = 5000
    model = E2E(80, V, ns).to(dev).train()
    model.sync_report = False
    ops.manual_seed(1234)
    flat = train.FlatParams(model)
    opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=a.adim, warmup=25000, max_grad_norm=5.0)
    xs, ilens, ys = bench.synth_batch(a.batch, a.frames, 100, V)
    batch = model.prepare(xs, ilens, ys)
    step = lambda: train.train_step(model, flat, opt, batch, None)  # noqa: E731
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = graphs.new_graph()
    with torch.cuda.graph(g):
        step()
    nodes = graphs.audit(g, "training step graph")
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    nparam = sum(p.numel() for p in model.parameters())
    print("adim %d aheads %d units %d %s: %.2f ms per step (B=%d T=%d), %.2f M frames/s, %d kernel nodes, %.1f M parameters, loss %.4f"
          % (a.adim, a.aheads, a.eunits, a.precision, ms, a.batch, a.frames, a.batch * a.frames / ms / 1e3, nodes.get("kernel", 0),
             nparam / 1e6, float(model.loss)))


if __name__ == "__main__":
    main()
