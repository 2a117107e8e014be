#!/usr/bin/env python3
"""One training step (forward + backward) of BASELINE configs 4 and 5 at full size on synthetic data:
wall time per step and sanity (finite loss, finite gradient norm).  Not the headline bench (bench.py = config 2)."""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd  # noqa: E402

DEV = "cuda"


def run(name, model, xs, ilens, ys, steps=3, graph=True):
    from espnet_amd import train
    model = model.to(DEV).train()
    nparam = sum(p.numel() for p in model.parameters())
    flat = train.FlatParams(model)          # gradients accumulate into the flat arena (one memset per step)
    ys = ys.cpu()                           # labels are parsed on the host (as in the reference): no device sync
    times = []

    def step():
        from espnet_amd import ops
        flat.zero_grad()
        loss = model(xs, ilens, ys)
        ops.wgrad_group_begin()         # small weight-gradient GEMMs (one per LSTM time step ...) leave as one grouped launch
        try:
            loss.backward()
        finally:
            ops.wgrad_group_end()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = step()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    torch.cuda.current_stream().wait_stream(side)
    eager = min(times)
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = step()
        torch.cuda.synchronize()
        times = []
        for i in range(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    flat.expose_grads()
    name = name + (" [hipGraph replay; eager %.1f ms]" % (eager * 1e3) if graph else " [eager]")
    gn = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in model.parameters() if p.grad is not None))
    frames = int(sum(ilens))
    print("%s: params %.1fM loss %.4f gradnorm %.4f step %.1f ms (best of %d) -> %.0f frames/s, peak mem %.1f GB" %
          (name, nparam / 1e6, float(loss), gn, min(times) * 1e3, steps, frames / min(times),
           torch.cuda.max_memory_allocated() / 2 ** 30))
    assert math.isfinite(float(loss)) and math.isfinite(gn)


def config4(B, T, L, V):
    from espnet_amd.nets.e2e_asr import E2E
    ns = argparse.Namespace(elayers=3, subsample="1_1_1_1", etype="vggblstm", eunits=1024, eprojs=1024, dtype="lstm",
                            dlayers=1, dunits=1024, atype="location", aheads=4, awin=5, aconv_chans=10, aconv_filts=100,
                            mtlalpha=0.5, lsm_type="", lsm_weight=0.0, sampling_probability=0.0, adim=1024,
                            dropout_rate=0.0, dropout_rate_decoder=0.0, verbose=0, char_list=None, outdir=None,
                            ctc_type="builtin", sym_space="<space>", sym_blank="<blank>", context_residual=False,
                            use_frontend=False, replace_sos=False)
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(B, T, 80, generator=g).to(DEV)
    ilens = [T - 7 * i for i in range(B)]
    ys = torch.randint(1, V - 1, (B, L), generator=g).to(DEV)
    torch.manual_seed(0)
    run("config4 VGG-BLSTM + AttLoc (B=%d T=%d L=%d V=%d)" % (B, T, L, V), E2E(80, V, ns), xs, ilens, ys)


def config5(B, T, L, V):
    from espnet_amd.nets.e2e_asr_transducer import E2E
    arch = [dict(type="conformer", d_hidden=256, d_ff=2048, heads=4, macaron_style=True, use_conv_mod=True,
                 conv_mod_kernel=31)]
    ns = argparse.Namespace(etype="transformer", enc_block_arch=arch, enc_block_repeat=12,
                            transformer_enc_input_layer="conv2d", transformer_enc_self_attn_type="rel_self_attn",
                            transformer_enc_positional_encoding_type="rel_pos",
                            transformer_enc_pw_activation_type="swish", transformer_enc_conv_mod_activation_type="swish",
                            dtype="lstm", dlayers=1, dunits=512, dec_embed_dim=512, joint_dim=320,
                            joint_activation_type="tanh", dropout_rate_decoder=0.0, dropout_rate_embed_decoder=0.0,
                            rnnt_mode="rnnt", trans_type="warp-transducer", sym_space="<space>", sym_blank="<blank>",
                            transformer_init="pytorch")
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(B, T, 80, generator=g).to(DEV)
    ilens = [T - 11 * i for i in range(B)]
    ys = torch.randint(1, V - 1, (B, L), generator=g).to(DEV)
    torch.manual_seed(0)
    run("config5 Conformer RNN-T (B=%d T=%d U=%d V=%d)" % (B, T, L + 1, V), E2E(80, V, ns), xs, ilens, ys)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="4,5")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    espnet_amd.set_precision(a.precision)
    if "4" in a.which:
        config4(*((4, 200, 20, 500) if a.small else (32, 1000, 100, 5000)))
    if "5" in a.which:
        config5(*((2, 300, 20, 500) if a.small else (16, 1500, 100, 5000)))
