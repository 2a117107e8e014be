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


def _replay_ms(replays, reps=5):
    """device time of a list of recorded launches replayed as one hipGraph"""
    def replay_all():
        from espnet_amd import ops
        sp = ops.stream_ptr()
        for r in replays:
            r(sp)
    gg = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        replay_all()
        torch.cuda.synchronize()
        with torch.cuda.graph(gg, stream=side):
            replay_all()
    torch.cuda.synchronize()
    gg.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gg.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _gemm_flops(p):
    return 2.0 * p.M * p.N * p.K * p.batch1 * p.batch2


def run(name, model, xs, ilens, ys, steps=3, graph=True, dominant=None, quiet=False):
    """one training step (forward + backward into the flat gradient arena) -> dict(ms_per_step, frames_per_s, peak_mem_gb,
    loss, eager_ms, dominant).  dominant = "lstm": the recurrent launches of the step (ops._rnn_record) replayed alone,
    against the L2 bandwidth their operands need; ("gemm", V): the MFMA products that touch the V-wide vocabulary axis
    (ops._gemm_record), against the MFMA peak of the precision."""
    from espnet_amd import ops, train
    model = model.to(DEV).train()
    nparam = sum(p.numel() for p in model.parameters())
    flat = train.FlatParams(model)          # gradients accumulate into the flat arena (one memset per step)
    ys = ys.cpu()                           # labels are parsed on the host (as in the reference): no device sync
    times = []

    def step():
        flat.zero_grad()
        loss = model(xs, ilens, ys)
        ops.wgrad_group_begin()         # small weight-gradient GEMMs (one per LSTM time step ...) leave as one grouped launch
        try:
            loss.backward()
        finally:
            ops.wgrad_group_end()
        return loss

    torch.cuda.reset_peak_memory_stats()
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
    dom = None
    if dominant is not None:
        grec, rrec = [], []
        ops._gemm_record, ops._rnn_record = grec, rrec
        try:
            step()
            torch.cuda.synchronize()
        finally:
            ops._gemm_record = ops._rnn_record = None
        if dominant == "lstm" and rrec:
            seq = [r for r in rrec if r[0].startswith("lstm_seq")]
            if seq:      # the persistent whole-sequence launches (csrc/lstm_seq.hip): fp32 MFMA recurrences, one grid hand-off per time step
                ms = _replay_ms([r[2] for r in seq])
                flop = float(sum(r[4] for r in seq))
                ach = flop / (ms * 1e-3) / 1e12
                nsteps = sum(r[1][0][0][0].shape[0] if r[0] == "lstm_seq_fwd" else r[1][0][0][2].shape[0] for r in seq)
                dom = dict(kernel="lstm_seq_fwd_kernel / lstm_seq_bwd_kernel: all time steps of both directions of a BLSTM layer in one "
                                  "persistent launch (recurrent weights in registers, h_t / dgates_t handed between workgroups per step)",
                           launches_per_step=len(seq), ms_per_step=round(ms, 3), time_steps=int(nsteps),
                           us_per_time_step=round(ms * 1e3 / nsteps, 2),
                           roofline=dict(bound="mfma", achieved=round(ach, 2), peak=157.3, unit="TFLOP/s", frac=round(ach / 157.3, 4),
                                         flop_per_step=flop,
                                         note="fp32 MFMA (v_mfma_f32_16x16x4_f32) in both precision modes; the rest of a time step "
                                              "is the grid-wide hand-off of h_t / dgates_t (DESIGN.md section 4)"),
                           other_recurrent_launches=len(rrec) - len(seq))
            else:
                ms = _replay_ms([r[2] for r in rrec])
                nbytes = float(sum(r[3] for r in rrec))
                ach = nbytes / (ms * 1e-3) / 1e9
                dom = dict(kernel="lstm_step_fwd_kernel / lstm_step_bwd_kernel: one launch per LSTM time step, recurrent weights + "
                                  "states from L2 / Infinity Cache (no HBM traffic: 16 MB per layer-direction stays resident)",
                           launches_per_step=len(rrec), ms_per_step=round(ms, 3), us_per_launch=round(ms * 1e3 / len(rrec), 2),
                           roofline=dict(bound="l2", achieved=round(ach, 1), peak=34500.0, unit="GB/s", frac=round(ach / 34500.0, 4),
                                         bytes_per_launch=int(nbytes / len(rrec)),
                                         note="bytes = W_hh + gate / state rows of the launch; peak = aggregate L2 bandwidth "
                                              "(MI355X_MICROARCH.md); the launches are latency-bound (DESIGN.md section 4)"))
        elif isinstance(dominant, tuple) and dominant[0] == "gemm":
            Vv = dominant[1]
            sel = [r for r in grec if r[0] is not None and not isinstance(r[0], dict) and Vv in (r[0].M, r[0].N, r[0].K)]
            if sel:
                ms = _replay_ms([r[2] for r in sel])
                fl = sum(_gemm_flops(r[0]) for r in sel)
                peak = 157.3 if espnet_amd.get_precision() == "fp32" else 2500.0
                ach = fl / (ms * 1e-3) / 1e12
                dom = dict(kernel="eamd_gemm products over the vocabulary axis (joint-network logits with the row-statistics / "
                                  "row-gradient epilogues, their dH and dW_out): gemm_%s_kernel<*>" %
                                  ("f32" if espnet_amd.get_precision() == "fp32" else "bf16"),
                           launches_per_step=len(sel), ms_per_step=round(ms, 3),
                           roofline=dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                                         flop_per_step=fl))
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
    gn = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in model.parameters() if p.grad is not None))
    frames = int(sum(ilens))
    res = dict(params_m=round(nparam / 1e6, 1), loss=round(float(loss), 4), grad_norm=round(gn, 4),
               ms_per_step=round(min(times) * 1e3, 2), eager_ms=round(eager * 1e3, 1), launch="hipGraph" if graph else "eager",
               frames_per_s=round(frames / min(times), 0), peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
               dominant=dom)
    if not quiet:
        print("%s: params %.1fM loss %.4f gradnorm %.4f step %.1f ms (best of %d; eager %.1f ms) -> %.0f frames/s, peak mem %.1f GB" %
              (name, nparam / 1e6, float(loss), gn, min(times) * 1e3, steps, eager * 1e3, frames / min(times), res["peak_mem_gb"]))
        if dom is not None:
            print("   dominant:", dom)
    assert math.isfinite(float(loss)) and math.isfinite(gn)
    return res


def config4(B, T, L, V, quiet=False):
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
    return run("config4 VGG-BLSTM + AttLoc (B=%d T=%d L=%d V=%d)" % (B, T, L, V), E2E(80, V, ns), xs, ilens, ys,
               dominant="lstm", quiet=quiet)


def config5(B, T, L, V, quiet=False):
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
    return run("config5 Conformer RNN-T (B=%d T=%d U=%d V=%d)" % (B, T, L + 1, V), E2E(80, V, ns), xs, ilens, ys,
               dominant=("gemm", V), quiet=quiet)


def extra_configs(precisions=("fp32", "bf16")):
    """bench.py's `configs` object: BASELINE configs[3] (c4) and configs[4] (c5) at full size, both precisions"""
    out = {"c4": {"workload": "BASELINE configs[3]: VGG-BLSTM 3x1024 + location-aware attention (10x100) + LSTM decoder 1024, "
                              "CTC 0.5, synthetic fbank B=32 T=1000 L=100 V=5000, forward + backward"},
           "c5": {"workload": "BASELINE configs[4]: RNN-Transducer, 12-block Conformer encoder d=256 + 1L LSTM predictor 512, joint 320, "
                              "streamed joint + rnnt loss, synthetic fbank B=16 T=1500 U=101 V=5000, forward + backward"}}
    keep = espnet_amd.get_precision()
    try:
        for prec in precisions:
            espnet_amd.set_precision(prec)
            out["c4"][prec] = config4(32, 1000, 100, 5000, quiet=True)
            torch.cuda.empty_cache()
            out["c5"][prec] = config5(16, 1500, 100, 5000, quiet=True)
            torch.cuda.empty_cache()
    finally:
        espnet_amd.set_precision(keep)
    return out


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
