#!/usr/bin/env python3
"""Where the GEMM family's time goes: every eamd_gemm descriptor of one config-2 training step, grouped by shape /
layout / epilogue, each group timed by graph replay of its own launches (device time, no host gaps)."""
import collections
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import espnet_amd  # noqa: E402
from espnet_amd import _lib as L_, ops, train  # noqa: E402
from espnet_amd.nets.e2e_asr_conformer import E2E  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    espnet_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
    B, T, L, V = 32, 1000, 100, 5000
    torch.manual_seed(0)
    if "--config4" in sys.argv:
        # BASELINE config 4 (VGG-BLSTM + location attention): the recurrent / decoder-step GEMMs
        import argparse
        from espnet_amd.nets.e2e_asr import E2E as E2ERnn
        ns = argparse.Namespace(elayers=3, subsample="1_1_1_1", etype="vggblstm", eunits=1024, eprojs=1024, dtype="lstm",
                                dlayers=1, dunits=1024, atype="location", aheads=4, awin=5, aconv_chans=10, aconv_filts=100,
                                mtlalpha=0.5, lsm_type="", lsm_weight=0.0, sampling_probability=0.0, adim=1024,
                                dropout_rate=0.0, dropout_rate_decoder=0.0, verbose=0, char_list=None, outdir=None,
                                ctc_type="builtin", sym_space="<space>", sym_blank="<blank>", context_residual=False,
                                use_frontend=False, replace_sos=False)
        g = torch.Generator().manual_seed(0)
        xs = torch.randn(B, T, 80, generator=g).to(dev)
        ilens = [T - 7 * i for i in range(B)]
        ys = torch.randint(1, V - 1, (B, L), generator=g)
        model = E2ERnn(80, V, ns).to(dev).train()
        flat = train.FlatParams(model)

        def one_step():
            flat.zero_grad()
            loss = model(xs, ilens, ys)
            ops.wgrad_group_begin()
            try:
                loss.backward()
            finally:
                ops.wgrad_group_end()
    else:
        model = E2E(80, V, bench.c2_args(0.1)).to(dev).train()
        model.sync_report = False
        ops.manual_seed(1234)
        flat = train.FlatParams(model)
        opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=256, warmup=25000, max_grad_norm=5.0)
        xs, ilens, ys = bench.synth_batch(B, T, L, V)
        batch = model.prepare(xs, ilens, ys)

        def one_step():
            train.train_step(model, flat, opt, batch, None)
    for _ in range(2):
        one_step()
    torch.cuda.synchronize()
    rec = []
    ops._gemm_record = rec
    one_step()
    torch.cuda.synchronize()
    ops._gemm_record = None
    lib = L_.lib()
    groups = collections.OrderedDict()
    tagged = {}
    for p, keep, replay in rec:
        if isinstance(p, dict):      # grouped / fused FFN / fused attention launch
            k_ = (p["kind"],) + (0,) * 16
            groups.setdefault(k_, []).append(replay)
            tagged[k_] = tagged.get(k_, 0.0) + p["flop"]
            continue
        key = (p.M, p.N, p.K, p.transA, p.transB, p.batch1 * p.batch2, p.splitk, p.epilogue, int(p.gather.enabled),
               int(bool(p.C)), int(bool(p.Cb)), int(bool(p.Hb)), int(bool(p.R)), int(bool(p.aux)), int(bool(p.colsum)),
               int(p.drop_p > 0), p.in_dtype)
        if "--config4" in sys.argv:
            key = key + (p.precision,)
        groups.setdefault(key, []).append(replay)
    rows = []
    for key, ps in groups.items():
        def f():
            sp = ops.stream_ptr()
            for replay in ps:
                replay(sp)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        reps = max(1, 40 // len(ps))
        with torch.cuda.stream(s):
            f()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(reps):
                    f()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 5 / reps / len(ps) * 1e6
        fl = tagged[key] / len(ps) if key in tagged else 2.0 * key[0] * key[1] * key[2] * key[5]
        rows.append((us * len(ps), len(ps), us, fl / us / 1e6, key))
    # algorithmic HBM bytes of the descriptor launches: every operand and result once (A, B, C, residual, aux, second output)
    esz = 2 if sys.argv[1:2] != ["fp32"] else 4
    alg = 0
    for p, keep, replay in rec:
        if isinstance(p, dict):
            continue
        nb = p.batch1 * p.batch2
        alg += nb * (p.M * p.K + p.N * p.K) * esz + nb * p.M * p.N * (4 if p.C else 0)
        alg += nb * p.M * p.N * ((2 if p.Cb else 0) + (4 if p.R else 0) + ((2 if p.aux_dtype else 4) if p.aux else 0) + (esz if p.Hb else 0))
    print("algorithmic operand + result bytes of the %d descriptor launches: %.2f GB per step" % (sum(1 for r in rec if r[0] is not None), alg / 1e9))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    print("total %.1f us over %d launches" % (tot, len(rec)))
    print("%9s %4s %8s %7s | M N K tA tB batch splitk epi gather C Cb Hb R aux colsum drop bf16in" % ("sum us", "n", "us each", "TF/s"))
    for r in rows:
        print("%9.1f %4d %8.1f %7.1f | %s" % (r[0], r[1], r[2], r[3], " ".join(str(k) for k in r[4])))


if __name__ == "__main__":
    main()
