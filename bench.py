#!/usr/bin/env python3
"""Headline benchmark: one full training step (forward + CTC/attention loss + backward + gradient
clip + Adam/Noam update; gradient all-reduce for N>1) of the 12-layer Conformer hybrid CTC/attention
model of BASELINE.json config 2 on synthetic 80-dim fbank (B=32 per GPU, T=1000, L=100, |V|=5000).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task description).  The headline (`value`, `ms_per_step`, `dtype`,
`roofline`) is measured at the REFERENCE's precision: fp32 storage and fp32 MFMA arithmetic everywhere
(`--precision fp32`, the default).  The same line carries
  * `bf16`         - the same K steps in bf16-operand mode (bf16 GEMM operands / MFMA, fp32 everything else), with its
                     own roofline and its step-0 loss against the fp32 one;
  * `parity`       - step-0, dropout-0 loss of the very batch being timed: HIP (both modes) vs the CPU oracle;
  * `roofline`     - the dominant kernel family (MFMA contractions), measured live with stream events;
  * `roofline_hbm` - the HBM-bound kernels of SURVEY 8(d), each timed live on its config-2 operands;
  * `cpu_baseline` - the CPU oracle timed on this box's host cores (1 warm-up + 3 timed steps, median).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# multiply-add = 2 FLOP, backward = 2x forward, per training step at config 2
FWD_GFLOP_PER_STEP = 832.6
STEP_FLOP = 2.498e12
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def c2_args(dropout=0.0):
    """egs/aishell/asr1/conf/tuning/train_pytorch_conformer_kernel31.yaml"""
    return argparse.Namespace(
        adim=256, aheads=4, elayers=12, eunits=2048, dlayers=6, dunits=2048, mtlalpha=0.3, lsm_weight=0.1,
        dropout_rate=dropout, transformer_attn_dropout_rate=0.0, transformer_length_normalized_loss=False,
        transformer_init="pytorch", transformer_input_layer="conv2d",
        transformer_encoder_pos_enc_layer_type="rel_pos", transformer_encoder_selfattn_layer_type="rel_selfattn",
        transformer_encoder_activation_type="swish", macaron_style=True, use_cnn_module=True, cnn_module_kernel=31)


def synth_batch(B, T, L, V, idim=80, seed=1):
    g = torch.Generator().manual_seed(seed)
    xs = torch.randn(B, T, idim, generator=g)
    ilens = [T] * B
    ys = torch.randint(1, V - 1, (B, L), generator=g)
    return xs, ilens, ys


def cpu_baseline(B, T, L, V, threads):
    """Times the CPU oracle (a restatement of the reference's PyTorch-CPU path, kind='port') on a bounded sample of
    the same workload: forward + backward of B utterances x T frames, dropout 0, the bench's seed-0 weights and
    rank-0 batch; 1 small warm-up + 3 timed steps, median.  Also returns the oracle's step-0 losses."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import asr_oracle as oracle
    from espnet_amd.nets.e2e_asr_conformer import E2E
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = E2E(80, V, c2_args())
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
          for k, v in model.state_dict().items()}
    cfg = dict(conformer=True, rel_pos=True, activation="swish", aheads=4, mtlalpha=0.3, lsm_weight=0.1, odim=V)
    xs, ilens, ys = synth_batch(1, 200, 10, V, seed=3)
    oracle.e2e_forward(sd, xs, ilens, ys, cfg, training=True)["loss"].backward()      # thread-pool warm-up
    xs, ilens, ys = synth_batch(B, T, L, V)
    dts, losses = [], None
    for _ in range(3):
        for v in sd.values():
            if torch.is_tensor(v) and v.grad is not None:
                v.grad = None
        t0 = time.perf_counter()
        out = oracle.e2e_forward(sd, xs, ilens, ys, cfg, training=True)
        out["loss"].backward()
        dts.append(time.perf_counter() - t0)
        losses = {k: float(out[k]) for k in ("loss", "loss_ctc", "loss_att")}
    dt = sorted(dts)[1]
    base = dict(value=round(B * T / dt, 1), unit="frames/s", cores=threads, kind="port",
                sample="fwd+bwd of B=%d T=%d L=%d V=%d fp32, dropout 0: 1 small warm-up + 3 timed steps %s s, median" %
                       (B, T, L, V, "/".join("%.1f" % d for d in dts)))
    return base, losses


def hbm_rooflines(prec):
    """The HBM-bound kernels of SURVEY 8(d) on their config-2 operands (storage dtype of `prec`), each replayed from
    a hipGraph between two stream events; bytes = ALGORITHMIC minimum traffic (every operand read once, every result
    written once)."""
    import espnet_amd  # noqa: F401
    from espnet_amd import ops
    dev = torch.device("cuda")
    adt = ops.act_dtype()
    asz = 2 if adt == torch.bfloat16 else 4
    M, D, V, Bq, Tq, Lq = 7968, 256, 5000, 32, 249, 100
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    out = []

    def timed(name, fn, nbytes, reps=20):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(gph, stream=s):
                for _ in range(reps):
                    fn()
        torch.cuda.synchronize()
        gph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * reps)
        gbs = nbytes / us / 1e3
        out.append(dict(kernel=name, bytes=int(nbytes), us=round(us, 2), achieved=round(gbs, 1), peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4)))

    x, gam, bet = rnd(M, D), torch.ones(D, device=dev), torch.zeros(D, device=dev)
    timed("layernorm_fwd [7968x256]", lambda: ops.layernorm_fwd(x, gam, bet, 1e-12, adt), M * D * (4 + asz))
    y, mean, rstd = ops.layernorm_fwd(x, gam, bet, 1e-12, adt)
    dy, dres, dg, db = rnd(M, D), rnd(M, D), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dln = ops.defer_ln_reduce
    ops.defer_ln_reduce = False
    timed("layernorm_bwd [7968x256] (+ residual gradient, gamma/beta reduction)",
          lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, dres, dg, db), M * D * 4 * 4)
    ops.defer_ln_reduce = dln
    acts = rnd(Bq, Tq, V)
    ys = torch.randint(1, V - 1, (Bq, Lq), generator=g).to(dev)
    hl = torch.full((Bq,), Tq, dtype=torch.int32, device=dev)
    timed("ctc_loss [32x249x5000] (prep + lse + alpha/beta + grad)", lambda: ops.ctc_loss(acts, ys, hl, 0, -1, 1.0 / Bq),
          2 * Bq * Tq * V * 4, reps=5)
    logits = rnd(Bq * (Lq + 1), V)
    tgt = torch.randint(0, V, (Bq * (Lq + 1),), generator=g).to(dev)
    timed("lsm_loss [3232x5000] (loss + grad + argmax)", lambda: ops.lsm_loss(logits, tgt, 0.1, 1.0 / Bq, -1),
          2 * Bq * (Lq + 1) * V * 4, reps=10)
    n = 46_840_000
    pp, gg, mm, vv = (torch.zeros(n, device=dev) for _ in range(4))
    p16 = torch.zeros(n, device=dev, dtype=torch.bfloat16) if asz == 2 else None
    state = torch.zeros(8, device=dev)
    state[0], state[1], state[6] = 1.0, 1e-3, 1.0
    timed("adam_step [46.84 M params]%s" % (" + bf16 shadow" if asz == 2 else ""),
          lambda: ops.adam_step(pp, gg, mm, vv, state, 0.9, 0.98, 1e-9, 0.0, p16=p16), n * (7 * 4 + (2 if asz == 2 else 0)), reps=5)
    del pp, gg, mm, vv, p16
    xin, w1, b1 = rnd(Bq, 1000, 80), rnd(256, 9), rnd(256)
    H1, W1 = 499, 39
    timed("conv1_fwd [32x1000x80 -> 32x499x39x256]", lambda: ops.conv1_fwd(xin, w1, b1, Bq, 1000, 80, 256, adt),
          Bq * 1000 * 80 * 4 + Bq * H1 * W1 * 256 * asz, reps=5)
    dy1 = rnd(Bq, H1, W1, 256).to(adt)
    dw1, db1 = torch.zeros(256, 9, device=dev), torch.zeros(256, device=dev)
    timed("conv1_bwd_w [dy 32x499x39x256]", lambda: ops.conv1_bwd_w(dy1, xin, dw1, db1, Bq, 1000, 80, 256),
          Bq * 1000 * 80 * 4 + Bq * H1 * W1 * 256 * asz, reps=5)
    del dy1
    xg, wd, bd = rnd(M, D), rnd(D, 31), rnd(D)
    timed("dwconv_fwd [32x249x256, k=31]", lambda: ops.dwconv_fwd(xg, wd, bd, Bq, Tq, D, 31), 2 * M * D * 4)
    return out


def run_precision(a, prec, rank, world, dev):
    """One full measurement at `prec`: fresh seed-0 model, K timed steps (graph replay), MFMA-family roofline.
    Returns (result dict, step-0 dropout-0 losses dict)."""
    import espnet_amd
    from espnet_amd import ops, train
    from espnet_amd.nets.e2e_asr_conformer import E2E
    espnet_amd.set_precision(prec)
    B, T, L, V = a.batch, a.frames, 100, 5000
    torch.manual_seed(0)
    model = E2E(80, V, c2_args(a.dropout)).to(dev).train()
    model.sync_report = False
    ops.manual_seed(1234 + 1000003 * rank)
    flat = train.FlatParams(model)
    opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=256, warmup=25000, max_grad_norm=5.0)
    xs, ilens, ys = synth_batch(B, T, L, V, seed=1 + rank)
    batch = model.prepare(xs, ilens, ys)

    # ---- step-0, dropout-0 losses on the very batch being timed (parity field) ----
    step0 = None
    if rank == 0:
        torch.manual_seed(0)
        m0 = E2E(80, V, c2_args(0.0)).to(dev).train()
        m0.sync_report = False
        m0.load_state_dict(model.state_dict())
        f0 = train.FlatParams(m0)
        with torch.no_grad():
            m0.forward_core(m0.prepare(xs, ilens, ys))
        step0 = dict(loss=float(m0.loss), loss_ctc=float(m0._loss_ctc_t), loss_att=float(m0._loss_att_t))
        del m0, f0
        torch.cuda.empty_cache()

    reducer = None
    if a.rehearse_dp and world == 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1)
    dp_graph = (world > 1 or a.rehearse_dp) and a.dp_mode in ("graph", "graph1") and not a.no_graph
    if world > 1 and not dp_graph:
        reducer = train.GradReducer(flat, bucket_mb=48.0)
        train.attach_reducer(reducer)
    ops.enable_wgrad_stream(a.wgrad_stream)

    def step():
        return train.train_step(model, flat, opt, batch, reducer)

    use_graph = (not a.no_graph) and world == 1 and not dp_graph
    dp_step = None
    if dp_graph:
        dp_step = train.GraphedDataParallelStep(model, flat, opt, batch, world=world, phases=(a.dp_mode == "graph"))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(0 if dp_step is not None else max(2, min(a.warmup, 3))):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if dp_step is not None:
        run = dp_step
    elif use_graph:
        from espnet_amd import graphs
        graph = graphs.new_graph()
        with torch.cuda.graph(graph):
            step()
        graph_nodes = graphs.audit(graph, "training step graph")      # node kinds; raises on memset nodes (espnet_amd/graphs.py)
        run = graph.replay
    else:
        run = step
    for _ in range(a.warmup):
        run()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    ms = dt / a.steps * 1e3
    loss_val = float(model.loss.detach())
    st = opt.stats()

    # ---- data-parallel runs explain themselves: collectives alone, the step with and without them, and the N = 1
    #      step (one graph, no phases) measured in this very process on rank 0 ----
    comm = None
    if dp_step is not None:
        comm = dp_step.comm_profile(steps=8)
        if world > 1:
            worst = torch.tensor([comm["step_ms"], comm["compute_only_ms"], comm["allreduce_ms_total"]], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(worst, op=torch.distributed.ReduceOp.MAX)
            comm["max_over_ranks"] = dict(step_ms=round(float(worst[0]), 3), compute_only_ms=round(float(worst[1]), 3),
                                          allreduce_ms_total=round(float(worst[2]), 3))
        if rank == 0:
            g1 = torch.cuda.CUDAGraph()
            s1 = torch.cuda.Stream()
            s1.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s1):
                step()
            torch.cuda.current_stream().wait_stream(s1)
            torch.cuda.synchronize()
            with torch.cuda.graph(g1):
                step()
            for _ in range(2):
                g1.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(8):
                g1.replay()
            torch.cuda.synchronize()
            comm["n1_ms"] = round((time.perf_counter() - t1) / 8 * 1e3, 3)
            comm["note"] = ("allreduce_ms: each phase's arena range reduced alone (stream events); exposed_comm_ms = step_ms - "
                            "compute_only_ms (same graphs, no collective); n1_ms = the single-graph step bench.py times at --gpus 1, "
                            "run on rank 0 while the other ranks idle")
            del g1
        if world > 1:
            torch.distributed.barrier()

    # ---- roofline of the dominant kernel family (MFMA contractions), measured live with stream events ----
    # Every MFMA-contraction launch of one training step (eamd_gemm descriptors, the fused attention kernels) is
    # recorded (operands kept alive), then the whole family is replayed back to back as ONE hipGraph on a stream and
    # bracketed by a single pair of events on that stream: device time of the kernels themselves, no host gaps - the
    # quantity the rocprofv3 kernel trace of the same command reports as the family's total duration.
    roof = None
    if rank == 0:
        rec = []
        rd = reducer
        try:
            if reducer is not None:
                train.attach_reducer(None)
            ops._gemm_record = rec
            train.train_step(model, flat, opt, batch, None)
            torch.cuda.synchronize()
        finally:
            ops._gemm_record = None
            if rd is not None:
                train.attach_reducer(rd)
        n = len(rec)

        def replay_all():
            sp = ops.stream_ptr()
            for _p, _keep, replay in rec:
                replay(sp)

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
        reps = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gg.replay()
        e1.record()
        torch.cuda.synchronize()
        gemm_ms = e0.elapsed_time(e1) / reps
        flop_per_launch = STEP_FLOP * (B / 32.0) * (T / 1000.0) / n
        avg_ms = gemm_ms / n
        ach = flop_per_launch / (avg_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch of this family from the last committed PMC passes of this precision
            import glob
            traffic_src = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_%s.json" % prec)))[-1]
            with open(traffic_src) as fh:
                traffic = round(json.load(fh)["gemm_family"]["hbm_bytes_per_launch"])
            traffic_src = os.path.basename(traffic_src)
        except Exception:  # noqa: BLE001
            traffic = None
        kern = ("gemm_f32_kernel<*> (+ gemm_kernel<*> for unaligned operands) + ffn_f32_direct_kernel<*> (fused position-wise FFN) + rowproj_f32_kernel<*> (row-block projections, LayerNorm forward / backward inside) + attn_f32_{fwd,bwd_q,bwd_kv}_kernel: every MFMA contraction, v_mfma_f32_16x16x4_f32"
                if prec == "fp32" else "gemm_bf16_*_kernel<*> + ffn_bf16_kernel<*> (fused position-wise FFN) + attn_{fwd,bwd_q,bwd_kv}_kernel: every MFMA contraction, v_mfma_f32_16x16x32_bf16")
        roof = dict(bound="mfma", achieved=round(ach, 2), peak=PEAK_TFLOPS[prec], unit="TFLOP/s",
                    frac=round(ach / PEAK_TFLOPS[prec], 4), traffic=traffic,
                    traffic_note=("HBM bytes per launch, rocprofv3 PMC (FETCH_SIZE doubled + WRITE_SIZE), profiles/%s" % traffic_src) if traffic else "no PMC pass committed for this build",
                    kernel=kern, launches_per_step=n, flop_per_launch=round(flop_per_launch),
                    avg_launch_us=round(avg_ms * 1e3, 2), gemm_ms_per_step=round(gemm_ms, 3),
                    whole_step_frac=round(STEP_FLOP * (B / 32.0) * (T / 1000.0) / (ms * 1e-3) / 1e12 / PEAK_TFLOPS[prec], 4),
                    timing="all MFMA-contraction launches of one step replayed as one hipGraph between two stream events")
        del gg, rec

    frames = B * T * world * a.steps
    res = dict(ms_per_step=round(ms, 3), value=round(frames / dt, 1), utt_per_s=round(B * world * a.steps / dt, 2),
               loss=round(loss_val, 4), optimizer_steps_done=st["step"], grad_norm=round(st["grad_norm"], 4),
               launch=(("hipGraph fwd+bwd in %d phases, RCCL all-reduce of each phase's arena range under the next | hipGraph optimizer"
                        % len(dp_step.ranges)) if dp_step is not None else "hipGraph" if use_graph else "eager"),
               roofline=roof)
    if use_graph and dp_step is None:
        res["graph_nodes"] = graph_nodes
    if comm is not None:
        res["comm"] = comm
    if reducer is not None:
        train.attach_reducer(None)
    del model, flat, opt, run
    torch.cuda.empty_cache()
    return res, step0


def ragged_leg(a, prec, dev):
    """--ragged: a stream of batches whose length varies, T ~ U(600, 1000) per batch (utterance lengths inside a batch
    fall off linearly to 0.6 T), through (i) eager steps on the exact shapes and (ii) train.BucketedGraphStep (64-frame
    buckets, LRU of 8 graphs).  Three passes over the same 28-batch stream; the last one is timed (every bucket is
    captured by then)."""
    import espnet_amd
    from espnet_amd import ops, train
    from espnet_amd.nets.e2e_asr_conformer import E2E
    espnet_amd.set_precision(prec)
    B, L, V = a.batch, 100, 5000
    g = torch.Generator().manual_seed(17)
    stream = []
    for _ in range(28):
        T = int(torch.randint(600, 1001, (1,), generator=g))
        xs = torch.randn(B, T, 80, generator=g)
        ilens = [int(round(v)) for v in torch.linspace(T, 0.6 * T, B).tolist()]
        for i, n in enumerate(ilens):
            xs[i, n:] = 0.0
        stream.append((xs.to(dev), ilens, torch.randint(1, V - 1, (B, L), generator=g).to(dev)))
    frames = sum(sum(il) for _x, il, _y in stream)
    out = {}
    # "composed" = train.ComposedStep: the bucketed step with the backward in the data-parallel phases (one hipGraph per phase and
    # bucket, each phase's arena range all-reduced under the next phase) - with --rehearse-dp through a one-rank RCCL group, so
    # that the collectives really run; without a process group the phases replay back to back
    for mode in ("eager", "bucketed", "composed"):
        torch.manual_seed(0)
        model = E2E(80, V, c2_args(a.dropout)).to(dev).train()
        model.sync_report = False
        flat = train.FlatParams(model)
        opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=256, warmup=25000, max_grad_norm=5.0)
        step = train.BucketedGraphStep(model, flat, opt, t_edge=64, l_edge=8, max_graphs=8) if mode == "bucketed" else None
        comp = (train.ComposedStep(train.E2EProgram(model, flat, t_edge=64, l_edge=8), flat, opt, max_graphs=8,
                                   rehearse=a.rehearse_dp and torch.distributed.is_initialized()) if mode == "composed" else None)
        for ep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for xs, ilens, ys in stream:
                if step is not None:
                    step(xs, ilens, ys, olens=[L] * B)       # label lengths from the host, as a data loader has them
                elif comp is not None:
                    comp.step((xs, ilens, ys, [L] * B))
                else:
                    train.train_step(model, flat, opt, model.prepare(xs, ilens, ys))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[mode] = dict(ms_per_step=round(dt / len(stream) * 1e3, 3), valid_frames_per_s=round(frames / dt, 1))
        st = step.stats() if step is not None else comp.stats() if comp is not None else None
        if st is not None:
            out[mode].update(hit_rate_last_pass=1.0 if st["evictions"] == 0 else None, buckets=st["graphs"],
                             captures=st["captures"], evictions=st["evictions"], hit_rate_overall=round(st["hit_rate"], 3))
        if comp is not None:
            out[mode].update(phases=st["phases"], collectives=("one-rank RCCL group: each phase's arena range all-reduced under the next phase"
                                                               if (a.rehearse_dp and torch.distributed.is_initialized()) else "none (no process group)"))
        del model, flat, opt, step, comp
        torch.cuda.empty_cache()
    out["what"] = "28 batches, B=%d, T ~ U(600,1000) per batch, lengths linspace(T, 0.6T), L=100; third pass timed" % B
    return out


DTYPE = {"fp32": ("f32", "fp32 storage and arithmetic everywhere (v_mfma_f32_16x16x4_f32): the reference's precision"),
         "bf16": ("bf16", "bf16 GEMM operands + MFMA, fp32 accumulate / residual stream / master weights / optimizer")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="fp32", choices=["bf16", "fp32"], help="precision of the headline")
    ap.add_argument("--no-second-precision", action="store_true", help="skip the other precision's object")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--dropout", type=float, default=0.1, help="dropout-rate of the recipe (yaml: 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-roofline", action="store_true")
    ap.add_argument("--wgrad-stream", action="store_true",
                    help="issue weight-gradient GEMMs on a second stream (measured: no gain on MI355X, off by default)")
    ap.add_argument("--cpu-sample-batch", type=int, default=32)
    ap.add_argument("--dp-mode", default="graph", choices=["graph", "graph1", "overlap"],
                    help="N>1: 'graph' = hipGraph replay of forward + backward in phases (decoder/CTC | upper | "
                         "lower encoder layers), each phase's range of the gradient arena all-reduced over RCCL under "
                         "the next phase, then the optimizer graph; 'graph1' = one forward/backward graph, the whole "
                         "arena all-reduced behind it; 'overlap' = eager launches with backward-overlapped bucket "
                         "all-reduces")
    ap.add_argument("--ragged", action="store_true",
                    help="N=1: add a `ragged` object: variable-length batch stream, eager vs shape-bucketed hipGraph cache")
    ap.add_argument("--no-decode", action="store_true", help="skip the `decode` object (greedy CTC / beam searches, ~30 s)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the `configs` object (BASELINE configs 4 and 5 at full size, both precisions, ~40 s)")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="N=1 only: run the N>1 'graph' code path on a one-rank RCCL group")
    a = ap.parse_args()

    from espnet_amd import train
    rank, local_rank, world = train.init_distributed()
    if world != a.gpus and world > 1:
        a.gpus = world
    dev = torch.device("cuda", local_rank)
    B, T, L, V = a.batch, a.frames, 100, 5000

    head, step0_head = run_precision(a, a.precision, rank, world, dev)
    other = "bf16" if a.precision == "fp32" else "fp32"
    second = step0_other = None
    if not a.no_second_precision:
        second, step0_other = run_precision(a, other, rank, world, dev)

    hbm = None
    if rank == 0 and not a.no_hbm_roofline:
        import espnet_amd
        espnet_amd.set_precision(a.precision)
        hbm = hbm_rooflines(a.precision)

    ragged = None
    if rank == 0 and world == 1 and a.ragged:
        ragged = ragged_leg(a, a.precision, dev)

    extra = None
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if rank == 0 and world == 1 and not a.no_extra_configs:
        from tools import bench_rnn
        extra = bench_rnn.extra_configs()

    decode = None
    if rank == 0 and world == 1 and not a.no_decode:
        from tools import bench_decode
        decode = bench_decode.decode_leg(dev, c2_args)

    cpu = orc = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        threads = os.cpu_count() or 1
        try:
            threads = len(os.sched_getaffinity(0))
        except Exception:  # noqa: BLE001
            pass
        cpu, orc = cpu_baseline(a.cpu_sample_batch, T, L, V, min(threads, 64))

    if rank == 0:
        parity = None
        if step0_head is not None:
            parity = dict(what="step-0 loss of the timed batch, dropout 0, seed-0 weights",
                          loss_hip=round(step0_head["loss"], 6), precision=a.precision)
            if orc is not None and a.cpu_sample_batch == B:
                parity.update(loss_oracle=round(orc["loss"], 6),
                              rel=float("%.3e" % (abs(step0_head["loss"] - orc["loss"]) / abs(orc["loss"]))),
                              loss_ctc_rel=float("%.3e" % (abs(step0_head["loss_ctc"] - orc["loss_ctc"]) / abs(orc["loss_ctc"]))),
                              loss_att_rel=float("%.3e" % (abs(step0_head["loss_att"] - orc["loss_att"]) / abs(orc["loss_att"]))))
            if step0_other is not None:
                parity["loss_hip_" + other] = round(step0_other["loss"], 6)
                if orc is not None and a.cpu_sample_batch == B:
                    parity["rel_" + other] = float("%.3e" % (abs(step0_other["loss"] - orc["loss"]) / abs(orc["loss"])))
        sec = None
        if second is not None:
            sec = dict(dtype=DTYPE[other][0], dtype_detail=DTYPE[other][1], ms_per_step=second["ms_per_step"],
                       value=second["value"], unit="frames/s", utt_per_s=second["utt_per_s"], loss=second["loss"],
                       roofline=second["roofline"])
            if second.get("comm") is not None:
                sec["comm"] = second["comm"]
            if step0_head is not None and step0_other is not None:
                sec["loss_rel_vs_%s_step0" % a.precision] = float(
                    "%.3e" % (abs(step0_other["loss"] - step0_head["loss"]) / abs(step0_head["loss"])))
        out = {
            "metric": "frames/sec, 12L Conformer hybrid CTC/attention training step (B=32 T=1000 d=256 per GPU)",
            "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[a.precision][0], "dtype_detail": DTYPE[a.precision][1],
            "data": "synthetic", "utt_per_s": head["utt_per_s"], "loss": head["loss"],
            "config": {"workload": "BASELINE configs[1]: 12L Conformer enc d=256 h=4 ff=2048 k=31 macaron+cnn rel_pos, "
                                   "6L Transformer dec, V=5000, fbank B=%d T=%d L=100, mtlalpha 0.3, lsm 0.1" % (B, T),
                       "global_batch": B * world, "frames": T, "parallelism": "dp%d" % world,
                       "dropout": a.dropout, "optimizer": "adam+noam, clip 5.0", "launch": head["launch"],
                       "wgrad_side_stream": a.wgrad_stream, "optimizer_steps_done": head["optimizer_steps_done"],
                       "grad_norm": head["grad_norm"], "graph_nodes": head.get("graph_nodes")},
            "roofline": head["roofline"], "roofline_hbm": hbm, "cpu_baseline": cpu, "parity": parity,
            other: sec,
        }
        if head.get("comm") is not None:
            out["comm"] = head["comm"]
            out["rccl_ranks"] = head["comm"]["rccl_ranks"]
        if ragged is not None:
            out["ragged"] = ragged
        if extra is not None:
            out["configs"] = extra
        if decode is not None:
            out["decode"] = decode
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
