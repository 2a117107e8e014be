#!/usr/bin/env python3
"""Headline benchmark: one full training step (forward + CTC/attention loss + backward + gradient
clip + Adam/Noam update; gradient all-reduce for N>1) of the 12-layer Conformer hybrid CTC/attention
model of BASELINE.json config 2 on synthetic 80-dim fbank (B=32 per GPU, T=1000, L=100, |V|=5000).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (MFMA GEMM family,
measured live with stream events) and `cpu_baseline` (the CPU oracle timed on this box's host cores on
a bounded sample).
"""
import argparse
import ctypes
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
    """Times the CPU oracle (a restatement of the reference's PyTorch-CPU path, kind='port') on a
    bounded sample of the same workload: one forward+backward of B utterances x T frames."""
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
    t0 = time.perf_counter()
    out = oracle.e2e_forward(sd, xs, ilens, ys, cfg, training=True)
    out["loss"].backward()
    dt = time.perf_counter() - t0
    return dict(value=B * T / dt, unit="frames/s", cores=threads, kind="port",
                sample=f"1 fwd+bwd step, B={B} T={T} L={L} V={V} fp32, {dt:.1f}s, loss={float(out['loss']):.3f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--dropout", type=float, default=0.1, help="dropout-rate of the recipe (yaml: 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wgrad-stream", action="store_true",
                    help="issue weight-gradient GEMMs on a second stream (measured: no gain on MI355X, off by default)")
    ap.add_argument("--cpu-sample-batch", type=int, default=32)
    ap.add_argument("--dp-mode", default="graph", choices=["graph", "graph1", "overlap"],
                    help="N>1: 'graph' = hipGraph replay of forward + backward in three phases (decoder/CTC | upper | "
                         "lower encoder layers), each phase's range of the gradient arena all-reduced over RCCL under "
                         "the next phase, then the optimizer graph; 'graph1' = one forward/backward graph, the whole "
                         "arena all-reduced behind it; 'overlap' = eager launches with backward-overlapped bucket "
                         "all-reduces")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="N=1 only: run the N>1 'graph' code path on a one-rank RCCL group (exercises the path "
                         "the multi-GPU runs take)")
    a = ap.parse_args()

    import espnet_amd
    from espnet_amd import ops, train
    from espnet_amd.nets.e2e_asr_conformer import E2E

    rank, local_rank, world = train.init_distributed()
    if world != a.gpus and world > 1:
        a.gpus = world
    dev = torch.device("cuda", local_rank)
    espnet_amd.set_precision(a.precision)
    B, T, L, V = a.batch, a.frames, 100, 5000

    torch.manual_seed(0)
    model = E2E(80, V, c2_args(a.dropout)).to(dev).train()
    model.sync_report = False
    ops.manual_seed(1234 + 1000003 * rank)
    flat = train.FlatParams(model)
    opt = train.NoamAdam(flat, mode="noam", factor=1.0, model_size=256, warmup=25000, max_grad_norm=5.0)
    reducer = None
    if a.rehearse_dp and world == 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1)
    dp_graph = (world > 1 or a.rehearse_dp) and a.dp_mode in ("graph", "graph1") and not a.no_graph
    if world > 1 and not dp_graph:
        reducer = train.GradReducer(flat, bucket_mb=48.0)
        train.attach_reducer(reducer)
    xs, ilens, ys = synth_batch(B, T, L, V, seed=1 + rank)
    batch = model.prepare(xs, ilens, ys)
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
            loss = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = None
    if dp_step is not None:
        run = dp_step
    elif use_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = step()
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
    loss_val = float(model.loss)
    st = opt.stats()

    # ---- roofline of the dominant kernel family (MFMA GEMM), measured live with stream events ----
    # Every MFMA-contraction launch of one training step (eamd_gemm descriptors, the fused attention kernels
    # eamd_attn_fwd / eamd_attn_bwd_q) is recorded (operands kept alive), then the whole family is
    # replayed back to back as ONE hipGraph on the current stream and bracketed by a single pair of stream
    # events: device time of the kernels themselves, no host gaps, no per-launch event overhead - the quantity
    # the rocprofv3 kernel trace of the same command reports as the family's total duration.
    roof = None
    if rank == 0:
        from espnet_amd import _lib as L_
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
        lib = L_.lib()

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
        try:   # HBM bytes per GEMM launch from the last committed PMC passes (profiles/, FETCH_SIZE x2 + WRITE_SIZE)
            import glob
            traffic_src = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
            with open(traffic_src) as fh:
                traffic = round(json.load(fh)["gemm_family"]["hbm_bytes_per_launch"])
            traffic_src = os.path.basename(traffic_src)
        except Exception:  # noqa: BLE001
            traffic = None
        roof = dict(bound="mfma", achieved=round(ach, 2), peak=PEAK_TFLOPS[a.precision], unit="TFLOP/s",
                    frac=round(ach / PEAK_TFLOPS[a.precision], 4), traffic=traffic,
                    traffic_note="HBM bytes per launch, rocprofv3 PMC (FETCH_SIZE doubled + WRITE_SIZE), profiles/%s" % traffic_src,
                    kernel="gemm_*_kernel<*> + attn_{fwd,bwd_q}_kernel (all MFMA contractions)", launches_per_step=n,
                    avg_launch_us=round(avg_ms * 1e3, 2), gemm_ms_per_step=round(gemm_ms, 3),
                    timing="all MFMA-contraction launches of one step replayed as one hipGraph between two stream events")

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        threads = os.cpu_count() or 1
        try:
            threads = len(os.sched_getaffinity(0))
        except Exception:  # noqa: BLE001
            pass
        cpu = cpu_baseline(a.cpu_sample_batch, T, L, V, min(threads, 64))

    if rank == 0:
        frames = B * T * world * a.steps
        out = {
            "metric": "frames/sec, 12L Conformer hybrid CTC/attention training step (B=32 T=1000 d=256 per GPU)",
            "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if a.precision == "bf16" else "f32",
            "dtype_detail": "bf16 GEMM operands + MFMA, fp32 accumulate / residual stream / master weights / optimizer" if a.precision == "bf16" else "fp32 everywhere (v_mfma_f32_16x16x4_f32)",
            "data": "synthetic", "utt_per_s": round(B * world * a.steps / dt, 2), "loss": round(loss_val, 4),
            "config": {"workload": "BASELINE configs[1]: 12L Conformer enc d=256 h=4 ff=2048 k=31 macaron+cnn rel_pos, "
                                   "6L Transformer dec, V=5000, fbank B=%d T=%d L=100, mtlalpha 0.3, lsm 0.1" % (B, T),
                       "global_batch": B * world, "frames": T, "parallelism": "dp%d" % world,
                       "dropout": a.dropout, "optimizer": "adam+noam, clip 5.0", "launch": (("hipGraph fwd+bwd in %d phases, RCCL all-reduce of each phase's arena range under the next | hipGraph optimizer" % len(dp_step.ranges)) if dp_step is not None else "hipGraph" if use_graph else "eager"), "wgrad_side_stream": a.wgrad_stream,
                       "optimizer_steps_done": st["step"], "grad_norm": round(st["grad_norm"], 4)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
