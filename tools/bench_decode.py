#!/usr/bin/env python3
"""Decode leg of bench.py (`decode` object): BASELINE configs[1]'s model (random-init weights) on 32 synthetic utterances
of T = 1000 frames (10 ms frames: 10 s of audio each).

  greedy_ctc        E2E.greedy_ctc_batch: encoder + CTC argmax + collapse of the whole batch, on the device
  beam_search       nets.beam_search.BeamSearch, beam 10, ctc_weight 0.3 (decoder 0.7), one utterance at a time
  batch_beam_search nets.batch_beam_search.BatchBeamSearch, same weights (the reference's vectorised arithmetic)

Per mode: utterances / s, real-time factor (decode time / audio time), and for the beam searches the kernel launches and
device<->host copies of ONE beam step (torch.profiler), hypothesis bookkeeping on the host as in the reference.  With
random-init weights no hypothesis ends before the length cap, so the searches run `maxlenratio` * T' steps (the cap is
part of the workload description); a subset of the utterances is searched and the rate extrapolated (stated).
cpu_baseline_decode: the CPU oracle's greedy CTC (encoder + argmax + collapse) on a bounded sample, token ids compared with
the device's on every utterance of the sample (frames whose top-2 logit gap is below fp32 summation-order noise excluded).
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _count_step(fn):
    """kernel launches and memcpys of one call of fn, from torch.profiler's device activity records"""
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            fn()
            torch.cuda.synchronize()
        kern = d2h = h2d = 0
        for ev in prof.events():
            if str(getattr(ev, "device_type", "")).endswith("CUDA") or getattr(ev, "device_type", None) == torch.autograd.DeviceType.CUDA:
                n = ev.name
                if "Memcpy" in n or "memcpy" in n or "copyBuffer" in n:
                    if "DtoH" in n or "DeviceToHost" in n:
                        d2h += 1
                    elif "HtoD" in n or "HostToDevice" in n:
                        h2d += 1
                elif "Memset" not in n:
                    kern += 1
        return dict(kernel_launches=kern, d2h_copies=d2h, h2d_copies=h2d)
    except Exception as e:  # noqa: BLE001
        return dict(error=str(e)[:120])


def _graph_leg(tag, n_utts, beam, ctc_weight, maxlenratio):
    """single-utterance searches with the steps as hipGraph replays (BeamSearch.graph_steps), in a CHILD process: replaying the
    step graphs of multi-utterance searches has ended in a GPU fault on this ROCm, and no fault may take the bench with it"""
    import json
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_decode_graph.py"), str(n_utts), str(beam), str(ctc_weight),
                            str(maxlenratio), tag], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return dict(error="child exited with %d: %s" % (r.returncode, r.stderr[-200:]))
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return dict(error=str(e)[:200])


def decode_leg(dev, c2_args, n_beam_utts=2, beam=10, ctc_weight=0.3, maxlenratio=0.2, cpu_sample=4, threads=None):
    import espnet_amd
    from espnet_amd.nets.batch_beam_search import BatchBeamSearch
    from espnet_amd.nets.beam_search import BeamSearch
    from espnet_amd.nets.ctc_prefix_score import CTCPrefixScorer, LengthBonus
    from espnet_amd.nets.e2e_asr_conformer import E2E
    espnet_amd.set_precision("fp32")
    B, T, V = 32, 1000, 5000
    torch.manual_seed(0)
    model = E2E(80, V, c2_args(0.0))
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(11)
    xs = torch.randn(B, T, 80, generator=g)
    ilens = [T - 13 * i for i in range(B)]
    for i, n in enumerate(ilens):
        xs[i, n:] = 0.0
    xd = xs.to(dev)
    audio_s = sum(ilens) * 0.01
    out = {"workload": "BASELINE configs[1] model, random-init weights, 32 synthetic utterances T=1000..597 frames (%.0f s of audio)" % audio_s}

    # ---- (i) batched greedy CTC ----
    ids, n = model.greedy_ctc_batch(xd, ilens)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ids, n = model.greedy_ctc_batch(xd, ilens)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["greedy_ctc"] = dict(utt_per_s=round(B / dt, 1), rtf=float("%.3e" % (dt / audio_s)), ms_per_batch=round(dt * 1e3, 2),
                             d2h_copies_per_batch=0, note="ids and lengths stay on the device")
    ids_h, n_h = ids.cpu(), n.cpu()

    # ---- (ii) / (iii) beam searches on encoder outputs ----
    with torch.no_grad():
        from espnet_amd.nets.modules import make_non_pad_mask
        hs, _ = model.encoder(xd, make_non_pad_mask(ilens).unsqueeze(-2).to(dev))
    hl = [int(v) for v in __import__("espnet_amd.nets.modules", fromlist=["x"]).embed_output_lengths(model.encoder.embed, ilens, T)]
    weights = dict(decoder=1.0 - ctc_weight, ctc=ctc_weight, length_bonus=0.0)
    dev_best = {}
    for tag, cls in (("beam_search", BeamSearch), ("batch_beam_search", BatchBeamSearch)):
        scorers = dict(decoder=model.decoder, ctc=CTCPrefixScorer(model.ctc, model.eos), length_bonus=LengthBonus(V))
        bs = cls(scorers, weights, beam, V, model.sos, model.eos, pre_beam_score_key="full")
        enc0 = hs[0, : hl[0]].contiguous()
        bs(enc0, maxlenratio=0.02)               # warm-up (lazy buffers, first-launch costs)
        torch.cuda.synchronize()
        # launches and copies of a whole short search (16 beam steps), per beam step: the hypotheses, scores and scorer
        # states stay on the device; the host fetches the step log once every `sync_every` steps
        nstep = 16
        ratio = (nstep + 0.5) / enc0.shape[0]
        counts = _count_step(lambda: bs(enc0, maxlenratio=ratio))
        counts = {k: (round(v / nstep, 2) if isinstance(v, (int, float)) else v) for k, v in counts.items()}
        counts["device_loop"] = bool(bs._device_loop_ok(enc0))
        counts["sync_every"] = bs.sync_every
        steps = tot = 0
        t0 = time.perf_counter()
        for b in range(n_beam_utts):
            enc = hs[b, : hl[b]].contiguous()
            nb_dev = bs(enc, maxlenratio=maxlenratio)
            if b == 0:
                dev_best[tag] = (nb_dev[0].yseq.tolist(), float(nb_dev[0].score))
            steps += max(1, int(maxlenratio * hl[b]))
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        a_s = sum(ilens[:n_beam_utts]) * 0.01
        # all 32 utterances in ONE search (forward_batch: 32 x beam slots share every launch of a beam step)
        encs = [hs[b, : hl[b]].contiguous() for b in range(B)]
        bs.forward_batch(encs[:4], maxlenratio=0.02)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = bs.forward_batch(encs, maxlenratio=maxlenratio)
        torch.cuda.synchronize()
        tb = time.perf_counter() - t0
        bsteps = max(1, int(maxlenratio * max(hl)))
        batched = dict(utterances=B, utt_per_s=round(B / tb, 2), rtf=float("%.3e" % (tb / audio_s)), beam_steps=bsteps,
                       ms_per_beam_step=round(tb / bsteps * 1e3, 3), hypotheses_found=[len(u) for u in nb][:4],
                       note="BeamSearch.forward_batch: one device-resident search over all utterances")
        graph = _graph_leg(tag, n_beam_utts, beam, ctc_weight, maxlenratio)
        eager = dict(utt_per_s=round(n_beam_utts / tot, 2), rtf=float("%.3e" % (tot / a_s)), beam_steps=steps,
                     ms_per_beam_step=round(tot / steps * 1e3, 3))
        out[tag] = dict(beam=beam, ctc_weight=ctc_weight, maxlenratio=maxlenratio, utterances_timed=n_beam_utts, graph_steps=graph,
                        per_beam_step=counts, batched=batched, eager=eager,
                        note="encoder outputs precomputed (the greedy leg times the encoder); hypotheses, scores and scorer "
                             "states on the device, one device->host copy of the step log per sync_every steps")
        # the figures of the search as a corpus would be decoded - steps replayed as hipGraphs (graph_steps: two warm searches per
        # (utterances, beam, padded frames) signature, then replays; measured in the child process above) - where that leg ran;
        # the eager search (one utterance after another, every launch issued by the host) beside it under `eager`
        if isinstance(graph, dict) and graph.get("active") and "utt_per_s" in graph:
            out[tag].update(mode="graph_steps", utt_per_s=graph["utt_per_s"], ms_per_beam_step=graph["ms_per_beam_step"],
                            beam_steps=steps)
            if isinstance(graph.get("batched"), dict) and "utt_per_s" in graph["batched"]:
                out[tag]["batched"] = dict(batched, mode="graph_steps", utt_per_s=graph["batched"]["utt_per_s"],
                                           ms_per_beam_step=graph["batched"]["ms_per_beam_step"],
                                           eager=dict(utt_per_s=batched["utt_per_s"], ms_per_beam_step=batched["ms_per_beam_step"]))
        else:
            out[tag].update(mode="eager", **eager)

    # ---- CPU oracle beside it: greedy CTC of a bounded sample, ids compared ----
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import asr_oracle as oracle
    if threads:
        torch.set_num_threads(threads)
    cfg = dict(conformer=True, rel_pos=True, activation="swish", aheads=4, mtlalpha=0.3, lsm_weight=0.1, odim=V)
    k = cpu_sample
    t0 = time.perf_counter()
    with torch.no_grad():
        il = ilens[:k]
        hs_c, _ = oracle.encoder(sd_cpu, "encoder.", xs[:k, : max(il)], oracle.non_pad_mask(il).unsqueeze(-2), cfg, training=False)
        lg = oracle.linear(sd_cpu, "ctc.ctc_lo.", hs_c)
    hyp = [oracle.greedy_ctc(lg[b, : hl_b]) for b, hl_b in enumerate(
        __import__("espnet_amd.nets.modules", fromlist=["x"]).embed_output_lengths(model.encoder.embed, il, max(il)))]
    dt_c = time.perf_counter() - t0
    a_s = sum(il) * 0.01
    # the device ran the batch at T = 1000; the sample's longest utterance is utterance 0 (T = 1000): same shapes
    same = []
    top2 = lg.topk(2, dim=-1)
    gap = (top2.values[..., 0] - top2.values[..., 1]) / lg.abs().amax(dim=-1).clamp_min(1e-20)
    for b in range(k):
        clear = bool((gap[b, : len(lg[b])] > 1e-4)[: hl[b]].all())
        got = ids_h[b, : int(n_h[b])].tolist()
        same.append(got == hyp[b] if clear else None)
    out["cpu_baseline_decode"] = dict(kind="port", what="oracle greedy CTC (encoder + argmax + collapse), %d utterances" % k,
                                      cores=torch.get_num_threads(), utt_per_s=round(k / dt_c, 3), rtf=float("%.3e" % (dt_c / a_s)),
                                      ids_bit_exact=[s for s in same], note="None = an utterance with a near-tie frame (top-2 gap < 1e-4)")
    # ... and the CPU oracle's beam search (oracle.beam_search: the reference's BeamSearch algorithm, all running hypotheses
    # scored together, pinned to the reference's recorded searches by tests/test_oracle_golden.py) on utterance 0, same weights / cap
    t0 = time.perf_counter()
    with torch.no_grad():
        nb_c = oracle.beam_search(sd_cpu, hs_c[0, : hl[0]], cfg, weights, beam, maxlenratio, mode="ids")
    dt_b = time.perf_counter() - t0
    nsteps = max(1, int(maxlenratio * hl[0]))
    ref_y, ref_s = dev_best.get("beam_search", (None, float("nan")))
    out["cpu_baseline_decode"]["beam_search"] = dict(
        kind="port", what="oracle.beam_search (BeamSearch semantics, CTCPrefixScore numpy recursion), utterance 0 (T' = %d), beam %d, "
        "ctc_weight %.1f, maxlenratio %.1f, encoder output precomputed" % (hl[0], beam, ctc_weight, maxlenratio),
        cores=torch.get_num_threads(), utterances=1, utt_per_s=round(1.0 / dt_b, 3), rtf=float("%.3e" % (dt_b / (ilens[0] * 0.01))),
        beam_steps=nsteps, ms_per_beam_step=round(dt_b / nsteps * 1e3, 2), best_ids_equal_device=(nb_c[0]["yseq"] == ref_y),
        best_score_diff_vs_device=float("%.3e" % abs(nb_c[0]["score"] - ref_s)))
    del model
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2)
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    import json
    print(json.dumps(decode_leg(torch.device("cuda"), bench.c2_args, n_beam_utts=a.utts), indent=1))
