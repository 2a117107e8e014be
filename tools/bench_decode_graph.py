#!/usr/bin/env python3
"""Child of tools/bench_decode.py: BeamSearch with graph_steps on utterances of one frame bucket (config-2 model, random-init
weights) - one eager search, one capturing search, then the timed replays; prints one JSON line.
usage: bench_decode_graph.py <utterances timed> <beam> <ctc weight> <maxlenratio> [beam_search | batch_beam_search]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_timed, beam, cw, ratio = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
    which = sys.argv[5] if len(sys.argv) > 5 else "beam_search"
    import bench
    import espnet_amd
    if which == "batch_beam_search":
        from espnet_amd.nets.batch_beam_search import BatchBeamSearch as BeamSearch
    else:
        from espnet_amd.nets.beam_search import BeamSearch
    from espnet_amd.nets.ctc_prefix_score import CTCPrefixScorer
    from espnet_amd.nets.e2e_asr_conformer import E2E
    from espnet_amd.nets.modules import embed_output_lengths, make_non_pad_mask
    espnet_amd.set_precision("fp32")
    dev = torch.device("cuda")
    V, T = 5000, 1000
    torch.manual_seed(0)
    model = E2E(80, V, bench.c2_args(0.0)).to(dev).eval()
    nu = n_timed + 2
    ilens = [T - 3 * i for i in range(nu)]                     # 249 .. 24x frames after subsampling: one bucket of 32
    xs = torch.randn(nu, T, 80, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        hs, _ = model.encoder(xs, make_non_pad_mask(ilens).unsqueeze(-2).to(dev))
    hl = [int(v) for v in embed_output_lengths(model.encoder.embed, ilens, T)]
    mk = lambda: BeamSearch(dict(decoder=model.decoder, ctc=CTCPrefixScorer(model.ctc, model.eos)),  # noqa: E731
                            dict(decoder=1.0 - cw, ctc=cw), beam, V, model.sos, model.eos, pre_beam_score_key="full")
    eager, gs = mk(), mk()
    gs.graph_steps = True
    encs = [hs[b, : hl[b]].contiguous() for b in range(nu)]
    eager(encs[0], maxlenratio=0.02)
    for e in encs[-2:]:                                         # eager on the static buffers, then the capturing search
        gs(e, maxlenratio=ratio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = [gs(e, maxlenratio=ratio) for e in encs[:n_timed]]
    torch.cuda.synchronize()
    tg = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = [eager(e, maxlenratio=ratio) for e in encs[:n_timed]]
    torch.cuda.synchronize()
    te = time.perf_counter() - t0
    steps = sum(max(1, int(ratio * hl[b])) for b in range(n_timed))
    # 32 utterances in one search (forward_batch): the bench's utterance lengths, three searches in rotated orders
    NB = 32
    il2 = [int(v) for v in torch.linspace(T, 0.6 * T, NB).round().tolist()]
    xs2 = torch.randn(NB, T, 80, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        hs2, _ = model.encoder(xs2, make_non_pad_mask(il2).unsqueeze(-2).to(dev))
    hl2 = [int(v) for v in embed_output_lengths(model.encoder.embed, il2, T)]
    e2 = [hs2[b, : hl2[b]].contiguous() for b in range(NB)]
    rot = lambda k: e2[k:] + e2[:k]  # noqa: E731
    gs.forward_batch(rot(1), maxlenratio=ratio)
    gs.forward_batch(rot(2), maxlenratio=ratio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nbg = gs.forward_batch(e2, maxlenratio=ratio)
    torch.cuda.synchronize()
    tgb = time.perf_counter() - t0
    t0 = time.perf_counter()
    nbe = eager.forward_batch(e2, maxlenratio=ratio)
    torch.cuda.synchronize()
    teb = time.perf_counter() - t0
    bsteps = max(1, int(ratio * max(hl2)))
    batched = dict(utterances=NB, utt_per_s=round(NB / tgb, 2), ms_per_beam_step=round(tgb / bsteps * 1e3, 3),
                   eager_same_process=dict(utt_per_s=round(NB / teb, 2), ms_per_beam_step=round(teb / bsteps * 1e3, 3)),
                   same_best_as_eager=sum(int(a[0].yseq.tolist() == b_[0].yseq.tolist()) for a, b_ in zip(nbg, nbe)))
    print(json.dumps(dict(batched=batched, active=bool(gs.graph_steps), frame_bucket=gs.graph_frame_bucket, utterances_timed=n_timed,
                          utt_per_s=round(n_timed / tg, 2), ms_per_beam_step=round(tg / steps * 1e3, 3),
                          eager_same_process=dict(utt_per_s=round(n_timed / te, 2), ms_per_beam_step=round(te / steps * 1e3, 3)),
                          same_best_as_eager=[g[0].yseq.tolist() == r[0].yseq.tolist() for g, r in zip(got, ref)],
                          best_score_diff=[abs(float(g[0].score) - float(r[0].score)) for g, r in zip(got, ref)],
                          captured_steps=sum(len(G["graphs"]) for G in gs._step_graphs.values()),
                          note="single-utterance searches, steps replayed as hipGraphs after one eager and one capturing search "
                               "of the (utterances, beam, padded frames) signature")))


if __name__ == "__main__":
    main()
