#!/usr/bin/env python3
"""Phases of a graph_steps beam search at config-2 size, one progress line per phase (gpurun_out/sgd.log)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, espnet_amd
from espnet_amd.nets.beam_search import BeamSearch
from espnet_amd.nets.ctc_prefix_score import CTCPrefixScorer
from espnet_amd.nets.e2e_asr_conformer import E2E
from espnet_amd.nets.modules import make_non_pad_mask, embed_output_lengths

log = open(os.path.join(ROOT, "gpurun_out", "sgd.log"), "w")
def say(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)

espnet_amd.set_precision("fp32")
torch.manual_seed(0)
V, B, T = 5000, 4, 1000
model = E2E(80, V, bench.c2_args(0.0)).to("cuda").eval()
ilens = [1000, 990, 985, 980]
xs = torch.randn(B, T, 80, device="cuda")
with torch.no_grad():
    hs, _ = model.encoder(xs, make_non_pad_mask(ilens).unsqueeze(-2).to("cuda"))
hl = [int(v) for v in embed_output_lengths(model.encoder.embed, ilens, T)]
say("hl", hl)
scorers = dict(decoder=model.decoder, ctc=CTCPrefixScorer(model.ctc, model.eos))
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
gs = BeamSearch(scorers, dict(decoder=0.7, ctc=0.3), 10, V, model.sos, model.eos, pre_beam_score_key="full")
gs.graph_steps = True
ratio = float(os.environ.get("SGD_RATIO", "0.05"))
for rnd in range(4):
    b = rnd % B
    say("search", rnd, "utt", b, "start (1st eager on padded memory, 2nd captures, then replays)")
    out = gs(hs[b, : hl[b]].contiguous(), maxlenratio=ratio)
    torch.cuda.synchronize()
    say("search", rnd, "done: best", out[0].yseq.tolist()[:8], "score %.4f" % float(out[0].score), "graphs on:", gs.graph_steps)
    if mode == "eager" and rnd == 0:
        break
if mode in ("batch", "all2"):
    NB = int(os.environ.get("SGD_NB", "4"))
    ilens2 = [1000 - 3 * i for i in range(NB)]
    xs2 = torch.randn(NB, T, 80, device="cuda")
    with torch.no_grad():
        hs2, _ = model.encoder(xs2, make_non_pad_mask(ilens2).unsqueeze(-2).to("cuda"))
    hl2 = [int(v) for v in embed_output_lengths(model.encoder.embed, ilens2, T)]
    encs = [hs2[b, : hl2[b]].contiguous() for b in range(NB)]
    for rnd in range(3):
        order = encs[rnd:] + encs[:rnd]
        say("batched search", rnd, "of", NB, "utterances: start")
        nb = gs.forward_batch(order, maxlenratio=ratio)
        torch.cuda.synchronize()
        say("batched search", rnd, "done: best of first", nb[0][0].yseq.tolist()[:6], "graphs on:", gs.graph_steps)
say("finished")
