"""single-utterance beam searches with step graphs, for a rocprofv3 kernel trace of the REPLAYED steps (tools/trace_decode_graph.sh):
config 2's decoder (6 blocks, d 256, ff 2048, V 5000), T = 1000 frames in, beam 10, CTC weight 0.3; prints the wall time per step"""
import sys, os, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import bench, espnet_amd
from espnet_amd.nets.beam_search import BeamSearch
from espnet_amd.nets.ctc_prefix_score import CTCPrefixScorer
from espnet_amd.nets.e2e_asr_conformer import E2E
from espnet_amd.nets.modules import make_non_pad_mask
espnet_amd.set_precision("fp32"); torch.manual_seed(0)
V = 5000
model = E2E(80, V, bench.c2_args(0.0)).to("cuda").eval()
x = torch.randn(1, 1000, 80, device="cuda")
with torch.no_grad():
    hs, _ = model.encoder(x, make_non_pad_mask([1000]).unsqueeze(-2).to("cuda"))
enc = hs[0].contiguous()
bs = BeamSearch(dict(decoder=model.decoder, ctc=CTCPrefixScorer(model.ctc, model.eos)), dict(decoder=0.7, ctc=0.3), 10, V, model.sos,
                model.eos, pre_beam_score_key="full")
bs.graph_steps = True
nstep = int(os.environ.get("EAMD_TRACE_STEPS", "24"))
ratio = (nstep + 0.5) / enc.shape[0]
nutt = int(os.environ.get("EAMD_TRACE_BATCH", "1"))          # > 1: that many utterances per search (forward_batch)
for it in range(4):          # eager, capture, replay, replay
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nb = bs(enc, maxlenratio=ratio) if nutt == 1 else bs.forward_batch([enc] * nutt, maxlenratio=ratio)[0]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("search %d: %.3f ms per step (%d steps, %d utterances), best len %d" % (it, (t1 - t0) * 1e3 / nstep, nstep, nutt, len(nb[0].yseq)), flush=True)
