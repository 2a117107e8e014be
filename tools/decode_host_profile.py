import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import bench, espnet_amd
from espnet_amd.nets.beam_search import BeamSearch
from espnet_amd.nets.ctc_prefix_score import CTCPrefixScorer, LengthBonus
from espnet_amd.nets.e2e_asr_conformer import E2E
espnet_amd.set_precision("fp32")
torch.manual_seed(0)
V = 5000
model = E2E(80, V, bench.c2_args(0.0)).to("cuda").eval()
x = torch.randn(1000, 80, device="cuda")
with torch.no_grad():
    enc = model.encode(x) if hasattr(model, "encode") else None
enc = enc if isinstance(enc, torch.Tensor) else torch.as_tensor(enc, device="cuda")
scorers = model.scorers(); scorers["length_bonus"] = LengthBonus(V)
bs = BeamSearch(beam_size=10, vocab_size=V, weights=dict(decoder=0.7, ctc=0.3, length_bonus=0.0), scorers=scorers, sos=V - 1, eos=V - 1,
                token_list=None, pre_beam_score_key="full")
with torch.no_grad():
    bs.forward(enc, maxlenratio=0.2)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    bs.forward(enc, maxlenratio=0.2)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
