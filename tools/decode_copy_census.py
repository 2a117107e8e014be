"""which host-side tensor expressions of a beam step become device-to-device copies (hipMemcpyAsync nodes in a step graph): counts per
step with the Python frames that issue them.  GPU box: python tools/decode_copy_census.py"""
import sys, os, torch, collections
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
bs(enc, maxlenratio=0.02)
from torch.profiler import profile, ProfilerActivity
nstep = 16
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    bs(enc, maxlenratio=(nstep + 0.5) / enc.shape[0]); torch.cuda.synchronize()
dev = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA and ("Memcpy" in ev.name or "Memset" in ev.name or "copyBuffer" in ev.name):
        dev[ev.name[:60]] += 1
print("device copies per step:", {k: round(v / nstep, 2) for k, v in dev.items()})
rows = prof.key_averages(group_by_stack_n=8)
for r in sorted(rows, key=lambda r: -r.count):
    if r.key in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::cat", "aten::index_select", "aten::index", "aten::fill_", "aten::zero_"):
        st = [s for s in r.stack if "espnet_amd" in s or "tools/" in s][:4]
        print("%6.2f %-20s %s" % (r.count / nstep, r.key, " <- ".join(s.split("/")[-1][:60] for s in st)))
