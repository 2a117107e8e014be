import sys, os, torch, collections
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
import bench, espnet_amd
from espnet_amd.nets.beam_search import BeamSearch
from espnet_amd.nets.ctc_prefix_score import CTCPrefixScorer
from espnet_amd.nets.e2e_asr_conformer import E2E
espnet_amd.set_precision("fp32"); torch.manual_seed(0)
V=5000
model=E2E(80,V,bench.c2_args(0.0)).to("cuda").eval()
x=torch.randn(1,1000,80,device="cuda")
from espnet_amd.nets.modules import make_non_pad_mask
with torch.no_grad():
    hs,_=model.encoder(x, make_non_pad_mask([1000]).unsqueeze(-2).to("cuda"))
enc=hs[0].contiguous()
bs=BeamSearch(dict(decoder=model.decoder, ctc=CTCPrefixScorer(model.ctc, model.eos)), dict(decoder=0.7, ctc=0.3), 10, V, model.sos, model.eos, pre_beam_score_key="full")
bs(enc, maxlenratio=0.02)
from torch.profiler import profile, ProfilerActivity
nstep=16
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    bs(enc, maxlenratio=(nstep+0.5)/enc.shape[0]); torch.cuda.synchronize()
c=collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in ev.name and "Memset" not in ev.name:
        c[ev.name[:100]]+=1
tot=sum(c.values())
print("kernels per step %.1f" % (tot/nstep))
for k,v in c.most_common(40): print("%6.2f  %s" % (v/nstep, k))
