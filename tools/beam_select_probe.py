"""eamd_weighted_sum / eamd_beam_select against the tensor expressions they replace, bit for bit (the probe that showed hipcc fusing
a * b + c inside the kernel: DESIGN.md section 4, round 4).  usage: python tools/beam_select_probe.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from espnet_amd import ops
g = torch.Generator().manual_seed(3)
n, V, P, beam = 10, 5000, 15, 10
lp = torch.log_softmax(4 * torch.randn(n, V, generator=g), -1).cuda()
ones = torch.ones(n, V, device="cuda")
pre = ops.weighted_sum([lp, ones], [0.7, 0.1])
w = torch.zeros(n, V, device="cuda"); w += 0.7 * lp; w += 0.1 * ones
print("weighted_sum equal:", torch.equal(pre, w), (pre - w).abs().max().item())
ids = ops.topk_rows(pre, P)[1]
psi = (-5 - 3 * torch.rand(n, P, generator=g)).cuda()
c_s = (-1 - torch.rand(n, generator=g)).cuda()
hyp = (-10 * torch.rand(n, generator=g)).cuda()
top_s, top_i, cl = ops.beam_select(pre, ids, psi, c_s, hyp, 0.3, 1, beam)
c_local = psi - c_s[:, None]
print("c_local equal:", torch.equal(cl, c_local))
kept = torch.full_like(w, -float("inf"))
kept.scatter_(1, ids, torch.gather(w, 1, ids) + 0.3 * c_local)
kept += hyp[:, None]
s1, i1 = ops.topk_rows(kept, beam)
ts, i2 = ops.topk_rows(s1.view(1, beam * beam), beam)
ti = (i2 // beam) * V + i1.view(1, beam * beam).gather(1, i2)
print("top_s equal:", torch.equal(ts, top_s), (ts - top_s).abs().max().item(), "top_i equal:", torch.equal(ti, top_i))
cand_a = (torch.gather(w, 1, ids) + 0.3 * c_local) + hyp[:, None]
cand_fma = torch.addcmul(torch.gather(w, 1, ids).double(), c_local.double(), torch.tensor(float(torch.tensor(0.3, dtype=torch.float32)), dtype=torch.float64, device="cuda")).float() + hyp[:, None]
mine = torch.full_like(w, -float("inf"))
# reconstruct my kernel's candidate values from top_s / top_i
print("max of cand_a sorted top:", torch.sort(cand_a.view(-1), descending=True)[0][:10].tolist())
print("max of cand_fma sorted top:", torch.sort(cand_fma.view(-1), descending=True)[0][:10].tolist())
print("mine:", top_s.view(-1).tolist())
