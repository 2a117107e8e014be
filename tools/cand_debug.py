import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_model import c2width_setup
from espnet_amd.nets.beam_search import BeamSearch
from espnet_amd.nets.ctc_prefix_score import LengthBonus
from espnet_amd.nets.modules import Decoder
SW, model, g, encs = c2width_setup()
spec = SW.DECODE_R4
def run(sel, src):
    import espnet_amd.ops as ops
    scorers = model.scorers(); scorers["length_bonus"] = LengthBonus(spec["odim"])
    bs = BeamSearch(scorers, dict(decoder=0.7, ctc=0.3, length_bonus=0.1), spec["beam"], spec["odim"], model.sos, model.eos, pre_beam_score_key="full")
    bs.candidate_select = sel
    old = ops.decode_src_attn
    if not src:
        ops.decode_src_attn = lambda *a, **k: None
    try:
        nb = bs(encs[1], maxlenratio=0.2)
    finally:
        ops.decode_src_attn = old
    return [(h.yseq.tolist(), float(h.score)) for h in nb[:10]]
for src in (True, False):
    a, b, a2 = run(True, src), run(False, src), run(True, src)
    print("src kernel", src, "sel T==F:", a == b, "T==T:", a == a2)
    for k, (x, y) in enumerate(zip(a, b)):
        if x != y:
            print("  first diff at", k, x[1], y[1], x[0] == y[0]); break

import espnet_amd.ops as ops
logs = {}
orig_bf = ops.beam_finish
def rec_run(sel):
    log = []
    def spy(top_s, top_i, *a, **k):
        log.append((top_s.clone(), top_i.clone()))
        return orig_bf(top_s, top_i, *a, **k)
    ops.beam_finish = spy
    try:
        run(sel, True)
    finally:
        ops.beam_finish = orig_bf
    return log
la, lb = rec_run(True), rec_run(False)
for i, ((sa, ia), (sb, ib)) in enumerate(zip(la, lb)):
    if not (torch.equal(sa.view(-1), sb.view(-1)) and torch.equal(ia.view(-1), ib.view(-1))):
        print("first differing step", i, (sa.view(-1) - sb.view(-1)).abs().max().item(), torch.equal(ia.view(-1), ib.view(-1)))
        print(sa.view(-1).tolist()); print(sb.view(-1).tolist()); print(ia.view(-1).tolist()); print(ib.view(-1).tolist())
        break
else:
    print("all steps equal", len(la), len(lb))
