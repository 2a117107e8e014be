#!/usr/bin/env python3
"""Child process of test_beam_search_step_graphs: eager vs graph_steps searches, exit code 0 when they agree.
usage: step_graph_check.py <batch 0|1> <lm None|tlm|rlm|dlm> <ctc weight> <lm weight>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_model import DEV, _fusion_models  # noqa: E402


def c2width():
    """the reference-recorded searches at config 2's width through graph_steps: eager, capture, replay of every search"""
    from test_gpu_model import c2width_compare, c2width_setup
    from espnet_amd.nets.batch_beam_search import BatchBeamSearch
    from espnet_amd.nets.beam_search import BeamSearch
    from espnet_amd.nets.ctc_prefix_score import LengthBonus
    SW, model, g, encs = c2width_setup()
    spec = SW.DECODE_R4
    n_graphs = 0
    for cw, ratio, pen in ((0.3, 0.2, 0.1), (0.3, 0.0, 0.0), (1.0, 0.2, 0.1)):
        for cls, nm in ((BeamSearch, "beam"), (BatchBeamSearch, "bbeam")):
            scorers = model.scorers()
            scorers["length_bonus"] = LengthBonus(spec["odim"])
            bs = cls(scorers, dict(decoder=1.0 - cw, ctc=cw, length_bonus=pen), spec["beam"], spec["odim"], model.sos, model.eos,
                     pre_beam_score_key=None if cw == 1.0 else "full")
            bs.graph_steps = True
            tags = ["u%d_%s_w%02d_r%02d" % (u, nm, int(cw * 10), int(ratio * 10)) for u in range(len(encs))]
            for rnd in range(3):                                  # eager, capture + replay, replay
                for u, enc in enumerate(encs):
                    c2width_compare("graph_steps[%d] %s" % (rnd, cls.__name__), bs(enc, maxlenratio=ratio), g, tags[u])
                together = bs.forward_batch(encs, maxlenratio=ratio)
                for u in range(len(encs)):
                    c2width_compare("graph_steps[%d] %s.forward_batch" % (rnd, cls.__name__), together[u], g, tags[u])
            assert bs.graph_steps, "a step could not be captured: the searches above ran eagerly"
            n_graphs += sum(len(G["graphs"]) + (1 if G.get("dyn") else 0) for G in bs._step_graphs.values())
            # with the candidate-selection kernels (pre-beam, CTC weight 0.3) every step >= 1 replays ONE graph that reads the step index
            # from the device; the tensor-expression path (CTC only, no pre-beam) keeps one graph per step
            one = [bool(G.get("dyn")) for G in bs._step_graphs.values() if G["searches"] >= 2 and len(G["graphs"]) > 0]
            if cw == 0.3 and ratio > 0:
                assert one and all(one), one
                assert all(len(G["graphs"]) == 1 for G in bs._step_graphs.values() if G.get("dyn"))
            elif cw == 1.0:
                assert not any(one), one
            bs._step_graphs = {}
    assert n_graphs > 0
    print("[parity] step graphs c2width: %d captured steps" % n_graphs)


def main():
    if sys.argv[1] == "c2width":
        return c2width()
    batch, lm, cw, lw = bool(int(sys.argv[1])), (None if sys.argv[2] == "None" else sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
    from espnet_amd.nets.batch_beam_search import BatchBeamSearch
    from espnet_amd.nets.beam_search import BeamSearch
    from espnet_amd.nets.ctc_prefix_score import CTCPrefixScorer, LengthBonus
    p, model, lms = _fusion_models()
    with torch.no_grad():
        enc, _ = model.encode(p["speech"].unsqueeze(0).to(DEV), torch.tensor([p["speech"].shape[0]]))
    x = enc[0]
    T = x.shape[0]
    scorers = dict(decoder=model.decoder, ctc=CTCPrefixScorer(model.ctc, model.eos) if cw > 0 else None, length_bonus=LengthBonus(30),
                   lm=lms[lm] if lm else None)
    cls = BatchBeamSearch if batch else BeamSearch
    mk = lambda: cls(scorers, dict(decoder=1.0 - cw, ctc=cw, lm=lw, length_bonus=0.1), 4, 30, model.sos, model.eos,  # noqa: E731
                     pre_beam_score_key=None if cw in (0.0, 1.0) else "full")
    eager, graphed = mk(), mk()
    graphed.graph_steps, graphed.graph_frame_bucket = True, 16
    lo = (T - 1) // 16 * 16 + 1                 # lengths lo .. T share the bucket of T
    variants = [x, (x * 1.3).contiguous(), x.flip(0).contiguous(), x[: max(lo, T - 2)].contiguous(), (x[: max(lo, T - 1)] * 0.8).contiguous()]
    for ratio in (0.5, 0.0):
        for rnd, u in enumerate(variants):                              # eager, capture, replay, replay, replay
            a, g = eager(u, maxlenratio=ratio), graphed(u, maxlenratio=ratio)
            assert [h.yseq.tolist() for h in g[:3]] == [h.yseq.tolist() for h in a[:3]], (ratio, rnd)
            for ha, hg in zip(a[:3], g[:3]):
                assert abs(float(ha.score) - float(hg.score)) <= 1e-4 * max(1.0, abs(float(ha.score))), (ratio, rnd)
        sets = [variants[:3], variants[2:5], [variants[4], variants[0], variants[3]], variants[1:4]]
        for rnd, us in enumerate(sets):                                 # three utterances per search: eager, capture, replay, replay
            a, g = eager.forward_batch(us, maxlenratio=ratio), graphed.forward_batch(us, maxlenratio=ratio)
            for b in range(len(us)):
                assert [h.yseq.tolist() for h in g[b][:3]] == [h.yseq.tolist() for h in a[b][:3]], (ratio, rnd, b)
                for ha, hg in zip(a[b][:3], g[b][:3]):
                    assert abs(float(ha.score) - float(hg.score)) <= 1e-4 * max(1.0, abs(float(ha.score))), (ratio, rnd, b)
    assert graphed.graph_steps, "a step could not be captured: the searches above ran eagerly"
    n_graphs = sum(len(G["graphs"]) for G in graphed._step_graphs.values())
    assert n_graphs > 0
    # least-recently-used eviction: with room for ONE signature, alternating two signatures never replays a stale graph
    small = mk()
    small.graph_steps, small.graph_frame_bucket, small.graph_max_signatures = True, 16, 1
    for rnd in range(4):
        for us in ([variants[0]], variants[:2]):
            a, g = eager.forward_batch(us, maxlenratio=0.5) if len(us) > 1 else [eager(us[0], maxlenratio=0.5)], \
                small.forward_batch(us, maxlenratio=0.5)
            assert [h.yseq.tolist() for h in g[0][:2]] == [h.yseq.tolist() for h in a[0][:2]], ("lru", rnd, len(us))
            assert len(small._step_graphs) <= 1
    print("[parity] step graphs %s lm=%s: %d signatures, %d captured steps" % (cls.__name__, lm, len(graphed._step_graphs), n_graphs))


if __name__ == "__main__":
    main()
