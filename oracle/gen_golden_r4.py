#!/usr/bin/env python3
"""Round-4 fixture, by RUNNING THE REFERENCE (kan-bayashi/espnet v0.9.5, PyTorch CPU) in the build container:

  decode_c2width.npz   joint CTC/attention beam search at BASELINE config 2's WIDTH: espnet1 Conformer E2E with adim 256,
                       aheads 4 (d_k = 64), eunits = dunits = 2048, |V| = 5000, 2 encoder + 2 decoder layers, three utterances
                       of 1000 / 640 / 300 frames (T' = 249 / 159 / 74), beam 10, ctc_weight {0, 0.3, 1} x maxlenratio {0, 0.2}
                       (length bonus 0 / 0.1), searched by the reference's BeamSearch (beam_search.py:36-458, CTCPrefixScore
                       numpy recursion) AND its BatchBeamSearch (batch_beam_search.py:31-348, CTCPrefixScoreTH): 5-best token
                       sequences, total and per-scorer scores, number of ended hypotheses; plus samples of the encoder output
                       and of the CTC posteriors.

Weights and inputs are NOT stored: both sides build them from oracle/seeded_weights.py (DECODE_R4: name-keyed generator,
sharpened output layers, <eos> / blank biases).  Usage: python oracle/gen_golden_r4.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import install_stubs, save  # noqa: E402
import seeded_weights as SW  # noqa: E402

NBEST = 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    a = ap.parse_args()
    install_stubs()
    sys.path.insert(0, a.ref)
    torch.set_num_threads(8)
    from espnet.nets.batch_beam_search import BatchBeamSearch
    from espnet.nets.beam_search import BeamSearch
    from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E
    from espnet.nets.scorers.length_bonus import LengthBonus

    spec = SW.DECODE_R4
    model = SW.decode_r4_model(E2E)
    rec = {}
    with torch.no_grad():
        for u, x in enumerate(SW.decode_r4_inputs()):
            enc = model.encode(x.numpy())
            logp = model.ctc.log_softmax(enc.unsqueeze(0))[0]
            rec["u%d_enc" % u] = enc[::8].clone()
            rec["u%d_logp" % u] = logp[::8, ::50].clone()
            rec["u%d_ctc_argmax" % u] = logp.argmax(-1)
            for cw, ratio, pen in SW.DECODE_R4_CASES:
                for cls in (BeamSearch, BatchBeamSearch):
                    scorers = model.scorers()
                    scorers["length_bonus"] = LengthBonus(spec["odim"])
                    bs = cls(beam_size=spec["beam"], vocab_size=spec["odim"],
                             weights=dict(decoder=1.0 - cw, ctc=cw, length_bonus=pen), scorers=scorers, sos=model.sos,
                             eos=model.eos, token_list=None, pre_beam_score_key=None if cw == 1.0 else "full")
                    t0 = time.time()
                    nb_all = bs(x=enc, maxlenratio=ratio, minlenratio=0.0)
                    nb = nb_all[:NBEST]
                    tag = "u%d_%s_w%02d_r%02d" % (u, "bbeam" if cls is BatchBeamSearch else "beam", int(cw * 10), int(ratio * 10))
                    rec[tag + "_scores"] = np.asarray([float(h.score) for h in nb], dtype=np.float64)
                    rec[tag + "_lens"] = np.asarray([len(h.yseq) for h in nb], dtype=np.int64)
                    rec[tag + "_yseq"] = np.asarray(sum([[int(t) for t in h.yseq] for h in nb], []), dtype=np.int64)
                    rec[tag + "_nended"] = np.asarray(len(nb_all), dtype=np.int64)
                    for k in sorted(nb[0].scores):
                        rec[tag + "_sc_" + k] = np.asarray([float(h.scores[k]) for h in nb], dtype=np.float64)
                    rec[tag + "_cpu_seconds"] = np.asarray(time.time() - t0)
                    print(tag, "%.1f s" % (time.time() - t0), len(nb_all), rec[tag + "_lens"].tolist(),
                          np.round(rec[tag + "_scores"], 3).tolist(), flush=True)
    save(os.path.join(a.out, "decode_c2width.npz"), **rec)


if __name__ == "__main__":
    main()
