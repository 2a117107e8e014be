#!/usr/bin/env python3
"""Round-4 fixture (c), by RUNNING THE REFERENCE (kan-bayashi/espnet v0.9.5, PyTorch CPU) in the build container:

  decode_c2width_long.npz   the joint CTC/attention beam search of decode_c2width.npz (same model: oracle/seeded_weights.py DECODE_R4)
                            on a LONG memory - the reference encoder's outputs of utterances 0, 1, 0 back to back, 657 frames -
                            by the reference's BeamSearch and BatchBeamSearch, ctc_weight 0.3, maxlenratio 0.2, length bonus 0.1:
                            5-best token sequences, total and per-scorer scores.  (More than 512 frames: the CTC candidate
                            scores and the survivors' scan of csrc/ctc.hip run with 16 frames per lane there.)

Usage: python oracle/gen_golden_r4c.py [--ref /root/reference] [--out tests/golden]
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
        encs = [model.encode(x.numpy()) for x in SW.decode_r4_inputs()]
        enc = torch.cat([encs[0], encs[1], encs[0]], 0)
        rec["enc_sample"] = enc[::16].clone()
        cw, ratio, pen = 0.3, 0.2, 0.1
        for cls in (BeamSearch, BatchBeamSearch):
            scorers = model.scorers()
            scorers["length_bonus"] = LengthBonus(spec["odim"])
            bs = cls(beam_size=spec["beam"], vocab_size=spec["odim"], weights=dict(decoder=1.0 - cw, ctc=cw, length_bonus=pen),
                     scorers=scorers, sos=model.sos, eos=model.eos, token_list=None, pre_beam_score_key="full")
            t0 = time.time()
            nb_all = bs(x=enc, maxlenratio=ratio, minlenratio=0.0)
            nb = nb_all[:NBEST]
            tag = "long_%s" % ("bbeam" if cls is BatchBeamSearch else "beam")
            rec[tag + "_scores"] = np.asarray([float(h.score) for h in nb], dtype=np.float64)
            rec[tag + "_lens"] = np.asarray([len(h.yseq) for h in nb], dtype=np.int64)
            rec[tag + "_yseq"] = np.asarray(sum([[int(t) for t in h.yseq] for h in nb], []), dtype=np.int64)
            rec[tag + "_nended"] = np.asarray(len(nb_all), dtype=np.int64)
            for k in sorted(nb[0].scores):
                rec[tag + "_sc_" + k] = np.asarray([float(h.scores[k]) for h in nb], dtype=np.float64)
            print(tag, "%.1f s" % (time.time() - t0), len(nb_all), rec[tag + "_lens"].tolist(), np.round(rec[tag + "_scores"], 3).tolist(),
                  flush=True)
    save(os.path.join(a.out, "decode_c2width_long.npz"), **rec)


if __name__ == "__main__":
    main()
