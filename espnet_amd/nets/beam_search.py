"""Joint CTC/attention beam search for one utterance (hypotheses batched through the HIP kernels).

reference semantics: espnet/nets/beam_search.py:36-458 (BeamSearch: full scorers, pre-beam of
int(1.5*beam) on the full score, partial (CTC prefix) scorer on the pre-beam, weighted sum, top-`beam`
over all expansions, <eos> handling in post_process), espnet/nets/e2e_asr_common.py:21-51 (end_detect),
legacy E2E.recognize options (e2e_asr_transformer.py:259-477: ctc_weight, penalty, maxlenratio,
minlenratio, nbest).  Utterances are independent: decode many by running one search per utterance on
separate streams / GPUs (SURVEY.md §8e "replicas only").
"""
import math

import numpy as np
import torch

from .ctc_prefix_score import CTCPrefixScorer
from .modules import subsequent_mask


class Hypothesis:
    __slots__ = ("yseq", "score", "scores", "dec_state", "ctc_state")

    def __init__(self, yseq, score, scores, dec_state, ctc_state):
        self.yseq, self.score, self.scores, self.dec_state, self.ctc_state = yseq, score, scores, dec_state, ctc_state

    def asdict(self):
        return {"yseq": list(self.yseq), "score": float(self.score), "scores": dict(self.scores)}


def end_detect(ended_hyps, i, M=3, D_end=math.log(1 * math.exp(-10))):
    """reference: e2e_asr_common.py:21-51"""
    if len(ended_hyps) == 0:
        return False
    best = max(h["score"] for h in ended_hyps)
    count = 0
    for m in range(M):
        same = [h["score"] for h in ended_hyps if len(h["yseq"]) == i - m]
        if same and max(same) - best < D_end:
            count += 1
    return count == M


class BeamSearch:
    def __init__(self, decoder, ctc_scorer, weights, beam_size, vocab_size, sos, eos, pre_beam_ratio=1.5):
        self.decoder = decoder if weights.get("decoder", 0) != 0 else None
        self.ctc = ctc_scorer if (ctc_scorer is not None and weights.get("ctc", 0) != 0) else None
        self.weights = weights
        self.beam_size, self.n_vocab, self.sos, self.eos = beam_size, vocab_size, sos, eos
        self.pre_beam_size = int(pre_beam_ratio * beam_size)
        # pre-beam on the full (attention) score, as asr_inference / recog_v2 configure it
        self.do_pre_beam = self.decoder is not None and self.ctc is not None and self.pre_beam_size < vocab_size

    def _expand(self, hyps, x):
        n, V, dev = len(hyps), self.n_vocab, x.device
        w = self.weights
        weighted = torch.zeros(n, V, device=dev, dtype=torch.float32)
        dec_logp = new_dec_states = None
        if self.decoder is not None:
            ys = torch.tensor([h.yseq for h in hyps], dtype=torch.int64).to(dev)
            dec_logp, new_dec_states = self.decoder.batch_score(ys, [h.dec_state for h in hyps],
                                                                x.unsqueeze(0).expand(n, *x.shape).contiguous())
            weighted += w["decoder"] * dec_logp
        if w.get("length_bonus", 0) != 0:
            weighted += w["length_bonus"]
        cand = psi = r_new = ctc_delta = None
        if self.ctc is not None:
            if self.do_pre_beam:
                cand = torch.topk(weighted, self.pre_beam_size, dim=1)[1].to(torch.int32)
            else:
                cand = torch.arange(V, device=dev, dtype=torch.int32).unsqueeze(0).expand(n, V).contiguous()
            ctc_delta, (psi, r_new) = self.ctc.batch_score_partial([h.yseq for h in hyps], cand,
                                                                   [h.ctc_state for h in hyps])
            if self.do_pre_beam:   # tokens outside the pre-beam are dropped (reference: beam())
                masked = torch.full_like(weighted, -float("inf"))
                masked.scatter_(1, cand.long(), torch.gather(weighted, 1, cand.long()) + w["ctc"] * ctc_delta)
                weighted = masked
            else:
                weighted += w["ctc"] * ctc_delta
        weighted += torch.tensor([h.score for h in hyps], dtype=torch.float32).to(dev)[:, None]
        # host side: pick the global top-`beam` expansions (hypothesis bookkeeping is Python in the reference too)
        k = min(self.beam_size, V)
        top_s, top_i = torch.topk(weighted, k, dim=1)
        top_s, top_i = top_s.cpu().numpy(), top_i.cpu().numpy()
        flat = [(float(top_s[a, b]), a, int(top_i[a, b])) for a in range(n) for b in range(k)
                if np.isfinite(top_s[a, b])]
        flat.sort(key=lambda t: -t[0])
        flat = flat[: self.beam_size]
        dec_np = dec_logp.cpu().numpy() if dec_logp is not None else None
        cand_np = cand.cpu().numpy() if cand is not None else None
        delta_np = ctc_delta.cpu().numpy() if ctc_delta is not None else None
        psi_np = psi.cpu().numpy() if psi is not None else None
        out = []
        for score, a, tok in flat:
            h = hyps[a]
            scores = dict(h.scores)
            if dec_np is not None:
                scores["decoder"] = scores.get("decoder", 0.0) + float(dec_np[a, tok])
            if w.get("length_bonus", 0) != 0:
                scores["length_bonus"] = scores.get("length_bonus", 0.0) + 1.0
            ctc_state = None
            if cand_np is not None:
                j = int(np.nonzero(cand_np[a] == tok)[0][0])
                scores["ctc"] = scores.get("ctc", 0.0) + float(delta_np[a, j])
                ctc_state = (float(psi_np[a, j]), r_new[a, j])
            out.append(Hypothesis(h.yseq + [tok], score, scores,
                                  new_dec_states[a] if new_dec_states is not None else None, ctc_state))
        return out

    def __call__(self, x, maxlenratio=0.0, minlenratio=0.0):
        """x: (T, D) encoder output.  Returns the ended hypotheses, best first."""
        T = x.shape[0]
        maxlen = T if maxlenratio == 0 else max(1, int(maxlenratio * T))
        minlen = int(minlenratio * T)
        init_scores = {}
        ctc_state = self.ctc.init_state(x) if self.ctc is not None else None
        running = [Hypothesis([self.sos], 0.0, init_scores, None, ctc_state)]
        ended = []
        with torch.no_grad():
            for i in range(maxlen):
                best = self._expand(running, x)
                if i == maxlen - 1:      # force <eos> at the last position (beam_search.py:436-441)
                    for h in best:
                        h.yseq = h.yseq + [self.eos]
                running = []
                for h in best:
                    if h.yseq[-1] == self.eos:     # v0.9.5 post_process applies no minlen filter
                        ended.append(h)
                    else:
                        running.append(h)
                if maxlenratio == 0.0 and end_detect([h.asdict() for h in ended], i):
                    break
                if not running:
                    break
        ended.sort(key=lambda h: -h.score)
        return ended


def recognize_beam(model, enc_output, recog_args, char_list=None, rnnlm=None):
    """E2E.recognize for ctc_weight < 1 (reference: e2e_asr_transformer.py:286-477 / recog_v2).
    Returns [{"score": float, "yseq": [int]}] n-best, yseq starts with <sos> and ends with <eos>."""
    if rnnlm is not None:
        raise NotImplementedError("LM fusion is the next row of the scope table (SURVEY.md §8f)")
    ctc_weight = float(getattr(recog_args, "ctc_weight", 0.0))
    if model.ctc is None:
        ctc_weight = 0.0
    weights = dict(decoder=1.0 - ctc_weight, ctc=ctc_weight, length_bonus=float(getattr(recog_args, "penalty", 0.0)))
    scorer = CTCPrefixScorer(model.ctc, model.eos) if ctc_weight > 0 else None
    bs = BeamSearch(model.decoder, scorer, weights, int(recog_args.beam_size), model.odim, model.sos, model.eos)
    hyps = bs(enc_output, float(getattr(recog_args, "maxlenratio", 0.0)), float(getattr(recog_args, "minlenratio", 0.0)))
    nbest = int(getattr(recog_args, "nbest", 1))
    return [{"score": float(h.score), "yseq": [int(t) for t in h.yseq], "scores": h.scores} for h in hyps[:nbest]]
